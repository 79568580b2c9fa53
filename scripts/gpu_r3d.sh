#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
bash scripts/gpu_pmc_gemm.sh r3_nt nt 34144 768 4096 8 2>&1 | tail -20
bash scripts/gpu_pmc_gemm.sh r3_tn tn 4096 768 34144 8 5 2>&1 | tail -20
