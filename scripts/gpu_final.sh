#!/bin/bash
# the round's final call: evidence (tests, kernel stats, PMC passes, bench line), the other workloads, the per-shape A/B of cfg 10
bash scripts/gpu_evidence.sh
bash scripts/gpu_workloads.sh
ALT_CFG=10 timeout 400 python tools/gemm8h_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_gemm4w_bench.log
tail -3 gpurun_out/r04_gemm4w_bench.log
