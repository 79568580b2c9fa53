#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm8p_gpu.py -x -q -s -p no:cacheprovider -k grouped 2>&1 | grep -v "^\[grouped" | tail -8
timeout 300 python tools/wgrad_group_bench.py 2>&1 | tee gpurun_out/wgrad_group.log | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ssl_gpu.py tests/test_block_extras_gpu.py tests/test_boundary_gpu.py tests/test_parity_bs_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -6
VTP_OVERLAP=0 bash scripts/gpu_prof.sh 2>&1 | tail -45
