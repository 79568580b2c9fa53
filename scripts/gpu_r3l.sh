#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_kernels_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/gemm8p_timeline.py 2>&1 > gpurun_out/gemm8p_timeline7.log; grep -A3 "grid=256" gpurun_out/gemm8p_timeline7.log | grep -v "^--\|per k-tile" | cut -c1-150
REPS=2 bash scripts/gpu_ab.sh "VTP_WGRAD_GROUPED=1"
