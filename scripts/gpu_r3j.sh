#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm8p_gpu.py tests/test_model_gpu.py tests/test_boundary_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
REPS=2 bash scripts/gpu_ab.sh "VTP_WGRAD_GROUPED=1"
