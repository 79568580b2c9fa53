#!/bin/bash
# one-wave-per-SIMD kernels in the step: all GEMM / model / parity tests, then the same-box step A/B
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gemm4w_gpu.py tests/test_gemm4w_tn_gpu.py tests/test_gemm8p_gpu.py tests/test_model_gpu.py tests/test_parity_ssl_gpu.py tests/test_parity_gpu.py tests/test_ddp_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r04_4w_tests.log
cat gpurun_out/r04_4w_tests.log
REPS=2 bash scripts/gpu_ab.sh "VTP_GEMM4W=0 VTP_GEMM4W_TN=0" "VTP_GEMM4W=1 VTP_GEMM4W_TN=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_4w_step_ab.log
cat gpurun_out/r04_4w_step_ab.log
