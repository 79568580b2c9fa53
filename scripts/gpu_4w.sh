#!/bin/bash
# development call for the one-wave-per-SIMD GEMM (cfg 10): parity tests, then timing
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gemm4w_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r04_4w_tests.log
cat gpurun_out/r04_4w_tests.log
timeout 200 python tools/gemm4w_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_4w_diag.log
cat gpurun_out/r04_4w_diag.log
