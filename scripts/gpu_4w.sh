#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm4w_tn_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r04_4w_tests.log
cat gpurun_out/r04_4w_tests.log
timeout 300 python tools/wgrad_kernel_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_wgrad_kernel_ab3.log
cat gpurun_out/r04_wgrad_kernel_ab3.log
