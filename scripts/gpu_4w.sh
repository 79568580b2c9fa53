#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm4w_tn_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r04_4w_tests.log
cat gpurun_out/r04_4w_tests.log
