#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm8h_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15 | cut -c1-250
timeout 600 python tools/gemm8h_bench.py > gpurun_out/r04_gemm8h_bench2.log 2>&1; echo "bench rc=$?"; cat gpurun_out/r04_gemm8h_bench2.log | cut -c1-230
