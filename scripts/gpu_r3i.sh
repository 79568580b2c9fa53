#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 300 python tools/gemm8p_timeline.py 2>&1 > gpurun_out/gemm8p_timeline6.log; grep -A3 "^==" gpurun_out/gemm8p_timeline6.log | grep -v "^--" | cut -c1-200
