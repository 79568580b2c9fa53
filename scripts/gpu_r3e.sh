#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm8p_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/wgrad_group_bench.py 2>&1 | tee gpurun_out/wgrad_group2.log | tail -8
timeout 300 python tools/gemm8p_timeline.py 2>&1 | grep -v "delay sweep" > gpurun_out/gemm8p_timeline2.log; grep -A3 "grid=256" gpurun_out/gemm8p_timeline2.log | head -40
REPS=2 bash scripts/gpu_ab.sh "VTP_WGRAD_GROUPED=0" "VTP_WGRAD_GROUPED=1"
