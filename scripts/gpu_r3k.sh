#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/wgrad_group_bench.py 2>&1 | grep -v amdgpu
