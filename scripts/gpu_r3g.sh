#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_ssl_gpu.py tests/test_rccl_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
REPS=1 bash scripts/gpu_ab.sh "VTP_WGRAD_GROUPED=1" "VTP_WGRAD_GROUPED=0" "VTP_WGRAD_GROUPED=1"
