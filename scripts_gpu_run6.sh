#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" > gpurun_out/rc.log
cat gpurun_out/rc.log
grep -E "passed|failed|FAILED|Error" gpurun_out/t_all.log | tail -15
