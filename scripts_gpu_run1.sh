#!/bin/bash
# first GPU bring-up: kernel parity tests, model parity tests, smoke, short bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" > gpurun_out/rc.log
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/t_model.log 2>&1
echo "model rc=$?" >> gpurun_out/rc.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/rc.log
cat gpurun_out/rc.log
tail -5 gpurun_out/t_kernels.log
tail -5 gpurun_out/t_model.log
tail -3 gpurun_out/bench.log
