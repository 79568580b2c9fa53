#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" > gpurun_out/rc.log
timeout 600 python tools/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1
echo "gemm_bench rc=$?" >> gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1
echo "bench graph rc=$?" >> gpurun_out/rc.log
cat gpurun_out/rc.log
tail -4 gpurun_out/t_all.log
tail -1 gpurun_out/bench_graph.log | cut -c1-600
