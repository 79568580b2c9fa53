#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python bench.py --steps 8 --warmup 3 > gpurun_out/bench_full.log 2>&1
echo "bench full rc=$?" > gpurun_out/rc.log
timeout 600 python bench.py --steps 8 --warmup 3 --no-graphs --no-cpu-baseline > gpurun_out/bench_full_eager.log 2>&1
echo "bench full eager rc=$?" >> gpurun_out/rc.log
cat gpurun_out/rc.log
tail -2 gpurun_out/bench_full.log
tail -1 gpurun_out/bench_full_eager.log | cut -c1-220
