#!/bin/bash
# full GPU regression: every gpu-marked test + the three bench workloads
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" > gpurun_out/rc.log
timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_full.log 2>&1
echo "bench full rc=$?" >> gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload vtp_base_rec > gpurun_out/bench_rec.log 2>&1
echo "bench rec rc=$?" >> gpurun_out/rc.log
cat gpurun_out/rc.log
grep -E "passed|failed|FAILED|rror" gpurun_out/t_all.log | tail -8
tail -1 gpurun_out/bench_full.log | cut -c1-200
tail -1 gpurun_out/bench_rec.log | cut -c1-200
