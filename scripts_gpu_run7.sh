#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_rec_clip.log 2>&1
echo "bench rec_clip rc=$?" > gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --workload vtp_base_rec --no-cpu-baseline > gpurun_out/bench_rec.log 2>&1
echo "bench rec rc=$?" >> gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --workload vtp_small_rec --no-cpu-baseline > gpurun_out/bench_small.log 2>&1
echo "bench small rc=$?" >> gpurun_out/rc.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01c -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graphs > $R/gpurun_out/bench_prof.log 2>&1
echo "prof rc=$?" >> $R/gpurun_out/rc.log
cd $R
cat gpurun_out/rc.log
tail -1 gpurun_out/bench_rec_clip.log
tail -1 gpurun_out/bench_rec.log | cut -c1-250
tail -1 gpurun_out/bench_small.log | cut -c1-250
