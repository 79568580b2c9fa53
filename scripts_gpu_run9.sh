#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
VTP_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01d -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graphs --workload vtp_base_rec > $R/gpurun_out/bench_prof.log 2>&1
echo "prof rc=$?" > $R/gpurun_out/rc.log
cd $R
cat gpurun_out/rc.log
tail -1 gpurun_out/bench_prof.log | cut -c1-200
