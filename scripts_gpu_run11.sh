#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
VTP_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_full -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graphs > $R/gpurun_out/bench_prof.log 2>&1
echo "prof rc=$?" > $R/gpurun_out/rc.log
cd $R; cat gpurun_out/rc.log
