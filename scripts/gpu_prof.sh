#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (eager so every kernel is attributed) -> gpurun_out/prof_full
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
mkdir -p $R/gpurun_out/prof_full
cd /tmp && export TMPDIR=/tmp
VTP_OVERLAP=${VTP_OVERLAP:-1} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_full -o full -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-lpips-run --no-separate-run --no-graphs > $R/gpurun_out/prof_full.log 2>&1
echo "prof rc=$?"
cd $R
ls gpurun_out/prof_full | head
f=$(find gpurun_out/prof_full -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-200
# keep only the summary (the trace is large)
find gpurun_out/prof_full -name "*kernel_trace.csv" -delete
find gpurun_out/prof_full -name "*.db" -delete
tail -1 gpurun_out/prof_full.log | cut -c1-300
