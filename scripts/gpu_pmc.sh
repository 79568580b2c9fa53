#!/bin/bash
# HBM traffic counters of the default bench command (eager launches, no stream overlap), one PMC pass per counter
export PYTHONDONTWRITEBYTECODE=1 VTP_OVERLAP=0
R=$PWD
rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-lpips-run --no-graphs > $R/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
cd $R
python tools/pmc_summarize.py gpurun_out/pmc gpurun_out/pmc_summary.json
find gpurun_out/pmc -name "*.csv" -size +1M -delete
