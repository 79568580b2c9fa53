#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1 VTP_OVERLAP=0
R=$PWD
rm -rf $R/gpurun_out/prof_at; mkdir -p $R/gpurun_out/prof_at
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_at -o at -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graphs > $R/gpurun_out/prof_at.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_at/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/step', tot/5e6)
for r in rows:
    n=r['Name']
    if 'attn' in n or 'rope' in n:
        print(n[:70], r['Calls'], round(float(r['TotalDurationNs'])/5e6,3),'ms/step', round(float(r['AverageNs'])/1e3,1),'us', r['MinNs'], r['MaxNs'])
PY
find gpurun_out/prof_at -name "*kernel_trace.csv" -delete
