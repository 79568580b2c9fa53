#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1 VTP_OVERLAP=0
R=$PWD
rm -rf $R/gpurun_out/prof_lp; mkdir -p $R/gpurun_out/prof_lp
timeout 600 python -m pytest tests/test_lpips_gpu.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lp -o lp -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graphs --workload vtp_base_rec --perceptual-weight 1.0 > $R/gpurun_out/prof_lp.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_lp/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('lpips','maxpool','6, false, false','7, false, false')):
        print(n[:90], r['Calls'], round(float(r['TotalDurationNs'])/5e6,3),'ms/step', round(float(r['AverageNs'])/1e3,1),'us')
PY
find gpurun_out/prof_lp -name "*kernel_trace.csv" -delete
