#!/bin/bash
# round 6: HBM-side traffic and L2 hit rate of ONE grouped weight-gradient launch in isolation (trunk: 34 144 token rows, decoder: 8 192)
# usage: scripts/r06_pmc_wgrad.sh <tag> [kernel id, default 1]
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
TAG=${1:-base}; KERN=${2:-1}
mkdir -p gpurun_out/r06_pmc_wgrad
for K in 34144 8192; do
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    i=$((i+1))
    rm -rf gpurun_out/pmc_w; cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -o pmc -- python $R/tools/one_wgrad_group.py $KERN $K > $R/gpurun_out/pmc_w.log 2>&1
    cd $R
    python - "$K" "$c" <<'PY' | tee -a gpurun_out/r06_pmc_wgrad/$TAG.txt
import csv, glob, sys
from collections import defaultdict
acc, n = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/pmc_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "grouped" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("Ktok", sys.argv[1], {k: round(v / n[k], 1) for k, v in acc.items()}, "dispatches", dict(n))
PY
  done
done
