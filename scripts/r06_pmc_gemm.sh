#!/bin/bash
# round 6: HBM-side traffic and L2 hit rate of single NT GEMM launches (plain bf16 epilogue, default dispatch = cfg -1) against their
# algorithmic bytes: where does the GEMM family's 1.6 x come from?
export PYTHONDONTWRITEBYTECODE=1
R=$PWD; mkdir -p gpurun_out/r06_pmc_gemm
IFS=';' read -ra LIST <<< "${SHAPES:-34144 2304 768;34144 768 768;34144 768 4096;34144 4096 768;16448 2304 768;8192 768 4096}"
for shape in "${LIST[@]}"; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf gpurun_out/pmc_g; cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g -o pmc -- python $R/tools/one_gemm.py nt $shape -1 > $R/gpurun_out/pmc_g.log 2>&1
    cd $R
    python - "$shape" <<'PY' | tee -a gpurun_out/r06_pmc_gemm/summary.txt
import csv, glob, sys
from collections import defaultdict
acc, n, name = defaultdict(float), defaultdict(int), ""
for f in glob.glob("gpurun_out/pmc_g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1; name = r["Kernel_Name"][:50]
M, N, K = (int(x) for x in sys.argv[1].split())
alg = 2.0 * K * (M + N) + 2.0 * M * N
out = {k: round(v / n[k], 1) for k, v in acc.items()}
extra = ""
if "FETCH_SIZE" in out: extra = f" fetch {2 * out['FETCH_SIZE'] * 1024 / 1e6:.1f} MB vs operands {2.0 * K * (M + N) / 1e6:.1f} MB"
if "WRITE_SIZE" in out: extra = f" write {out['WRITE_SIZE'] * 1024 / 1e6:.1f} MB vs result {2.0 * M * N / 1e6:.1f} MB"
print(sys.argv[1], name, out, extra)
PY
  done
done
