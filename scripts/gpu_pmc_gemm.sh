#!/bin/bash
# SQ / LDS counters of one GEMM shape (separate --pmc passes, kernel-trace only): where the wave cycles of the main loop go
# usage: scripts/gpu_pmc_gemm.sh <tag> <nt|tn> M N K cfg [splits]
export PYTHONDONTWRITEBYTECODE=1
R=$PWD; TAG=$1; shift
OUT=$R/gpurun_out/pmc_gemm_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $R/tools/one_gemm.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "gemm" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[1:] if len(v) > 1 else v  # drop the first (cold) launch
        print(f"   {c:28s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
find $OUT -name "*.csv" -size +1M -delete
