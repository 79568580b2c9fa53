#!/bin/bash
# the round's evidence in one call: every GPU test with the parity lines kept, serial-mode kernel stats, the PMC passes, then the
# default bench line (driver contract, CPU baseline included) with the per-shape GEMM table.  TAG=r04 by default.
export PYTHONDONTWRITEBYTECODE=1
R=$PWD; T=${TAG:-r04}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${T}_gputests.log 2>&1
echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/${T}_gputests.log | tail -2
grep -E "^\.*F*(PARITY|TOOLS)|\[attn_bwd|split-K combine|fp8 vs bf16|resume shard|grouped wgrad" gpurun_out/${T}_gputests.log | sed 's/^[.F]*//' > gpurun_out/${T}_parity.log
wc -l gpurun_out/${T}_parity.log
VTP_OVERLAP=0 bash scripts/gpu_prof.sh > gpurun_out/prof_serial.log 2>&1
cp gpurun_out/prof_full/full_kernel_stats.csv gpurun_out/${T}_kernel_stats_full_eager_b32.csv
head -12 gpurun_out/${T}_kernel_stats_full_eager_b32.csv | cut -c1-160
bash scripts/gpu_pmc.sh 2>&1 | tail -16
cp gpurun_out/pmc_summary.json gpurun_out/${T}_pmc_summary.json
# the bench line last: its roofline block quotes the SQ counters of THIS build (the summary just written), not of an earlier one
cp gpurun_out/${T}_pmc_summary.json profiles/${T}_pmc_summary.json
VTP_BENCH_GEMM_TABLE=gpurun_out/${T}_gemm_table.txt timeout 1200 python bench.py > gpurun_out/${T}_bench_default_n1.json 2> gpurun_out/${T}_bench_default_n1.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/${T}_bench_default_n1.json
