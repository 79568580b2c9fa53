#!/bin/bash
# round 5, final build: serial-mode kernel stats, the three PMC passes, then the default bench line quoting THESE counters
export PYTHONDONTWRITEBYTECODE=1
VTP_OVERLAP=0 bash scripts/gpu_prof.sh > gpurun_out/prof_serial.log 2>&1
cp $(find gpurun_out/prof_full -name "*kernel_stats.csv" | head -1) gpurun_out/r05_kernel_stats_full_eager_b32.csv
head -8 gpurun_out/r05_kernel_stats_full_eager_b32.csv | cut -c1-160
bash scripts/gpu_pmc.sh 2>&1 | tail -6
cp gpurun_out/pmc_summary.json gpurun_out/r05_pmc_summary.json
cp gpurun_out/r05_pmc_summary.json profiles/r05_pmc_summary.json
VTP_BENCH_GEMM_TABLE=gpurun_out/r05_gemm_table.txt timeout 900 python bench.py > gpurun_out/r05_bench_default_n1.json 2> gpurun_out/r05_bench_default_n1.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r05_bench_default_n1.json
