#!/bin/bash
# the round's evidence in one call: parity logs, default bench line, serial-mode kernel stats, both PMC passes
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ssl_gpu.py tests/test_parity_bs_gpu.py -q -s -p no:cacheprovider > gpurun_out/r03_parity.log 2>&1
echo "parity rc=$?"; grep -E "passed|failed" gpurun_out/r03_parity.log | tail -2
timeout 1200 python bench.py > gpurun_out/r03_bench_default_n1.json 2> gpurun_out/r03_bench_default_n1.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/r03_bench_default_n1.json
VTP_OVERLAP=0 bash scripts/gpu_prof.sh > gpurun_out/prof_serial.log 2>&1
cp gpurun_out/prof_full/full_kernel_stats.csv gpurun_out/r03_kernel_stats_full_eager_b32.csv
head -12 gpurun_out/r03_kernel_stats_full_eager_b32.csv | cut -c1-160
bash scripts/gpu_pmc.sh 2>&1 | tail -14
cp gpurun_out/pmc_summary.json gpurun_out/r03_pmc_summary.json
