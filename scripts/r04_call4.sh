#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | cut -c1-250
timeout 1200 python -m pytest tests/test_parity_large_gpu.py tests/test_gemm8h_gpu.py "tests/test_kernels_gpu.py::test_attention_fwd_bwd" -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/r04_newtests2.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r04_newtests2.log | cut -c1-250; grep "POOLED\|ALL " gpurun_out/r04_newtests2.log | cut -c1-250
VTP_BENCH_GEMM_TABLE=gpurun_out/r04_gemm_table_8h.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run 2>gpurun_out/r04_bench_8h.err | tail -1 > gpurun_out/r04_bench_8h.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_8h.json')); r=d['roofline']; print('8H', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['gemm_ms_per_step'], r.get('algorithmic_bytes_per_launch_avg'), r.get('traffic_over_algorithmic'), r.get('mfma_util'))"
head -24 gpurun_out/r04_gemm_table_8h.txt
