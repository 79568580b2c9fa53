#!/bin/bash
# final tree of round 5: every GPU test (parity lines kept), smoke, serial-mode kernel stats, default bench line with the CPU-oracle leg
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r05_gputests.log 2>&1
echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r05_gputests.log | tail -2
grep -E "^\.*F*(PARITY|TOOLS)|\[attn_bwd|split-K combine|fp8 vs bf16|resume shard|grouped wgrad|optimizer lane" gpurun_out/r05_gputests.log | sed 's/^[.F]*//' > gpurun_out/r05_parity.log
wc -l gpurun_out/r05_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
VTP_BENCH_GEMM_TABLE=gpurun_out/r05_gemm_table.txt timeout 900 python bench.py > gpurun_out/r05_bench_default_n1.json 2> gpurun_out/r05_bench_default_n1.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r05_bench_default_n1.json; grep "per-step" gpurun_out/r05_bench_default_n1.err | cut -c1-300
