#!/bin/bash
# round 6, final tree (after the token-assembly backward kernel and the patch-embed gradient in the grouped launch changed the kernel mix):
# serial-mode kernel stats, the three PMC passes, the default bench line quoting THOSE counters, the in-situ GEMM table, the other
# workloads, the trace-gap report, every GPU test with the parity lines kept, smoke().  -> gpurun_out/r06ev3/
export PYTHONDONTWRITEBYTECODE=1 TAG=r06
O=gpurun_out/r06ev3; mkdir -p $O
VTP_OVERLAP=0 bash scripts/gpu_prof.sh > $O/prof_serial.log 2>&1
cp $(find gpurun_out/prof_full -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_full_eager_b32.csv
head -4 $O/r06_kernel_stats_full_eager_b32.csv | cut -c1-150
bash scripts/gpu_pmc.sh 2>&1 | tail -3
cp gpurun_out/pmc_summary.json $O/r06_pmc_summary.json
cp $O/r06_pmc_summary.json profiles/r06_pmc_summary.json; cp $O/r06_kernel_stats_full_eager_b32.csv profiles/r06_kernel_stats_full_eager_b32.csv
VTP_BENCH_GEMM_TABLE=$O/r06_gemm_table.txt timeout 900 python bench.py > $O/r06_bench_default_n1.json 2> $O/r06_bench_default_n1.err
echo "bench rc=$?"; cut -c1-260 $O/r06_bench_default_n1.json; grep "per-step ms" $O/r06_bench_default_n1.err | head -1 | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/r06_bench_default_n1_run2.json 2>/dev/null; cut -c1-200 $O/r06_bench_default_n1_run2.json
bash scripts/gpu_workloads.sh 2>&1 | tail -6; mv gpurun_out/r06_bench_vtp_*.json $O/ 2>/dev/null
bash scripts/r05_trace.sh > $O/trace.log 2>&1; cp gpurun_out/trace_gaps.txt $O/r06_trace_gaps.txt; head -3 $O/r06_trace_gaps.txt; rm -f gpurun_out/trace_kernels.csv
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/r06_gputests.log 2>&1
echo "gpu tests rc=$?"; grep -E "passed|failed" $O/r06_gputests.log | tail -2
grep -E "^\.*F*(PARITY|TOOLS)|\[attn_bwd|split-K combine|fp8 vs bf16|resume shard|grouped wgrad|optimizer lane|RoPE augmentation" $O/r06_gputests.log | sed 's/^[.F]*//' > $O/r06_parity.log
wc -l $O/r06_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/gpu_n2.sh 2>&1 | grep "rc=" | tee $O/rehearsal.log
bash scripts/gpu_n8.sh 2>&1 | grep "rc=" | tee -a $O/rehearsal.log
find $O -size +4M -delete
