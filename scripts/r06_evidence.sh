#!/bin/bash
# round 6 evidence on ONE build: serial-mode kernel stats, the three PMC passes, the default bench line quoting THOSE counters, the in-situ
# GEMM table, the other workloads, the vendor-library context table, the trace-gap report, the isolated weight-gradient launch counters,
# and (FULL=1) every GPU test with the parity lines kept.  Everything lands in gpurun_out/r06ev/ (copied into profiles/ afterwards).
export PYTHONDONTWRITEBYTECODE=1 TAG=r06
O=gpurun_out/r06ev; mkdir -p $O
VTP_OVERLAP=0 bash scripts/gpu_prof.sh > $O/prof_serial.log 2>&1
cp $(find gpurun_out/prof_full -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_full_eager_b32.csv
head -6 $O/r06_kernel_stats_full_eager_b32.csv | cut -c1-150
bash scripts/gpu_pmc.sh 2>&1 | tail -4
cp gpurun_out/pmc_summary.json $O/r06_pmc_summary.json
mkdir -p profiles && cp $O/r06_pmc_summary.json profiles/r06_pmc_summary.json && cp $O/r06_kernel_stats_full_eager_b32.csv profiles/r06_kernel_stats_full_eager_b32.csv
VTP_BENCH_GEMM_TABLE=$O/r06_gemm_table.txt timeout 900 python bench.py > $O/r06_bench_default_n1.json 2> $O/r06_bench_default_n1.err
echo "bench rc=$?"; cut -c1-260 $O/r06_bench_default_n1.json
bash scripts/gpu_workloads.sh 2>&1 | tail -6; mv gpurun_out/r06_bench_vtp_*.json $O/ 2>/dev/null
timeout 600 python tools/vendor_gemm_ref.py > $O/r06_vendor_gemm_ref.log 2>&1; tail -17 $O/r06_vendor_gemm_ref.log
bash scripts/r05_trace.sh > $O/trace.log 2>&1; cp gpurun_out/trace_gaps.txt $O/r06_trace_gaps.txt; head -4 $O/r06_trace_gaps.txt; rm -f gpurun_out/trace_kernels.csv
bash scripts/r06_pmc_wgrad.sh final 1 > /dev/null 2>&1; cp gpurun_out/r06_pmc_wgrad/final.txt $O/r06_pmc_wgrad.txt; cat $O/r06_pmc_wgrad.txt
python tools/wgrad_timeline.py 34144 8192 2>&1 | grep -v amdgpu.ids > $O/r06_wgrad_timeline.log; head -1 $O/r06_wgrad_timeline.log
python tools/wgrad_kernel_ab.py 2>&1 | grep -v amdgpu.ids > $O/r06_wgrad_kernel_ab.log
python tools/gemm_shapes.py final 2>&1 | grep -v amdgpu.ids > $O/r06_gemm_shapes.log
python tools/norm_bench.py final 2>&1 | grep -v amdgpu.ids > $O/r06_norm_bench.log
if [ "$FULL" = "1" ]; then
  timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/r06_gputests.log 2>&1
  echo "gpu tests rc=$?"; grep -E "passed|failed" $O/r06_gputests.log | tail -2
  grep -E "^\.*F*(PARITY|TOOLS)|\[attn_bwd|split-K combine|fp8 vs bf16|resume shard|grouped wgrad|optimizer lane|RoPE augmentation" $O/r06_gputests.log | sed 's/^[.F]*//' > $O/r06_parity.log
  wc -l $O/r06_parity.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
find $O -size +4M -delete
