#!/bin/bash
# round 6, after the queue-placement change (the library is the one of scripts/r06_evidence.sh: kernel stats, counters and the in-situ
# GEMM table stay valid): the default bench line (counters replayed from profiles/r06_pmc_summary.json), the other workloads, the
# trace-gap report of the new order, the multi-rank rehearsals, every GPU test with the parity lines kept, smoke().  -> gpurun_out/r06ev2/
export PYTHONDONTWRITEBYTECODE=1 TAG=r06
O=gpurun_out/r06ev2; mkdir -p $O
timeout 900 python bench.py > $O/r06_bench_default_n1.json 2> $O/r06_bench_default_n1.err
echo "bench rc=$?"; cut -c1-260 $O/r06_bench_default_n1.json; grep "per-step ms" $O/r06_bench_default_n1.err | head -1 | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/r06_bench_default_n1_run2.json 2>/dev/null; cut -c1-200 $O/r06_bench_default_n1_run2.json
bash scripts/gpu_workloads.sh 2>&1 | tail -6; mv gpurun_out/r06_bench_vtp_*.json $O/ 2>/dev/null
bash scripts/r05_trace.sh > $O/trace.log 2>&1; cp gpurun_out/trace_gaps.txt $O/r06_trace_gaps.txt; head -4 $O/r06_trace_gaps.txt; rm -f gpurun_out/trace_kernels.csv
bash scripts/gpu_n2.sh 2>&1 | tail -8 | tee $O/rehearsal_n2.log
bash scripts/gpu_n8.sh 2>&1 | tail -6 | tee $O/rehearsal_n8.log
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/r06_gputests.log 2>&1
echo "gpu tests rc=$?"; grep -E "passed|failed" $O/r06_gputests.log | tail -2
grep -E "^\.*F*(PARITY|TOOLS)|\[attn_bwd|split-K combine|fp8 vs bf16|resume shard|grouped wgrad|optimizer lane|RoPE augmentation" $O/r06_gputests.log | sed 's/^[.F]*//' > $O/r06_parity.log
wc -l $O/r06_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
find $O -size +4M -delete
