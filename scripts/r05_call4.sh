#!/bin/bash
# round 5, GPU call 4: the whole GPU suite (no -x) + same-box A/B of the balanced persistent grids
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05c4
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -s > $O/tests_all.log 2>&1
echo "tests_all rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests_all.log | tee -a $O/summary.txt
grep -E "^(FAILED|ERROR)" $O/tests_all.log | head -20 | tee -a $O/summary.txt
for rep in 1 2 3; do
  for cfg in "VTP_GRID_BALANCE=0" "VTP_GRID_BALANCE=1"; do
    v=$(env $cfg VTP_BENCH_GEMM_TABLE=$O/gemm_table_${cfg}.txt timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run 2>$O/ab.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], "host", d.get("host_enqueue_ms_per_step"), d.get("host_split_ms"))')
    echo "[$cfg] $v" | tee -a $O/summary.txt
  done
done
