#!/bin/bash
# round 5, GPU call 1: the new tests + same-box A/B of the optimizer lane (VTP_OPT_OVERLAP=0 / 1) and of the bucket size
export PYTHONDONTWRITEBYTECODE=1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r05c1
mkdir -p $O
timeout 900 python -m pytest tests/test_opt_lane_gpu.py tests/test_block_extras_gpu.py tests/test_boundary_gpu.py tests/test_gemm4w_tn_gpu.py \
  "tests/test_model_gpu.py" -x -q -m gpu -s > $O/tests_a.log 2>&1
echo "tests_a rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests_a.log | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_parity_bench_gpu.py -x -q -m gpu -s > $O/tests_parity_bench.log 2>&1
echo "parity_bench rc=$?" | tee -a $O/summary.txt
grep -E "PARITY .*(ALL|POOLED|loss|multi-seed:|oracle fp32)|passed|failed|Error" $O/tests_parity_bench.log | tail -20 | tee -a $O/summary.txt
ab() {
  for rep in 1 2; do
    for cfg in "$@"; do
      v=$(env $cfg timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run $EXTRA 2>$O/ab.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], "host", d.get("host_enqueue_ms_per_step"), d.get("host_split_ms"))')
      echo "[$cfg $EXTRA] $v" | tee -a $O/summary.txt
    done
  done
}
ab "VTP_OPT_OVERLAP=0" "VTP_OPT_OVERLAP=1"
EXTRA="--bucket-blocks 2" ab "VTP_OPT_OVERLAP=1"
EXTRA="--bucket-blocks 1" ab "VTP_OPT_OVERLAP=1"
