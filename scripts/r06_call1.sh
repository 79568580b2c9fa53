#!/bin/bash
# round 6, GPU call 1: the ADVICE r5 tests (decoder RoPE-augmentation step 2 / graphs, EMA without text) + this round's baseline line
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r06c1
mkdir -p $O
timeout 900 python -m pytest tests/test_block_extras_gpu.py tests/test_opt_lane_gpu.py -x -q -m gpu -s > $O/tests_a.log 2>&1
echo "tests_a rc=$?" | tee -a $O/summary.txt
grep -E "RoPE augmentation|passed|failed|Error|assert" $O/tests_a.log | tail -20 | tee -a $O/summary.txt
for rep in 1 2; do
  timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run 2>$O/bench.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("baseline", d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], d["roofline"]["frac"])' | tee -a $O/summary.txt
done
