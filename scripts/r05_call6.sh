#!/bin/bash
# round 5, GPU call 6: main chain captured on a high-priority stream (VTP_MAIN_PRIO) -- same-box A/B
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05c6
mkdir -p $O
for rep in 1 2 3; do
  for cfg in "VTP_MAIN_PRIO=0" "VTP_MAIN_PRIO=-1"; do
    v=$(env $cfg timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run 2>$O/ab.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], d["loss"])')
    echo "[$cfg] $v" | tee -a $O/summary.txt
  done
done
tail -3 $O/ab.err
