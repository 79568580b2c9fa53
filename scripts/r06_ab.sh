#!/bin/bash
# same-box A/B of the step: scripts/r06_ab.sh "ENV=a" "ENV=b" ...  (REPS repetitions, interleaved; EXTRA = more bench.py flags)
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/${OUT:-r06ab}
mkdir -p $O
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    v=$(env $cfg timeout 400 python bench.py --steps ${STEPS:-12} --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run $EXTRA 2>$O/ab.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], d["roofline"]["frac"])')
    echo "[$cfg] $v" | tee -a $O/summary.txt
  done
done
