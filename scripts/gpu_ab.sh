#!/bin/bash
# same-box A/B of the step rate: scripts/gpu_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one configuration's env;
# WORKLOAD=<bench workload> selects another workload, REPS the repetitions)
export PYTHONDONTWRITEBYTECODE=1
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    v=$(env $cfg timeout 400 python bench.py --workload ${WORKLOAD:-vtp_base_full} --steps 10 --warmup 3 --no-cpu-baseline --no-lpips-run 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], "host", d.get("host_enqueue_ms_per_step"), d.get("host_split_ms"))')
    echo "[$cfg] $v"
  done
done
