#!/bin/bash
# same-box A/B of an environment switch:  bash scripts/gpu_ab.sh VAR on off
export PYTHONDONTWRITEBYTECODE=1
VAR=${1:-VTP_TEXT_STREAM}; ON=${2:-1}; OFF=${3:-0}
for rep in 1 2; do
for v in $ON $OFF; do
  echo "$VAR=$v: full $(env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c72-90)  rec $(env $VAR=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>/dev/null | tail -1 | cut -c72-90)"
done; done
