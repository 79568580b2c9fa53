#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for rep in 1 2; do
  echo "full $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c72-90)  rec $(python bench.py --steps 16 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>/dev/null | tail -1 | cut -c72-90)"
done
