#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for rep in 1 2; do
for v in 1 0; do
  echo "RESIDENT=$v: full $(VTP_ATTN_RESIDENT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c72-90)  rec $(VTP_ATTN_RESIDENT=$v python bench.py --steps 16 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>/dev/null | tail -1 | cut -c72-90)"
done; done
