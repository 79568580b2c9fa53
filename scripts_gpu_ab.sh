#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for rep in 1 2; do
for v in "1 600" "0 600" "1 384" "1 100000000" "0 100000000"; do
  set -- $v
  echo "PERSIST=$1 BIG_TILES=$2: full $(VTP_GEMM_PERSIST=$1 VTP_GEMM_BIG_TILES=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c72-90)  rec $(VTP_GEMM_PERSIST=$1 VTP_GEMM_BIG_TILES=$2 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>/dev/null | tail -1 | cut -c72-90)"
done; done
