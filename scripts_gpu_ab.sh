#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for rep in 1 2; do
for mk in 1024 4096 100000000; do
  echo "PIPE_MINK=$mk full:"; VTP_GEMM_PIPE_MINK=$mk python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-140
  echo "PIPE_MINK=$mk rec:"; VTP_GEMM_PIPE_MINK=$mk python bench.py --steps 16 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>/dev/null | tail -1 | cut -c60-140
done; done
