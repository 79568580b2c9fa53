#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for rep in 1 2; do
for v in 1 0; do
  echo "TEXT_STREAM=$v: full $(VTP_TEXT_STREAM=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c72-90)"
done; done
