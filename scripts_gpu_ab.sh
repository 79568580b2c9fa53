#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
python - <<'PY'
import os
os.environ["PYTHONDONTWRITEBYTECODE"]="1"
PY
for rep in 1 2; do
for v in 3 1; do
  echo "SWZ=$v: full $(VTP_GEMM_SWZ=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c72-90)  rec $(VTP_GEMM_SWZ=$v python bench.py --steps 16 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>/dev/null | tail -1 | cut -c72-90)"
done; done
