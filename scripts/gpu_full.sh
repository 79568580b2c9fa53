#!/bin/bash
# every GPU test, then the default bench line without the CPU baseline / LPIPS / separate-pass legs
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], d["roofline"]["frac"])'
