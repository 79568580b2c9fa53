#!/bin/bash
# GEMM regression check: kernel tests, the 8-phase timeline of the fp32-residual shape, the in-situ per-shape table of the step
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_model_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/gemm8p_timeline.py 2>&1 > gpurun_out/gemm8p_timeline10.log; grep -A3 "f32res: M=34144" gpurun_out/gemm8p_timeline10.log | grep -v "^--\|per k-tile\|tile  wgs" | cut -c1-150
VTP_BENCH_GEMM_TABLE=gpurun_out/gemm_table_r3p.txt python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], d["roofline"]["frac"])'
head -24 gpurun_out/gemm_table_r3p.txt
