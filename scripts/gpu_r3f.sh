#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_kernels_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/gemm8p_timeline.py 2>&1 | grep -v "delay sweep" > gpurun_out/gemm8p_timeline3.log; grep -A3 "grid=256" gpurun_out/gemm8p_timeline3.log | head -40
REPS=1 bash scripts/gpu_ab.sh "VTP_WGRAD_GROUPED=1" "VTP_WGRAD_GROUPED=1 VTP_OVERLAP=0"
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-lpips-run --no-graphs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['value'], d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'))"
