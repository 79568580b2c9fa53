#!/bin/bash
# round-3 call A: new parity / RCCL / kernel tests, GEMM timeline diagnostics, baseline bench of this box
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ssl_gpu.py -x -q -s -p no:cacheprovider > gpurun_out/r3_parity_ssl.log 2>&1; echo "parity_ssl rc=$?"
grep -c "^PARITY" gpurun_out/r3_parity_ssl.log; tail -5 gpurun_out/r3_parity_ssl.log
timeout 600 python -m pytest tests/test_rccl_gpu.py -x -q -s -p no:cacheprovider > gpurun_out/r3_rccl.log 2>&1; echo "rccl rc=$?"
tail -15 gpurun_out/r3_rccl.log
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_ssl_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/gemm8p_timeline.py > gpurun_out/gemm8p_timeline.log 2>&1; echo "timeline rc=$?"
grep "delay sweep" gpurun_out/gemm8p_timeline.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lpips-run > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/r3a_bench.json
