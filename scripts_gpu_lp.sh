#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_lpips_gpu.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --perceptual-weight 1.0 2>&1 | tail -1 > gpurun_out/bench_full_lpips.json; cut -c1-200 gpurun_out/bench_full_lpips.json
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --workload vtp_base_rec --perceptual-weight 1.0 2>&1 | tail -1 > gpurun_out/bench_rec_lpips.json; cut -c1-200 gpurun_out/bench_rec_lpips.json
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>&1 | tail -1 | cut -c1-200
