#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_parity_bs_gpu.py --deselect tests/test_parity_ssl_gpu.py 2>&1 | tail -15
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lpips-run > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3h_bench.json'))
print(d['value'], d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), d.get('host_split_ms'))
print('separate', d.get('separate_passes'))
print('roof', d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
tail -3 gpurun_out/r3h_bench.err
