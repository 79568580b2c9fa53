#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" > gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_tn.log 2>&1
echo "bench tn rc=$?" >> gpurun_out/rc.log
VTP_WGRAD=transpose timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_tr.log 2>&1
echo "bench transpose rc=$?" >> gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload vtp_base_rec > gpurun_out/bench_rec.log 2>&1
cat gpurun_out/rc.log
grep -E "passed|failed|FAILED" gpurun_out/t_all.log | tail -8
tail -1 gpurun_out/bench_tn.log | cut -c1-200
tail -1 gpurun_out/bench_tr.log | cut -c1-200
tail -1 gpurun_out/bench_rec.log | cut -c1-200
