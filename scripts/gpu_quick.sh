#!/bin/bash
# kernel + model parity tests, then one bench line of each workload (quick check after a kernel change)
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_ssl_gpu.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
echo "full $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c72-90)  rec $(python bench.py --steps 12 --warmup 3 --no-cpu-baseline --workload vtp_base_rec 2>/dev/null | tail -1 | cut -c72-90)"
