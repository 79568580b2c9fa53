#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
for M in 8224 34144; do
python tools/gemm_bench.py $M 2>&1 | grep "proj_fwd\|w12_fwd\|w3_fwd" | grep "cfg=0 \|cfg=5 \|cfg=4 " | awk '{print $1,$2,$3,$4,$5,$6,$7,$8, $(NF-4),$(NF-3), $(NF-2), $(NF-1), $NF}' | cut -c1-150
done
