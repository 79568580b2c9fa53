#!/bin/bash
# round-3 call B: grouped wgrad kernel test, model / parity tests on the grouped path, same-box A/B of the step
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm8p_gpu.py -x -q -s -p no:cacheprovider -k grouped 2>&1 | grep -v "^\[grouped" | tail -15
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ssl_gpu.py tests/test_block_extras_gpu.py tests/test_boundary_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -6
timeout 900 python -m pytest tests/test_parity_bs_gpu.py tests/test_parity_ssl_gpu.py -x -q -s -p no:cacheprovider > gpurun_out/r3b_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r3b_parity.log
grep "ALL\|losses" gpurun_out/r3b_parity.log | cut -c1-220
REPS=2 bash scripts/gpu_ab.sh "VTP_WGRAD_GROUPED=0" "VTP_WGRAD_GROUPED=1"
