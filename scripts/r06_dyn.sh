#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/${OUT:-r06dyn}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm8p_gpu.py tests/test_gemm8h_gpu.py tests/test_gemm4w_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
for d in 0 1; do VTP_GEMM_DYN=$d python tools/cu_thief.py 2>&1 | grep -v amdgpu.ids | tee -a $O/thief.log; done
VTP_GEMM_DYN=0 VTP_GEMM_CUS=224 python tools/cu_thief.py 2>&1 | grep -v amdgpu.ids | tee -a $O/thief.log
for d in 0 1; do VTP_GEMM_DYN=$d python tools/gemm_shapes.py dyn$d 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes.log; done
