#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/${OUT:-r06wg2}
mkdir -p $O
timeout 600 python -m pytest tests/test_gemm4w_tn_gpu.py tests/test_gemm8p_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
python tools/wgrad_timeline.py 34144 8192 2>&1 | grep -v amdgpu.ids | grep Ktok | tee $O/timeline.log
ACC=0 python tools/wgrad_timeline.py 34144 8192 2>&1 | grep -v amdgpu.ids | grep Ktok | sed 's/^/[ACC=0] /' | tee -a $O/timeline.log
python tools/wgrad_kernel_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
OUT=${OUT:-r06wg2} REPS=2 bash scripts/r06_ab.sh "VTP_WGRAD_ITEMS=0 VTP_WGRAD_OVERWRITE=0" "VTP_WGRAD_ITEMS=1 VTP_WGRAD_OVERWRITE=0" "VTP_WGRAD_ITEMS=1 VTP_WGRAD_OVERWRITE=1" "VTP_WGRAD_ITEMS=1 VTP_WGRAD_OVERWRITE=1 VTP_WGRAD_INLINE=1"
