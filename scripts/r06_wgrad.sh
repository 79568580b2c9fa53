#!/bin/bash
# round 6: the grouped weight-gradient launch -- correctness of the item-list geometry, timeline, A/B against the uniform geometry
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/${OUT:-r06wg}
mkdir -p $O
timeout 600 python -m pytest tests/test_gemm4w_tn_gpu.py tests/test_gemm8p_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
python tools/wgrad_timeline.py 34144 8192 2>&1 | grep -v amdgpu.ids | tee $O/timeline.log
for e in 0 1; do
  VTP_WGRAD_ITEMS=$e python tools/wgrad_kernel_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[ITEMS=$e] /" | tee -a $O/ab.log
done
