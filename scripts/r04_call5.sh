#!/bin/bash
# A/B: rebalanced fragment reads in the 8-phase NT k loop (new) vs before (old): correctness, per-shape, and the step
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
VTP_HIP_LIB=$PWD/vtp_amd/lib/libvtp_hip_new.so timeout 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm8h_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4 | cut -c1-250
for v in old new old new; do
  VTP_HIP_LIB=$PWD/vtp_amd/lib/libvtp_hip_$v.so timeout 600 python tools/gemm8h_bench.py quick 2>/dev/null | awk -v v=$v '{printf "%s %-16s %s %s | 8p %s us %s TF/s\n", v, $1, $2, $4, $(NF-4), $(NF-2)}'
done > gpurun_out/r04_rebal_ab.log
sort -k2,2 -k3,3 -k1,1 gpurun_out/r04_rebal_ab.log | cut -c1-150
bash scripts/gpu_ab.sh "VTP_HIP_LIB=$PWD/vtp_amd/lib/libvtp_hip_old.so" "VTP_HIP_LIB=$PWD/vtp_amd/lib/libvtp_hip_new.so"
