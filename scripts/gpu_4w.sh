#!/bin/bash
mkdir -p gpurun_out
L=$PWD/vtp_amd/lib
VTP_HIP_LIB=$L/libvtp_hip_2ph.so timeout 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm8h_gpu.py tests/test_fp8_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/r04_2ph_tests.log
cat gpurun_out/r04_2ph_tests.log
REPS=3 bash scripts/gpu_ab.sh "VTP_HIP_LIB=$L/libvtp_hip.so" "VTP_HIP_LIB=$L/libvtp_hip_2ph.so" 2>&1 | grep -v amdgpu.ids | sed "s#$L/##" > gpurun_out/r04_2ph_step_ab.log
cat gpurun_out/r04_2ph_step_ab.log
