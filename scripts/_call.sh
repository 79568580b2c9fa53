mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gemm4w_tn_gpu.py -x -q 2>&1 | tail -2
timeout 200 python tools/wgrad_kernel_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_wgrad_kernel_ab4.log
