#!/bin/bash
# round 4, call 2: half-size co-resident GEMM (cfg 9): correctness vs cfg 8 / torch, per-shape A/B; bias-gradient diagnostic; smoke
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm8h_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 | cut -c1-250
echo "=== bench"
timeout 600 python tools/gemm8h_bench.py > gpurun_out/r04_gemm8h_bench2.log 2>&1; echo "bench rc=$?"; cat gpurun_out/r04_gemm8h_bench1.log | cut -c1-200
echo "=== attention test"
timeout 600 python -m pytest "tests/test_kernels_gpu.py::test_attention_fwd_bwd" -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/r04_attn_test.log 2>&1; echo "attn rc=$?"; tail -3 gpurun_out/r04_attn_test.log; grep "spiked" gpurun_out/r04_attn_test.log | awk '{print $(NF-8), $(NF-7), $(NF-6), $NF}' | sort | uniq -c | sort -rn | head -5
echo "=== smoke (main flow, then driver flow)"
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r04_smoke_main.log 2>&1; echo "smoke main rc=$?"; tail -4 gpurun_out/r04_smoke_main.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
echo "=== bias diagnostic"
timeout 900 python tools/diag_parity_bias.py 41 43 > gpurun_out/r04_diag_bias.log 2>&1; echo "diag rc=$?"; grep -v Warning gpurun_out/r04_diag_bias.log | cut -c1-330
