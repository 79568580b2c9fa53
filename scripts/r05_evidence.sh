#!/bin/bash
# round 5 evidence, one call, final build: all GPU tests (parity lines kept), serial-mode kernel stats, PMC passes, default bench line
# (CPU-oracle leg included), the other workloads, the vendor-library context table, the lane A/B
export PYTHONDONTWRITEBYTECODE=1 TAG=r05
bash scripts/gpu_evidence.sh 2>&1 | tail -40
bash scripts/gpu_workloads.sh 2>&1 | tail -8
timeout 600 python tools/vendor_gemm_ref.py > gpurun_out/r05_vendor_gemm_ref.log 2>&1; tail -16 gpurun_out/r05_vendor_gemm_ref.log
for rep in 1 2 3; do
  for cfg in "VTP_OPT_OVERLAP=0" "VTP_OPT_OVERLAP=1"; do
    v=$(env $cfg timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"])')
    echo "[$cfg] $v" | tee -a gpurun_out/r05_lane_step_ab.log
  done
done
