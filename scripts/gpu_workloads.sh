#!/bin/bash
# the other bench workloads (BASELINE configs 2, 4, 5 and the rec-only VTP-B step): one line each
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
for w in vtp_small_rec vtp_base_rec vtp_large_full_512 vtp_large_fp8_fwd; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run 2>/dev/null | tail -1 > gpurun_out/${TAG:-r04}_bench_$w.json
  python - "$w" <<'PY'
import json, sys
w = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/{__import__('os').environ.get('TAG', 'r04')}_bench_{w}.json"))
    r = d.get("roofline") or {}
    print(w, d["value"], d["unit"], d["ms_per_step"], "ms", "step_tflops", d.get("step_tflops_per_gpu"), "frac", d.get("step_frac"), "gemm", r.get("achieved"), r.get("frac"), {k: v for k, v in d.items() if k.startswith("fp8")})
except Exception as e:
    print(w, "FAILED", e)
PY
done
