#!/bin/bash
# measurement bundle for profiles/: default bench line (with cpu_baseline), rocprofv3 kernel stats of the same command in eager
# / no-overlap mode (per-kernel durations comparable with bench's HIP-event roofline), PMC traffic passes
export PYTHONDONTWRITEBYTECODE=1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_default.json
VTP_OVERLAP=0 bash scripts/gpu_prof.sh > gpurun_out/prof_run.log 2>&1; tail -3 gpurun_out/prof_run.log | cut -c1-300
bash scripts/gpu_pmc.sh
