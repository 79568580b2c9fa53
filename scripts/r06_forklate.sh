#!/bin/bash
# round 6: issue order of forked side work inside a captured segment (engine.Overlap.defer; VTP_FORK_LATE=0 = the old order):
# parity of the step first, then the same-box A/B, then one traced step per setting (queue ids, idle gaps)
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
O=gpurun_out/r06fl
mkdir -p $O
timeout 900 python -m pytest tests/test_parity_ssl_gpu.py tests/test_trainer_inputs_gpu.py tests/test_opt_lane_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
OUT=r06fl REPS=${REPS:-3} STEPS=20 bash scripts/r06_ab.sh "VTP_FORK_LATE=0" "VTP_FORK_LATE=1"
for v in 0 1; do
  rm -rf $O/trace
  (cd /tmp && export TMPDIR=/tmp && VTP_FORK_LATE=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run > $R/$O/trace_bench_$v.log 2>&1)
  python - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob("gpurun_out/r06fl/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "dino_ce_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-3], marks[-2]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
end = 0; idle = 0.0; n10 = 0
with open(f"gpurun_out/r06fl/trace_step_{v}.csv", "w") as fh:
    fh.write("start_us,dur_us,queue,kernel\n")
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if end and s > end:
            idle += (s - end) / 1e3
            n10 += (s - end) > 8000
        end = max(end, e)
        fh.write(f'{(s - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},{r.get("Queue_Id","")},"{r["Kernel_Name"][:70]}"\n')
print(f"VTP_FORK_LATE={v}: {len(sel)} dispatches, step {(t1 - t0) / 1e6:.3f} ms, all-queues-idle {idle:.1f} us in the step, gaps > 8 us: {n10}")
PY
done 2>&1 | tee -a $O/summary.txt
rm -rf $O/trace
