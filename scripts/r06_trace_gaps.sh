#!/bin/bash
# round 6: one traced step of the default bench (queue ids, all-queues-idle gaps) -> gpurun_out/$OUT/trace_step.csv + summary line
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
O=gpurun_out/${OUT:-r06tg}
mkdir -p $O
rm -rf $O/trace
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run $EXTRA > $R/$O/trace_bench.log 2>&1)
python - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "dino_ce_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-3], marks[-2]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
end = 0; idle = 0.0; n8 = 0
with open(O + "/trace_step.csv", "w") as fh:
    fh.write("start_us,dur_us,queue,kernel\n")
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if end and s > end:
            idle += (s - end) / 1e3
            n8 += (s - end) > 8000
        end = max(end, e)
        fh.write(f'{(s - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},{r.get("Queue_Id","")},"{r["Kernel_Name"][:70]}"\n')
print(f"{len(sel)} dispatches, step {(t1 - t0) / 1e6:.3f} ms, all-queues-idle {idle:.1f} us in the step, gaps > 8 us: {n8}")
PY
rm -rf $O/trace
