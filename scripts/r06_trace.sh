#!/bin/bash
# round 6: kernel trace of the replayed step (graphs + side streams): one step's rows with queue ids -> gpurun_out/r06_trace_step.csv
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
rm -rf gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run $EXTRA > $R/gpurun_out/trace_bench.log 2>&1
cd $R
tail -1 gpurun_out/trace_bench.log | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "dino_ce_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-3], marks[-2]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
with open("gpurun_out/r06_trace_step.csv", "w") as fh:
    fh.write("start_us,dur_us,queue,grid,wg,lds,vgpr,kernel\n")
    for r in sel:
        fh.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:.2f},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.2f},{r.get("Queue_Id","")},'
                 f'{r.get("Grid_Size","")},{r.get("Workgroup_Size","")},{r.get("LDS_Block_Size","")},{r.get("VGPR_Count","")},"{r["Kernel_Name"][:70]}"\n')
print(len(sel), "dispatches in the step,", (t1 - t0) / 1e6, "ms")
PY
rm -rf gpurun_out/trace
