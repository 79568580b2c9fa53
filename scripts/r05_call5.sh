#!/bin/bash
# round 5, GPU call 5: kernel trace of the graph-mode step (per-stream timeline of ONE step) + side-stream priority A/B
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
O=$R/gpurun_out/r05c5
mkdir -p $O
python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" | tee -a $O/summary.txt
for rep in 1 2; do
  for cfg in "VTP_SIDE_PRIO=0" "VTP_SIDE_PRIO=1" "VTP_SIDE_PRIO=-1"; do
    v=$(env $cfg timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run 2>$O/ab.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"])')
    echo "[$cfg] $v" | tee -a $O/summary.txt
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o lane -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-lpips-run --no-separate-run > $O/prof.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os
fs = glob.glob("gpurun_out/r05c5/prof/**/*kernel_trace.csv", recursive=True)
print("trace files", fs)
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    print("columns", list(rows[0].keys()))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last full step: from the last im2col16 of the teacher/student start ... take the last 1300 kernels
    last = rows[-1400:]
    t0 = int(last[0]["Start_Timestamp"])
    with open("gpurun_out/r05c5/trace_tail.csv", "w") as fh:
        fh.write("start_us,dur_us,queue,stream,kernel\n")
        for r in last:
            fh.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f},{r.get("Queue_Id", "")},{r.get("Stream_Id", "")},{r["Kernel_Name"][:70]}\n')
    for f in fs:
        os.remove(f)
PY
find $O/prof -name "*.db" -delete
ls -la $O | tee -a $O/summary.txt
