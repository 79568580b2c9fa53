#!/bin/bash
# which launches are the __amd_rocclr_copyBuffer / fill kernels of a step? (kernel trace, eager, one stream: grid sizes + neighbours)
export PYTHONDONTWRITEBYTECODE=1 VTP_OVERLAP=0
R=$PWD
rm -rf $R/gpurun_out/prof_cp; mkdir -p $R/gpurun_out/prof_cp
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cp -o cp -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lpips-run --no-separate-run --no-graphs > $R/gpurun_out/prof_cp.log 2>&1
cd $R
f=$(find gpurun_out/prof_cp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# last step only: find last adamw
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
step = rows[lo:hi]
print("kernels in last step", len(step))
c = collections.Counter()
for i, r in enumerate(step):
    nm = r["Kernel_Name"]
    if "copyBuffer" in nm or "FillFunctor" in nm:
        prev = step[i - 1]["Kernel_Name"][:40] if i else ""
        nxt = step[i + 1]["Kernel_Name"][:40] if i + 1 < len(step) else ""
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = ("copy" if "copy" in nm else "fill", r.get("Grid_Size_X", r.get("Workgroup_Size_X", "?")), prev, nxt)
        c[key] += 1
        c[("dur", key)] += dur
for k, v in sorted(((k, v) for k, v in c.items() if k[0] != "dur"), key=lambda kv: -c[("dur", kv[0])])[:40]:
    print(f"{v:4d} x {k[0]} grid={k[1]:>9s} total {c[('dur', k)]:8.1f} us | after {k[2]} | before {k[3]}")
PY
rm -rf gpurun_out/prof_cp
