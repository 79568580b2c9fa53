#!/bin/bash
# round 6: where the text-tower backward sits relative to the decoder backward in a TRACED step -- NOT evidence of what the untraced replay
# does: rocprofv3 serialises kernel chains on different queues (profiles/r06_text_backward_trace_artifact.txt).  One traced step per setting; prints the
# start of the first / last causal attention backward (text) and of the decoder's first / last fused attention backward
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
for cfg in "$@"; do
  O=gpurun_out/r06to_$(echo "$cfg" | tr ' =' '__'); mkdir -p $O; rm -rf $O/trace
  (cd /tmp && export TMPDIR=/tmp && env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-lpips-run --no-separate-run $EXTRA > $R/$O/trace_bench.log 2>&1)
  python - "$O" "$cfg" <<'PY'
import csv, glob, sys
O, cfg = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "dino_ce_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-3], marks[-2]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
txt = [(int(r["Start_Timestamp"]) - t0) / 1e3 for r in sel if "attn_bwd_dq_kernel<true>" in r["Kernel_Name"]]
dec = [(int(r["Start_Timestamp"]) - t0) / 1e3 for r in sel if "attn_bwd_fused_kernel<8>" in r["Kernel_Name"]][:12]
print(f"[{cfg}] step {(t1 - t0) / 1e6:.3f} ms; text attention backward {txt[0]:.0f} .. {txt[-1]:.0f} us, decoder attention backward {dec[0]:.0f} .. {dec[-1]:.0f} us")
PY
  rm -rf $O/trace
done
