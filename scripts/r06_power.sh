#!/bin/bash
# round 6: is the step power-limited?  rocm-smi samples (power, sclk) every 50 ms while the default bench command replays the step
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r06_power; mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 > $O/idle.txt
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.05; done > $O/samples.txt ) &
SP=$!
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-lpips-run --no-separate-run > $O/bench.json 2> $O/bench.err
kill $SP 2>/dev/null
python - <<'PY'
import re
rows = [l for l in open("gpurun_out/r06_power/samples.txt") if "Power" in l]
pw = [float(m.group(1)) for l in rows for m in [re.search(r"Power \(W\): ([0-9.]+)", l)] if m]
ck = [float(m.group(1)) for l in rows for m in [re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)] if m]
print("samples", len(pw), "power W: min %.0f median %.0f max %.0f" % (min(pw), sorted(pw)[len(pw)//2], max(pw)) if pw else "no power")
print("sclk MHz: min %.0f median %.0f max %.0f" % (min(ck), sorted(ck)[len(ck)//2], max(ck)) if ck else "no clock")
PY
head -30 $O/idle.txt; tail -5 $O/samples.txt; cut -c1-200 $O/bench.json
