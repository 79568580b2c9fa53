"""GPU idle time inside the replayed step: reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv: Start_Timestamp / End_Timestamp per
dispatch), takes the union of the kernel intervals over all streams, and reports busy / idle time and the distribution of the gaps
between consecutive busy intervals for the steady-state part of the run (the last `--steps` steps, found from the period of a marker
kernel that runs once per step).
Usage: python tools/trace_gaps.py gpurun_out/trace/<host>/<pid>_kernel_trace.csv [--marker dino_ce_kernel] [--per-step 1]"""
import argparse
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="dino_ce_kernel")
    ap.add_argument("--per-step", type=int, default=1, help="dispatches of the marker kernel per step")
    ap.add_argument("--skip", type=int, default=2, help="steps to drop at the front of the marker list")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [s for s, e, n in rows if a.marker in n][:: a.per_step]
    if len(marks) < a.skip + 2:
        sys.exit(f"marker {a.marker}: only {len(marks)} dispatches")
    t0, t1 = marks[a.skip], marks[-1]
    nsteps = len(marks) - 1 - a.skip
    win = [(s, e, n) for s, e, n in rows if s >= t0 and s < t1]
    busy, gaps, cur_s, cur_e, last, detail = 0, [], None, None, None, []
    for s, e, n in win:
        if cur_e is None:
            cur_s, cur_e, last = s, e, n
        elif s <= cur_e:
            if e > cur_e:
                cur_e, last = e, n
        else:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            detail.append((s - cur_e, (cur_e - t0) / 1e6, last, n))
            cur_s, cur_e, last = s, e, n
    busy += min(cur_e, t1) - cur_s
    wall = t1 - t0
    ksum = sum(e - s for s, e, n in win)
    print(f"steps {nsteps}  wall/step {wall / nsteps / 1e6:.3f} ms  busy(union)/step {busy / nsteps / 1e6:.3f} ms  idle/step "
          f"{(wall - busy) / nsteps / 1e6:.3f} ms ({100 * (wall - busy) / wall:.1f} %)  sum of kernel durations/step {ksum / nsteps / 1e6:.3f} ms  "
          f"dispatches/step {len(win) / nsteps:.0f}")
    g = sorted(x for x, _ in gaps)
    if g:
        q = lambda p: g[min(len(g) - 1, int(p * len(g)))] / 1e3
        print(f"gaps/step {len(g) / nsteps:.0f}: median {q(0.5):.1f} us, p90 {q(0.9):.1f}, p99 {q(0.99):.1f}, max {g[-1] / 1e3:.1f}; "
              f"gaps > 10 us: {sum(1 for x in g if x > 10000) / nsteps:.1f}/step = {sum(x for x in g if x > 10000) / nsteps / 1e6:.3f} ms/step")
        big = {}
        for x, n in gaps:
            if x > 10000:
                k = n[:70]
                big[k] = big.get(k, 0) + x
        for k, v in sorted(big.items(), key=lambda kv: -kv[1])[:12]:
            print(f"   idle before {k:70s} {v / nsteps / 1e3:8.1f} us/step")
        print("largest gaps (us, at ms since window start, kernel that ended last -> kernel that starts):")
        for x, at, prev, nxt in sorted(detail, reverse=True)[:40]:
            print(f"   {x / 1e3:8.1f} @ {at:9.3f}  {prev[:60]:60s} -> {nxt[:60]}")


    return detail, nsteps, wall


if __name__ == "__main__":
    main()
