#!/bin/bash
# round 5, GPU call 2: parity at the benchmarked geometry (fixed centres) + multi-seed test, 8-rank gloo rehearsal, lane stream priority,
# kernel trace of the graph-mode step with the optimizer lane on
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05c2
mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_bench_gpu.py -q -m gpu -s > $O/tests_parity_bench.log 2>&1
echo "parity_bench rc=$?" | tee -a $O/summary.txt
grep -E "PARITY .*(ALL|POOLED|loss|multi-seed:|oracle fp32)|passed|failed|Error" $O/tests_parity_bench.log | tail -24 | tee -a $O/summary.txt
bash scripts/gpu_n8.sh 2>&1 | tee -a $O/summary.txt
ab() {
  for rep in 1 2; do
    for cfg in "$@"; do
      v=$(env $cfg timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run $EXTRA 2>$O/ab.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], "host", d.get("host_enqueue_ms_per_step"), d.get("host_split_ms"))')
      echo "[$cfg $EXTRA] $v" | tee -a $O/summary.txt
    done
  done
}
ab "VTP_LANE_PRIO=0" "VTP_LANE_PRIO=1" "VTP_LANE_PRIO=-1"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_lane -o lane -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-lpips-run --no-separate-run > $GRAFT_REPO_ROOT/$O/prof_lane.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la $O/prof_lane* | head; find $O/prof_lane -name "*kernel_trace.csv" | head -2
# keep the pulled files small: the trace of the last step only (by start timestamp) + the stats
python - <<'PY'
import csv, glob, os
fs = glob.glob("gpurun_out/r05c2/prof_lane/**/*kernel_trace.csv", recursive=True)
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-1500:]
    t0 = int(last[0]["Start_Timestamp"])
    with open("gpurun_out/r05c2/lane_trace_tail.csv", "w") as fh:
        fh.write("start_us,dur_us,queue,kernel\n")
        for r in last:
            fh.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f},{r.get("Queue_Id", "")},{r["Kernel_Name"][:60]}\n')
    os.remove(fs[0])
PY
