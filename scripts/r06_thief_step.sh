#!/bin/bash
# round 6: the step beside a kernel that holds CUs from ANOTHER process (RCCL-footprint proxy): static vs dynamic tile assignment, CU cap
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/${OUT:-r06thief}; mkdir -p $O
run() {  # $1 = CUs held (0: no thief), rest = env
  n=$1; shift
  if [ "$n" != "0" ]; then VTP_DIAG=1 python tools/cu_thief.py --serve $n 70 > $O/thief_$n.log 2>&1 & TP=$!; sleep 12; fi
  v=$(env "$@" timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-lpips-run --no-separate-run 2>$O/err.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  echo "[held=$n $*] $v" | tee -a $O/summary.txt
  if [ "$n" != "0" ]; then wait $TP; fi
}
for cfg in "$@"; do run $cfg; done
