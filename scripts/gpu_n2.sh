#!/bin/bash
# rehearsal of bench.py's multi-rank control flow on ONE GPU (two ranks share device 0, gloo collectives): all-reduce path and
# the sharded-optimizer path.  The PLAIN command is used on purpose: bench.py re-launches itself under torch.distributed.run
export PYTHONDONTWRITEBYTECODE=1 VTP_BENCH_BACKEND=gloo VTP_BENCH_SHARE_GPU=1
mkdir -p gpurun_out
for variant in "" "--shard-optimizer" "--shard-optimizer --grad-dtype bf16"; do
  timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-lpips-run --no-separate-run $variant > gpurun_out/bench_n2.log 2>&1
  echo "n2 [$variant] rc=$?"; tail -1 gpurun_out/bench_n2.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in ('value','n_gpus','ms_per_step','comm','loss')})"
  grep -i "error\|fail\|Traceback\|ranks up" gpurun_out/bench_n2.log | head -5
done
