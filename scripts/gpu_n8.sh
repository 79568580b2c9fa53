#!/bin/bash
# rehearsal of `bench.py --gpus 8` on ONE GPU: eight ranks share device 0 with gloo collectives at batch 1 per rank
# (memory: 8 full replicas of VTP-B + teacher + Adam state ~ 8 x 9 GB), so that the first real 8-GPU run cannot die on host-side
# contention (8 x mask collate / pinned staging / graph capture / rendezvous) or on the lag-2 optimizer lane's collective ordering.
# The plain command is used on purpose: bench.py re-launches itself under torch.distributed.run.  Never a reported number.
export PYTHONDONTWRITEBYTECODE=1 VTP_BENCH_BACKEND=gloo VTP_BENCH_SHARE_GPU=1
mkdir -p gpurun_out
for variant in "" "--shard-optimizer"; do
  timeout 1200 python bench.py --gpus 8 --batch 1 --steps 2 --warmup 1 --no-lpips-run --no-separate-run --no-cpu-baseline $variant > gpurun_out/bench_n8.log 2>&1
  echo "n8 [$variant] rc=$?"; tail -1 gpurun_out/bench_n8.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in ('value','n_gpus','ms_per_step','comm','loss','host_split_ms')})"
  grep -i "error\|fail\|Traceback\|ranks up" gpurun_out/bench_n8.log | head -5
done
