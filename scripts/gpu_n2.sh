#!/bin/bash
# rehearsal of bench.py's multi-rank control flow on ONE GPU (two ranks share device 0, gloo collectives)
export PYTHONDONTWRITEBYTECODE=1 VTP_BENCH_BACKEND=gloo VTP_BENCH_SHARE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_n2.log 2>&1
echo "n2 rc=$?"; tail -1 gpurun_out/bench_n2.log | cut -c1-400; grep -i "error\|fail\|Traceback" gpurun_out/bench_n2.log | head -5
