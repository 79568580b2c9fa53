#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" > gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-graphs --no-cpu-baseline > gpurun_out/bench_eager.log 2>&1
echo "bench eager rc=$?" >> gpurun_out/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_graph.log 2>&1
echo "bench graph rc=$?" >> gpurun_out/rc.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01b -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graphs > $R/gpurun_out/bench_prof.log 2>&1
echo "prof rc=$?" >> $R/gpurun_out/rc.log
cd $R
cat gpurun_out/rc.log
tail -4 gpurun_out/t_all.log
tail -1 gpurun_out/bench_eager.log
tail -1 gpurun_out/bench_graph.log
