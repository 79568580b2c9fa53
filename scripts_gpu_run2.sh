#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" > gpurun_out/rc.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
echo "prof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/rc.log
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_r01 | head -20
cat gpurun_out/rc.log
tail -4 gpurun_out/t_kernels.log
tail -2 gpurun_out/bench_prof.log
