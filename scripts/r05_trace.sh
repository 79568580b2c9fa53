#!/bin/bash
# kernel trace of the default bench (hipGraph segments + side streams) for tools/trace_gaps.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace
export PYTHONDONTWRITEBYTECODE=1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run > gpurun_out/trace_bench.log 2>&1
tail -1 gpurun_out/trace_bench.log | cut -c1-300
f=$(ls gpurun_out/trace/*/*_kernel_trace.csv | head -1)
ls -la $f
python tools/trace_gaps.py $f > gpurun_out/trace_gaps.txt 2>&1
cat gpurun_out/trace_gaps.txt
cp $f gpurun_out/trace_kernels.csv
rm -rf gpurun_out/trace
