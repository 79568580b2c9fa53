#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/${OUT:-r06super}; mkdir -p $O
VTP_GEMM_SUPERTILE=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm8p_gpu.py tests/test_gemm8h_gpu.py tests/test_gemm4w_gpu.py tests/test_gemm_dyn_gpu.py -x -q -m gpu > $O/tests_forced.log 2>&1
echo "tests (supertile forced on) rc=$?" | tee -a $O/summary.txt; tail -2 $O/tests_forced.log | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm8p_gpu.py tests/test_gemm8h_gpu.py tests/test_gemm4w_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests (default) rc=$?" | tee -a $O/summary.txt; tail -2 $O/tests.log | tee -a $O/summary.txt
for r in 1 2; do for e in 0 1; do VTP_GEMM_SUPERTILE=$e python tools/gemm_shapes.py st$e 2>&1 | grep -v amdgpu.ids >> $O/shapes.log; done; done
python - <<'PY' | tee -a gpurun_out/${OUT:-r06super}/summary.txt
import re, collections, os
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(f"gpurun_out/{os.environ.get('OUT', 'r06super')}/shapes.log"):
    m = re.match(r"\[(\w+)\] (.{22}) M=.*?(\d+\.\d) us", l)
    if m: d[m.group(2)][m.group(1)].append(float(m.group(3)))
for k, v in d.items():
    a, b = sum(v["st0"]) / len(v["st0"]), sum(v["st1"]) / len(v["st1"])
    print(f"{k} row-major {a:7.1f} supertile {b:7.1f}  x{a / b:.3f}")
PY
REPS=2 bash scripts/r06_ab.sh "VTP_GEMM_SUPERTILE=0" "VTP_GEMM_SUPERTILE=1" "VTP_GEMM_SUPERTILE=-"
