#!/bin/bash
# round 4, first GPU call: the new parity tests (no -x: every result is wanted), smoke, this round's baseline bench line with the
# per-shape GEMM table, SQ counter passes (MFMA utilisation / VALU / LDS conflicts) of the step, and the plain-command N=2 rehearsal
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_large_gpu.py "tests/test_fp8_gpu.py" "tests/test_ddp_gpu.py::test_resume_vtp_ssl_training_state_world2" \
  "tests/test_kernels_gpu.py::test_attention_fwd_bwd" "tests/test_trainer_inputs_gpu.py::test_prepare_ssl_batches_keep_their_index_tensors" \
  -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/r04_newtests.log 2>&1
echo "newtests rc=$?"; grep -c PARITY gpurun_out/r04_newtests.log; tail -15 gpurun_out/r04_newtests.log | cut -c1-300
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
VTP_BENCH_GEMM_TABLE=gpurun_out/r04_gemm_table_base.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lpips-run --no-separate-run 2>gpurun_out/r04_bench_base.err | tail -1 > gpurun_out/r04_bench_base.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_base.json')); r=d['roofline']; print('BASE', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['gemm_ms_per_step'])"
head -30 gpurun_out/r04_gemm_table_base.txt
# SQ counters of the step (eager launches, single stream): one pass, 8 SQ slots + GRBM
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq; mkdir -p $R/gpurun_out/pmc_sq
VTP_OVERLAP=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/p1 -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-lpips-run --no-separate-run --no-graphs > $R/gpurun_out/pmc_sq.log 2>&1
echo "pmc sq rc=$?"
# calibration: one GEMM of known FLOPs under the same counters (normalisation of MFMA_BUSY / GUI_ACTIVE)
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/cal -o pmc -- python $R/tools/one_gemm.py nt 8192 8192 4096 8 > $R/gpurun_out/pmc_sq_cal.log 2>&1
cd $R
python tools/pmc_summarize.py gpurun_out/pmc_sq/p1 gpurun_out/r04_pmc_sq_summary.json | head -20
python tools/pmc_summarize.py gpurun_out/pmc_sq/cal gpurun_out/r04_pmc_sq_cal.json | head -5
find gpurun_out/pmc_sq -name "*.csv" -size +1M -delete
# N = 2 rehearsal from the PLAIN command (bench.py re-launches itself), first variant only
export VTP_BENCH_BACKEND=gloo VTP_BENCH_SHARE_GPU=1
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-lpips-run --no-separate-run > gpurun_out/r04_bench_n2.log 2>&1
echo "n2 rc=$?"; tail -1 gpurun_out/r04_bench_n2.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in ('value','n_gpus','ms_per_step','comm','loss')})"
grep -i "re-running\|ranks up\|Traceback" gpurun_out/r04_bench_n2.log | head -5
