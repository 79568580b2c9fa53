#!/bin/bash
# calibration of the SQ-counter ratios of tools/pmc_summarize.py: ONE 8192 x 8192 x 4096 bf16 GEMM (known MFMA count) under the SQ
# counter set of scripts/gpu_pmc.sh -> gpurun_out/r04_pmc_sq_cal.json (SQ_VALU_MFMA_BUSY_CYCLES must equal 32 x 2^24 per launch;
# GRBM_GUI_ACTIVE / 8 / kernel time = the sustained clock)
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_sq_cal
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_cal -o pmc -- python $R/tools/one_gemm.py nt 8192 8192 4096 8 > $R/gpurun_out/pmc_sq_cal.log 2>&1
cd $R
python tools/pmc_summarize.py gpurun_out/pmc_sq_cal gpurun_out/r04_pmc_sq_cal.json | head -5
