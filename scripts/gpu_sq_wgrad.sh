#!/bin/bash
# SQ counters of ONE grouped weight-gradient launch (VTP-B block, 34 144 token rows) on the 8-phase kernel and on the one-wave-per-SIMD
# kernel -> gpurun_out/r04_pmc_sq_wgrad_k{0,1}.json (matrix-pipe utilisation of the step's dominant kernel in isolation)
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
mkdir -p gpurun_out
for k in 0 1; do
  rm -rf gpurun_out/pmc_sq_wgrad
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_wgrad -o pmc -- python $R/tools/one_wgrad_group.py $k > $R/gpurun_out/pmc_sq_wgrad.log 2>&1
  cd $R
  python tools/pmc_summarize.py gpurun_out/pmc_sq_wgrad gpurun_out/r04_pmc_sq_wgrad_k$k.json | grep grouped
done
