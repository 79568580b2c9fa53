#!/bin/bash
# calibration of the SQ-counter ratios of tools/pmc_summarize.py: ONE 8192 x 8192 x 4096 bf16 GEMM (known MFMA count) per tile
# configuration in CFGS (default "8 10") under the SQ counter set of scripts/gpu_pmc.sh -> gpurun_out/r04_pmc_sq_cal[_cfgN].json
# (SQ_VALU_MFMA_BUSY_CYCLES must equal 32 x 2^24 per launch; GRBM_GUI_ACTIVE / 8 / kernel time = the sustained clock;
#  MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs))
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
mkdir -p gpurun_out
for c in ${CFGS:-8 10}; do
  rm -rf gpurun_out/pmc_sq_cal
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_cal -o pmc -- python $R/tools/one_gemm.py nt ${SHAPE:-8192 8192 4096} $c > $R/gpurun_out/pmc_sq_cal.log 2>&1
  cd $R
  out=gpurun_out/r04_pmc_sq_cal.json; [ "$c" != 8 ] && out=gpurun_out/r04_pmc_sq_cal_cfg$c.json
  python tools/pmc_summarize.py gpurun_out/pmc_sq_cal $out | head -3
  python - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if k.startswith("gemm") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        gui = v["GRBM_GUI_ACTIVE"]["per_dispatch"] / 8
        print(k, "cycles/XCD %.0f" % gui, "mfma_util %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"]["per_dispatch"] / (gui * 1024)),
              "lds_active/CU %.3f" % (v["SQ_LDS_IDX_ACTIVE"]["per_dispatch"] / 256 / gui), "valu insts", v["SQ_INSTS_VALU"]["per_dispatch"])
PY
done
