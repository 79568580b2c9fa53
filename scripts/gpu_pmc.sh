#!/bin/bash
# HBM traffic and SQ counters of the default bench command (eager launches, no stream overlap), separate PMC passes
export PYTHONDONTWRITEBYTECODE=1 VTP_OVERLAP=0
R=$PWD
rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
# three separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ set = 8 SQ slots + GRBM): HBM bytes, and
# matrix-pipe / VALU / LDS occupancy (MfmaUtil of north_star -- tools/pmc_summarize.py `derived`)
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p$i -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-lpips-run --no-separate-run --no-graphs > $R/gpurun_out/pmc_p$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
cd $R
python tools/pmc_summarize.py gpurun_out/pmc gpurun_out/pmc_summary.json
find gpurun_out/pmc -name "*.csv" -size +1M -delete
