#!/bin/bash
# PMC read of the fp32 decode GEMM variants (experiments/f32_check one ...): where do the wave cycles go?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1)); rm -rf /tmp/pp_$i
  timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pp_$i -- $R/experiments/f32_check one $c 30 > /tmp/pp_$i.log 2>&1
  F=$(find /tmp/pp_$i -name '*counter_collection.csv' | head -1)
  echo "=== $c"; tail -1 /tmp/pp_$i.log
  [ -n "$F" ] && python $R/tools/pmc_table.py $F | grep -A4 dec_gemm
done
