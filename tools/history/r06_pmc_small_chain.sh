#!/bin/bash
# round 6: does the L2 run-ahead add HBM traffic?  FETCH_SIZE of 10 decode steps of a 2-row chain (XL, cfg 1, positions ~620), helpers on / off
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  rm -rf /tmp/pmc_sc_$mode
  if [ $mode = off ]; then export CAR_NO_RUNAHEAD=1; else unset CAR_NO_RUNAHEAD; fi
  ( timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_sc_$mode -- python $ROOT/tools/pmc_workload.py 2 511 500 > $OUT/r06_pmc_small_chain_$mode.log 2>&1 )
  F=$(find /tmp/pmc_sc_$mode -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python $ROOT/tools/pmc_summary.py $F > $OUT/r06_pmc_small_chain_FETCH_SIZE_$mode.txt
  echo "== helpers $mode"; grep -E "dec_gemm|dec_attn" $OUT/r06_pmc_small_chain_FETCH_SIZE_$mode.txt | head -8 | cut -c1-170
done
