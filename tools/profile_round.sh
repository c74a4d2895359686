#!/bin/bash
# Round profiling pass on the GPU box (run from the repo root via gpurun; TAG = round prefix of the output files, default r04):
#   1. kernel trace of the headline bench (bf16, 768 images) + PMC FETCH_SIZE / WRITE_SIZE passes over 4 decode steps at the mean position
#   2. the exact mode: kernel trace of `bench.py --precision fp32 --batch 192` (rocprofv3 crashes on the 384-image run: 162 GB of KV beside its buffers), PMC
#      FETCH_SIZE / WRITE_SIZE at 384 sequences, and the MFMA-pipe occupancy of its fp32 MFMA linears
#   3. kernel traces of the other BASELINE configs through bench.py --config N
# Raw traces stay on the box; summaries -> gpurun_out/ (copy what is to be judged into profiles/).  PMC passes are separate rocprofv3 runs without trace domains.
set -u
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trace() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  ( timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python $ROOT/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-variants > $OUT/${TAG}_${name}_under_rocprof.json 2> $OUT/${TAG}_${name}_trace.err )
  local T=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
  [ -n "$T" ] && python $ROOT/tools/trace_summary.py $T > $OUT/${TAG}_${name}_trace_summary.txt && python $ROOT/tools/trace_summary.py $T 0.5 > $OUT/${TAG}_${name}_trace_summary_decode_half.txt
  echo "== $name"; head -12 $OUT/${TAG}_${name}_trace_summary_decode_half.txt | cut -c1-150
}
pmc() {     # name, counter list, workload args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  ( timeout 500 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -- python $ROOT/tools/pmc_workload.py "$@" > $OUT/${TAG}_pmc_$name.log 2>&1 )
  find /tmp/pmc_$name -name '*counter_collection.csv' | head -1
}
# ---- 1. headline
trace bench_b768
F=$(pmc f FETCH_SIZE 768 514 509); W=$(pmc w WRITE_SIZE 768 514 509)
if [ -n "$F" ] && [ -n "$W" ]; then
  python $ROOT/tools/pmc_decode.py $F $W 4 768 $OUT/pmc_decode_step.json > $OUT/${TAG}_pmc_decode_b768.txt 2>&1
  python $ROOT/tools/pmc_summary.py $F > $OUT/${TAG}_pmc_FETCH_SIZE_all_kernels.txt; python $ROOT/tools/pmc_summary.py $W > $OUT/${TAG}_pmc_WRITE_SIZE_all_kernels.txt
fi
tail -3 $OUT/${TAG}_pmc_decode_b768.txt | cut -c1-300
# ---- 2. exact mode
trace bench_fp32_b192 --precision fp32 --batch 192
F=$(pmc xf FETCH_SIZE 384 515 511 fp32); W=$(pmc xw WRITE_SIZE 384 515 511 fp32)
if [ -n "$F" ] && [ -n "$W" ]; then python $ROOT/tools/pmc_decode.py $F $W 3 384 $OUT/pmc_decode_step_fp32.json fp32 > $OUT/${TAG}_pmc_decode_fp32_b384.txt 2>&1; fi
tail -3 $OUT/${TAG}_pmc_decode_fp32_b384.txt | cut -c1-300
M=$(pmc xm "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" 384 515 512 fp32)
[ -n "$M" ] && python $ROOT/tools/pmc_mfma.py $M > $OUT/${TAG}_pmc_mfma_exact_decode.txt && head -12 $OUT/${TAG}_pmc_mfma_exact_decode.txt | cut -c1-170
# ---- 3. the other BASELINE configs
for c in 2 3 5 1 4; do trace config$c --config $c; done
