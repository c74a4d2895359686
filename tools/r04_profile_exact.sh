#!/bin/bash
# Round-4 profiling of the EXACT mode on the GPU box (via gpurun): kernel trace of `bench.py --precision fp32` (384 images, bf16 VQ decoder) and PMC
# FETCH_SIZE / WRITE_SIZE passes over 3 decode steps at the mean position.  Raw traces stay on the box; summaries -> gpurun_out/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x /tmp/pmc_xf /tmp/pmc_xw
( timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $ROOT/bench.py --precision fp32 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/r04_bench_fp32_under_rocprof.json 2> $OUT/r04_bench_fp32_trace.err )
T=$(find /tmp/prof_x -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $ROOT/tools/trace_summary.py $T > $OUT/r04_bench_fp32_b384_trace_summary.txt && python $ROOT/tools/trace_summary.py $T 0.5 > $OUT/r04_bench_fp32_b384_trace_summary_decode_half.txt
( timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_xf -- python $ROOT/tools/pmc_workload.py 384 515 511 fp32 > $OUT/r04_pmc_xf.log 2>&1 )
( timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_xw -- python $ROOT/tools/pmc_workload.py 384 515 511 fp32 > $OUT/r04_pmc_xw.log 2>&1 )
F=$(find /tmp/pmc_xf -name '*counter_collection.csv' | head -1); W=$(find /tmp/pmc_xw -name '*counter_collection.csv' | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  python $ROOT/tools/pmc_decode.py $F $W 3 384 $OUT/pmc_decode_step_fp32.json fp32 > $OUT/r04_pmc_decode_fp32_b384.txt 2>&1
fi
tail -2 $OUT/r04_pmc_xf.log $OUT/r04_pmc_xw.log; tail -c 900 $OUT/r04_bench_fp32_under_rocprof.json; head -24 $OUT/r04_bench_fp32_b384_trace_summary_decode_half.txt; tail -6 $OUT/r04_pmc_decode_fp32_b384.txt
