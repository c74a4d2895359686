#!/bin/bash
# Round-3 profiling pass on the GPU box (run from the repo root via gpurun): kernel trace of the headline bench + PMC
# FETCH_SIZE / WRITE_SIZE passes over a few decode steps at the mean position.  Raw traces stay on the box; summaries -> gpurun_out/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_trace /tmp/pmc_f /tmp/pmc_w
( timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/r03_trace_bench.json 2> $OUT/r03_trace_bench.err )
T=$(find /tmp/prof_trace -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $ROOT/tools/trace_summary.py $T > $OUT/r03_bench_b768_trace_summary.txt && python $ROOT/tools/trace_summary.py $T 0.5 > $OUT/r03_bench_b768_trace_summary_decode_half.txt
S=$(find /tmp/prof_trace -name '*kernel_stats.csv' | head -1); [ -n "$S" ] && cp $S $OUT/r03_bench_b768_kernel_stats.csv
( timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $ROOT/tools/pmc_workload.py 768 514 509 > $OUT/r03_pmc_f.log 2>&1 )
( timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $ROOT/tools/pmc_workload.py 768 514 509 > $OUT/r03_pmc_w.log 2>&1 )
F=$(find /tmp/pmc_f -name '*counter_collection.csv' | head -1); W=$(find /tmp/pmc_w -name '*counter_collection.csv' | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  python $ROOT/tools/pmc_decode.py $F $W 4 768 $OUT/pmc_decode_step.json > $OUT/r03_pmc_decode_b768.txt 2>&1
  python $ROOT/tools/pmc_summary.py $F > $OUT/r03_pmc_FETCH_SIZE_all_kernels.txt; python $ROOT/tools/pmc_summary.py $W > $OUT/r03_pmc_WRITE_SIZE_all_kernels.txt
fi
tail -2 $OUT/r03_pmc_f.log $OUT/r03_pmc_w.log; cat $OUT/r03_trace_bench.json; head -30 $OUT/r03_bench_b768_trace_summary_decode_half.txt; tail -5 $OUT/r03_pmc_decode_b768.txt
# small-batch regime (BASELINE config 2: cfg 4, batch 1 = 2 rows): where the 1.4 ms of a decode step go, kernel by kernel
rm -rf /tmp/prof_c2
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $ROOT/bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/r03_trace_config2.json 2> $OUT/r03_trace_config2.err )
T2=$(find /tmp/prof_c2 -name '*kernel_trace.csv' | head -1)
[ -n "$T2" ] && python $ROOT/tools/trace_summary.py $T2 > $OUT/r03_config2_trace_summary.txt && head -24 $OUT/r03_config2_trace_summary.txt
