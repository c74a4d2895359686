#!/bin/bash
# kernel trace of one headline bench step (bf16, 768 images) for the round-5 evidence
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_h
( timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants > $O/r05_bench_b768_under_rocprof.json 2> /dev/null )
T=$(find /tmp/prof_h -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $R/tools/trace_summary.py $T > $O/r05_bench_b768_trace_summary.txt && python $R/tools/trace_summary.py $T 0.5 > $O/r05_bench_b768_trace_summary_decode_half.txt && head -12 $O/r05_bench_b768_trace_summary_decode_half.txt | cut -c1-150
