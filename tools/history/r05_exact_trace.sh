#!/bin/bash
# whole-pass kernel trace of the exact mode at 384 sequences (encode + prefill + a few decode steps at the mean position): where the non-decode time goes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_e
( timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -- python $R/tools/exact_probe.py 384 3 509 3 > $O/r05_exact_probe_trace.log 2>&1 )
T=$(find /tmp/prof_e -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $R/tools/trace_summary.py $T 1.0 > $O/r05_exact_b384_trace_all.txt && head -45 $O/r05_exact_b384_trace_all.txt | cut -c1-150
