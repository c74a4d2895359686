#!/bin/bash
# Round-5 final batch B on the GPU box: PMC traffic of the decode step (bf16 768 / fp32 384) tied to THIS build, MFMA occupancy of the exact step, kernel traces, BASELINE configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4),"traffic",r["traffic"],"kernels",c["decode_kernels_per_step"], c.get("variants"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1 ctr=$2; shift 2; rm -rf /tmp/pmc_$name; ( timeout 500 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -- python $R/tools/pmc_workload.py "$@" > $O/r05_pmc_$name.log 2>&1 ); find /tmp/pmc_$name -name '*counter_collection.csv' | head -1; }
F=$(pmc xf FETCH_SIZE 384 515 511 fp32); W=$(pmc xw WRITE_SIZE 384 515 511 fp32)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_decode.py $F $W 3 384 $O/pmc_decode_step_fp32.json fp32 > $O/r05_pmc_decode_fp32_b384.txt 2>&1; tail -1 $O/r05_pmc_decode_fp32_b384.txt | cut -c1-300
M=$(pmc xm "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" 384 515 512 fp32)
[ -n "$M" ] && python $R/tools/pmc_mfma.py $M > $O/r05_pmc_mfma_exact_decode.txt; head -8 $O/r05_pmc_mfma_exact_decode.txt | cut -c1-160
F=$(pmc f FETCH_SIZE 768 514 509); W=$(pmc w WRITE_SIZE 768 514 509)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_decode.py $F $W 4 768 $O/pmc_decode_step.json > $O/r05_pmc_decode_b768.txt 2>&1; tail -1 $O/r05_pmc_decode_b768.txt | cut -c1-300
cd /tmp; rm -rf /tmp/prof_x
( timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_x -- python $R/bench.py --precision fp32 --batch 192 --steps 1 --warmup 0 --no-cpu-baseline > $O/r05_bench_fp32_b192_under_rocprof.json 2> /dev/null )
T=$(find /tmp/prof_x -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $R/tools/trace_summary.py $T > $O/r05_bench_fp32_b192_trace_summary.txt && python $R/tools/trace_summary.py $T 0.5 > $O/r05_bench_fp32_b192_trace_summary_decode_half.txt && head -9 $O/r05_bench_fp32_b192_trace_summary_decode_half.txt | cut -c1-150
cd $R
for c in 2 3 4 5 1; do timeout 300 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > $O/r05_config$c.json 2> $O/r05_config$c.err; show $O/r05_config$c.json; done
