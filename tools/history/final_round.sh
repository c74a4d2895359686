#!/bin/bash
# End-of-round measurement batch on the GPU box (via gpurun): the GPU suite, smoke, the headline and exact-mode bench lines, the other BASELINE configs through
# bench.py --config N (un-profiled), then the PMC passes that tie profiles/pmc_decode_step{,_fp32}.json to THIS build of the library.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
python __graft_entry__.py smoke 2>&1 | tail -2
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4),"traffic",r["traffic"],"kernels",c["decode_kernels_per_step"], c.get("self_check"), "prefill_ms", round(c["prefill_ms"],1))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for c in 1 2 3 4 5; do timeout 300 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_config$c.json 2> $O/${TAG}_config$c.err; show $O/${TAG}_config$c.json; done
timeout 300 python bench.py --config 5 --fp8-mfma --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_config5_w8a8.json 2>/dev/null; show $O/${TAG}_config5_w8a8.json
cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1 ctr=$2; shift 2; rm -rf /tmp/pmc_$name; ( timeout 500 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -- python $R/tools/pmc_workload.py "$@" > $O/${TAG}_pmc_$name.log 2>&1 ); find /tmp/pmc_$name -name '*counter_collection.csv' | head -1; }
F=$(pmc f FETCH_SIZE 768 514 509); W=$(pmc w WRITE_SIZE 768 514 509)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_decode.py $F $W 4 768 $O/pmc_decode_step.json > $O/${TAG}_pmc_decode_b768.txt 2>&1 && python $R/tools/pmc_summary.py $F > $O/${TAG}_pmc_FETCH_SIZE_all_kernels.txt && python $R/tools/pmc_summary.py $W > $O/${TAG}_pmc_WRITE_SIZE_all_kernels.txt
F=$(pmc xf FETCH_SIZE 384 515 511 fp32); W=$(pmc xw WRITE_SIZE 384 515 511 fp32)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_decode.py $F $W 3 384 $O/pmc_decode_step_fp32.json fp32 > $O/${TAG}_pmc_decode_fp32_b384.txt 2>&1
M=$(pmc xm "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" 384 515 512 fp32)
[ -n "$M" ] && python $R/tools/pmc_mfma.py $M > $O/${TAG}_pmc_mfma_exact_decode.txt
cp $O/pmc_decode_step.json $O/pmc_decode_step_fp32.json $R/profiles/ 2>/dev/null      # so that the bench lines below can quote the traffic measured on this build
cd $R
timeout 400 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_fp32_exact_b384.json 2>/dev/null; show $O/${TAG}_bench_fp32_exact_b384.json
timeout 600 python bench.py --steps 2 --warmup 1 > $O/${TAG}_bench_b768.json 2>$O/${TAG}_bench_b768.err; show $O/${TAG}_bench_b768.json
cd /tmp; rm -rf /tmp/prof_x
( timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_x -- python $R/bench.py --precision fp32 --batch 192 --steps 1 --warmup 0 --no-cpu-baseline > $O/${TAG}_bench_fp32_b192_under_rocprof.json 2> /dev/null )
T=$(find /tmp/prof_x -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $R/tools/trace_summary.py $T > $O/${TAG}_bench_fp32_b192_trace_summary.txt && python $R/tools/trace_summary.py $T 0.5 > $O/${TAG}_bench_fp32_b192_trace_summary_decode_half.txt && head -8 $O/${TAG}_bench_fp32_b192_trace_summary_decode_half.txt | cut -c1-150
