#!/bin/bash
# Round-4 baseline traces on the GPU box (via gpurun): kernel traces of the BASELINE configs that never had one (3, 5, 1) and of the
# exact (fp32, bit-identical tokens) mode.  Raw traces stay on the box; summaries -> gpurun_out/<tag>_*.
set -u
TAG=${1:-r04a}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $ROOT/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_${name}_under_rocprof.json 2> $OUT/${TAG}_${name}.err )
  local T=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
  [ -n "$T" ] && python $ROOT/tools/trace_summary.py $T 0.5 > $OUT/${TAG}_${name}_trace_summary_decode_half.txt && python $ROOT/tools/trace_summary.py $T > $OUT/${TAG}_${name}_trace_summary.txt
  echo "== $name"; head -16 $OUT/${TAG}_${name}_trace_summary_decode_half.txt; tail -c 600 $OUT/${TAG}_${name}_under_rocprof.json
}
run config3 --config 3
run config5 --config 5
run config1 --config 1
run fp32_b64 --precision fp32 --batch 64
