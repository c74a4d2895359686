set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; TAG=r06; name=bench_b768
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
( timeout 700 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants > $OUT/${TAG}_${name}_under_rocprof.json 2> $OUT/${TAG}_${name}_trace.err )
T=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $ROOT/tools/trace_summary.py $T > $OUT/${TAG}_${name}_trace_summary.txt && python $ROOT/tools/trace_summary.py $T 0.5 > $OUT/${TAG}_${name}_trace_summary_decode_half.txt
head -14 $OUT/${TAG}_${name}_trace_summary_decode_half.txt | cut -c1-150
