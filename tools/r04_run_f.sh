ROOT=$(pwd); OUT=$ROOT/gpurun_out
for r in 2 8 64; do timeout 120 experiments/lat_probe $r 631 > $OUT/r04_lat_probe_rows$r.txt 2>&1; cat $OUT/r04_lat_probe_rows$r.txt | cut -c1-150; done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_x
( timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_x -- python $ROOT/bench.py --precision fp32 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/r04_bench_fp32_under_rocprof.json 2> $OUT/r04_bench_fp32_trace.err )
T=$(find /tmp/prof_x -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $ROOT/tools/trace_summary.py $T > $OUT/r04_bench_fp32_b384_trace_summary.txt && python $ROOT/tools/trace_summary.py $T 0.5 > $OUT/r04_bench_fp32_b384_trace_summary_decode_half.txt
head -20 $OUT/r04_bench_fp32_b384_trace_summary_decode_half.txt | cut -c1-160; tail -3 $OUT/r04_bench_fp32_trace.err
