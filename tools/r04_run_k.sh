ROOT=$(pwd); OUT=$ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "goldens_at_model_size" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_x
( timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_x -- python $ROOT/bench.py --precision fp32 --batch 192 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/r04_bench_fp32_b192_under_rocprof.json 2> $OUT/r04_bench_fp32_trace.err )
T=$(find /tmp/prof_x -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $ROOT/tools/trace_summary.py $T > $OUT/r04_bench_fp32_b192_trace_summary.txt && python $ROOT/tools/trace_summary.py $T 0.5 > $OUT/r04_bench_fp32_b192_trace_summary_decode_half.txt
head -16 $OUT/r04_bench_fp32_b192_trace_summary_decode_half.txt | cut -c1-160; tail -2 $OUT/r04_bench_fp32_trace.err
cd $ROOT; timeout 300 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r04k_bench_fp32_b384.json 2>/dev/null; python - $OUT/r04k_bench_fp32_b384.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
print(round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4), "traffic", r["traffic"], c.get("self_check"), "prefill_ms", c["prefill_ms"])
PY
