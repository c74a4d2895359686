ROOT=$(pwd); OUT=$ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r04n_bench_b768.json 2>$OUT/r04n.err; tail -2 $OUT/r04n.err; python - $OUT/r04n_bench_b768.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
print(round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4), "kernels", c["decode_kernels_per_step"], c.get("self_check"), "prefill_ms", c["prefill_ms"], "input_s", c["input_distribution_s"])
PY
timeout 300 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r04n_bench_fp32_b384.json 2>/dev/null; python - $OUT/r04n_bench_fp32_b384.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
print(round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4), c.get("self_check"), "prefill_ms", c["prefill_ms"])
PY
