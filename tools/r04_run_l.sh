ROOT=$(pwd); OUT=$ROOT/gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "batch_invariant or exact or golden_cases" 2>&1 | tail -4
timeout 300 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r04l_bench_fp32_b384.json 2>$OUT/r04l.err; tail -2 $OUT/r04l.err; python - $OUT/r04l_bench_fp32_b384.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
print(round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4), "kernels", c["decode_kernels_per_step"], c.get("self_check"), "prefill_ms", c["prefill_ms"])
PY
python tools/mid_ab.py 192 1.0 512 ";CAR_PHASE_OFFSET=0;CAR_SINGLE_CHAIN=1;CAR_CHAINS=3;CAR_CHAINS=4" xl fp32 2>&1 | grep -v amdgpu.ids
