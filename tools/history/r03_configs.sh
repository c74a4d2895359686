#!/bin/bash
# Round-3 measurement batch on the GPU box (via gpurun): exact-mode bench + the other BASELINE configs through bench.py --config N.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python bench.py --precision fp32 --batch 192 --steps 1 --warmup 1 --no-cpu-baseline > $O/r03_bench_fp32_exact_b192.json 2> $O/r03_bench_fp32.err
for c in 2 3 4 5; do timeout 300 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > $O/r03_config$c.json 2> $O/r03_config$c.err; done
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --weights-fp8 --condition-type hed --adapter-size base > $O/r03_config5_fp8_weight_only_b8.json 2>/dev/null
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --condition-type hed --adapter-size base > $O/r03_config5_bf16_twin_b8.json 2>/dev/null
for f in $O/r03_bench_fp32_exact_b192.json $O/r03_config*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4),"kernels",c["decode_kernels_per_step"], c.get("self_check"))
except Exception as e: print("FAILED", e)
PY
done
