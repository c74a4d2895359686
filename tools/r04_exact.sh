#!/bin/bash
# Round 4 on the GPU box: GPU test suite, then the exact-mode (bit-identical tokens) bench at 384 images with the bf16 VQ decoder, and at 192.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r04b}
timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest_gpu.txt 2>&1; tail -5 $O/${TAG}_pytest_gpu.txt
timeout 400 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_fp32_exact_b384.json 2> $O/${TAG}_bench_fp32_b384.err; tail -3 $O/${TAG}_bench_fp32_b384.err
python - "$O/${TAG}_bench_fp32_exact_b384.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4),"kernels",c["decode_kernels_per_step"], c.get("self_check"), "prefill_ms", c["prefill_ms"], "decode_fraction", c["decode_fraction_of_step"])
except Exception as e: print("FAILED", e)
PY
