#!/bin/bash
# Round-5 final batch A on the GPU box: the GPU suite, smoke, the default bench line (headline + the `exact` block + cpu baseline)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/parity_measured.jsonl
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 > $O/r05_bench_b768.json 2> $O/r05_bench_b768.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_b768.json"))
print("headline", round(d["value"], 3), "img/s  step", round(d["roofline"]["avg_launch_ms"], 3), "ms frac", round(d["roofline"]["frac"], 4), d["config"].get("stage_ms"), d["config"].get("vq_ms_per_image"), d["config"]["self_check"])
e = d.get("exact", {})
print("exact", e.get("value"), e.get("roofline", {}).get("avg_launch_ms"), e.get("golden_token_agreement"), e.get("error"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
tail -3 $O/r05_bench_b768.err
