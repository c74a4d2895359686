#!/bin/bash
# the default bench line once more, now that profiles/pmc_decode_step{,_fp32}.json carry THIS build's id (roofline.traffic quoted)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
timeout 900 python bench.py --steps 2 --warmup 1 > $O/r05_bench_b768.json 2> $O/r05_bench_b768.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_b768.json"))
print("headline", round(d["value"], 3), "img/s  step", round(d["roofline"]["avg_launch_ms"], 3), "ms frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], d["config"].get("stage_ms"))
e = d.get("exact", {})
print("exact", e.get("value"), e.get("roofline", {}).get("avg_launch_ms"), e.get("roofline", {}).get("traffic"), e.get("golden_token_agreement"), e.get("error"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
