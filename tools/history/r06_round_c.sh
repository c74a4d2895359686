#!/bin/bash
set -u
O=gpurun_out/r06_c; mkdir -p $O
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $O/r06_bench_b768_quick.json 2> $O/bench.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r06_bench_b768_quick.json") if l.startswith("{")][-1])
print("headline:", round(d["value"], 3), "img/s; decode step", round(d["roofline"]["avg_launch_ms"], 4), "ms; frac", round(d["roofline"]["frac"], 4), d["config"].get("self_check"))
PY
timeout 400 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $O/r06_config3.json 2> $O/config3.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r06_config3.json") if l.startswith("{")][-1])
print("config 3:", round(d["value"], 3), "img/s;", round(d["roofline"]["avg_launch_ms"], 4), "ms/token; frac", round(d["roofline"]["frac"], 4))
PY
