#!/bin/bash
# round 6, pass A: the GPU suite on the new small-chain schedule, then BASELINE configs 2 / 5 / 4 / 1 / 3 through bench.py
set -u
O=gpurun_out/r06_a; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "runahead or knobs or small" > $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
for c in 2 5 4 1 3; do timeout 400 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $O/r06_config$c.json 2> $O/config$c.err; python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/r06_config$c.json") if l.startswith("{")][-1])
    print("config $c:", round(d["value"], 4), "img/s", round(d["roofline"]["avg_launch_ms"], 4), "ms/token frac", round(d["roofline"]["frac"], 4))
except Exception as e:
    print("config $c failed", e)
PY
done
