#!/bin/bash
set -u
O=gpurun_out/r06_b; mkdir -p $O
timeout 300 experiments/kbench check > $O/kbench_check.txt 2>&1; tail -1 $O/kbench_check.txt
for a in "2 631" "2 631 0 1 12 2 192 160" "8 631 0 1 12 2 96 160" "8 631" "16 631" "64 631"; do n=$(echo $a | tr ' ' '_'); timeout 60 experiments/lat_probe $a > $O/lat_$n.txt 2>&1; done
grep -H "instrumented chain" $O/lat_*.txt
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
for c in 2 5 4; do timeout 400 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $O/r06_config$c.json 2> $O/config$c.err; python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/r06_config$c.json") if l.startswith("{")][-1])
    print("config $c:", round(d["value"], 4), "img/s", round(d["roofline"]["avg_launch_ms"], 4), "ms/token frac", round(d["roofline"]["frac"], 4))
except Exception as e:
    print("config $c failed", e)
PY
done
