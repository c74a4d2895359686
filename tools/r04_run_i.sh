ROOT=$(pwd); OUT=$ROOT/gpurun_out
for r in 2 8 64 128; do timeout 120 experiments/lat_probe $r 631 > $OUT/r04_lat_probe_v4_rows$r.txt 2>&1; cat $OUT/r04_lat_probe_v4_rows$r.txt | cut -c1-150 | head -9; done
timeout 600 python -m pytest tests -m gpu -q -x -k "not xl and not XL" 2>&1 | tail -3
for c in 2 5 3 1; do timeout 300 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r04i_config$c.json 2> $OUT/r04i_config$c.err; python - "$OUT/r04i_config$c.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(sys.argv[1][-20:], round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4),"kernels",c["decode_kernels_per_step"], c.get("self_check"))
except Exception as e: print("FAILED", e)
PY
done
