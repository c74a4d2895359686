set -u
TAG=r04
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1 ctr=$2; shift 2; rm -rf /tmp/pmc_$name; ( timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -- python $R/tools/pmc_workload.py "$@" > $O/${TAG}_pmc_$name.log 2>&1 ); find /tmp/pmc_$name -name '*counter_collection.csv' | head -1; }
F=$(pmc f FETCH_SIZE 768 514 509); W=$(pmc w WRITE_SIZE 768 514 509)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_decode.py $F $W 4 768 $O/pmc_decode_step.json > $O/${TAG}_pmc_decode_b768.txt 2>&1 && python $R/tools/pmc_summary.py $F > $O/${TAG}_pmc_FETCH_SIZE_all_kernels.txt && python $R/tools/pmc_summary.py $W > $O/${TAG}_pmc_WRITE_SIZE_all_kernels.txt
F=$(pmc xf FETCH_SIZE 384 515 511 fp32); W=$(pmc xw WRITE_SIZE 384 515 511 fp32)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_decode.py $F $W 3 384 $O/pmc_decode_step_fp32.json fp32 > $O/${TAG}_pmc_decode_fp32_b384.txt 2>&1
cp $O/pmc_decode_step.json $O/pmc_decode_step_fp32.json $R/profiles/ 2>/dev/null
cd $R
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"],4),"img/s", "ms/step(decode)",round(r["avg_launch_ms"],4),"frac",round(r["frac"],4),"traffic",r["traffic"], c.get("self_check"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 300 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_fp32_exact_b384.json 2>/dev/null; show $O/${TAG}_bench_fp32_exact_b384.json
timeout 400 python bench.py --steps 2 --warmup 1 > $O/${TAG}_bench_b768.json 2>$O/${TAG}_bench_b768.err; show $O/${TAG}_bench_b768.json
