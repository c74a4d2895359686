#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
timeout 300 python tools/exact_probe.py 384 3 509 20 2>&1 | grep -E "^\{|rror" | sed "s/^/default /"
timeout 300 python tools/exact_probe.py 384 3 509 20 CAR_F32_QKV_TILED_FROM=128 2>&1 | grep -E "^\{|rror" | sed "s/^/qkv_tiled /"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_e
( timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -- python $R/tools/exact_probe.py 384 3 509 4 > $O/r05_exact_probe_trace.log 2>&1 )
T=$(find /tmp/prof_e -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python - "$T" > $O/r05_exact_b384_3chains_timeline.txt <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"]))
rows.sort()
rows = rows[-700:-60]
t0 = rows[0][0]
for s, e, n, q in rows:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {n} g{q}")
PY
head -5 $O/r05_exact_b384_3chains_timeline.txt
