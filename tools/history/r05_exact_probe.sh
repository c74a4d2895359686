#!/bin/bash
# exact-mode decode step at the mean position by chain count and attention form (dev library: CAR_* switches)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for form in 3 1; do
  timeout 600 python tools/exact_probe.py 384 1,2,3,4 509 20 CAR_ATTN_F32_FORM=$form 2>&1 | grep -E "^\{|Error|error" | sed "s/^/form$form /"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_e
( timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -- python $R/tools/exact_probe.py 384 2 509 6 > $O/r05_exact_probe_trace.log 2>&1 )
T=$(find /tmp/prof_e -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python $R/tools/trace_summary.py $T 0.3 > $O/r05_exact_b384_trace_tail.txt && head -12 $O/r05_exact_b384_trace_tail.txt | cut -c1-160
[ -n "$T" ] && python - "$T" > $O/r05_exact_b384_timeline.txt <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
rows = rows[-420:-40]
t0 = rows[0][0]
for s, e, n, q in rows:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{q:>4s}  {n}")
PY
head -30 $O/r05_exact_b384_timeline.txt
