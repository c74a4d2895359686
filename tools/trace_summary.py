"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid): count, avg/min us, total ms, share.
Optionally restrict to the last `frac` of the trace (the steady-state decode loop)."""
import csv, sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        wx = max(int(r["Workgroup_Size_X"]), 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:],
                     int(r["Grid_Size_X"]) // wx, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), wx, r["LDS_Block_Size"], r["VGPR_Count"]))
rows.sort()
rows = rows[int(len(rows) * (1 - frac)):]
d = defaultdict(list)
for s, e, n, gx, gy, gz, wx, lds, vg in rows:
    d[(n, gx, gy, gz, wx, lds, vg)].append((e - s) / 1e3)
tot = sum(sum(v) for v in d.values())
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"kernels={len(rows)} busy={tot/1e3:.1f} ms span={span/1e3:.1f} ms")
print(f"{'kernel':50s} {'grid':>14s} {'wg':>5s} {'lds':>6s} {'vgpr':>4s} {'n':>7s} {'avg_us':>8s} {'min_us':>8s} {'tot_ms':>9s} {'share':>6s}")
for k, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    print(f"{k[0]:50s} {str(k[1:4]):>14s} {k[4]:5d} {k[5]:>6s} {k[6]:>4s} {len(v):7d} {sum(v)/len(v):8.2f} {min(v):8.2f} {sum(v)/1e3:9.2f} {100*sum(v)/tot:6.2f}")
