"""Summarise a rocprofv3 --pmc counter-collection CSV: per kernel, mean counter value per dispatch."""
import csv, sys
from collections import defaultdict
path = sys.argv[1]
d = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "").split("(")[0][-48:]
        grid = (r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        d[(name, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':50s} {'grid,wg':>18s} {'counter':>12s} {'n':>7s} {'mean':>14s} {'sum':>16s}")
for k, cs in sorted(d.items(), key=lambda x: -sum(sum(v) for v in x[1].values())):
    for cn, v in cs.items():
        print(f"{k[0]:50s} {str(k[1]):>18s} {cn:>12s} {len(v):7d} {sum(v)/len(v):14.1f} {sum(v):16.1f}")
