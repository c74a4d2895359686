"""MFMA-pipe occupancy per kernel from one rocprofv3 pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`.
MfmaUtil (gfx94x derived-counter formula; ROCm 7.2 has no gfx950 section, MI355X_MICROARCH.md) =
100 * SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs): rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs
(checked against kernel durations).  Printed next to the raw sums so the reader can re-derive it."""
import csv, sys
from collections import defaultdict
path = sys.argv[1]
d = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "").split("(")[0].replace("void ", "")[-44:]
        d[(name, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in d.items():
    m, g = sum(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), sum(cs.get("GRBM_GUI_ACTIVE", [0]))
    if m <= 0 or g <= 0:
        continue
    rows.append((g, k, len(cs["GRBM_GUI_ACTIVE"]), m, sum(cs.get("SQ_BUSY_CYCLES", [0]))))
print(f"{'kernel':46s} {'grid':>10s} {'wg':>5s} {'n':>5s} {'GRBM_GUI_ACTIVE':>16s} {'MFMA_BUSY_CYCLES':>17s} {'SQ_BUSY_CYCLES':>15s} {'MfmaUtil%':>9s}")
for g, k, n, m, sq in sorted(rows, reverse=True)[:40]:
    print(f"{k[0]:46s} {k[1]:>10s} {k[2]:>5s} {n:5d} {g:16.0f} {m:17.0f} {sq:15.0f} {100.0 * m / (g / 8 * 256 * 4):9.1f}")
