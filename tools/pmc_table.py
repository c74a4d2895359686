"""Per-kernel table of raw PMC sums from one or more rocprofv3 `--pmc ... --output-format csv` counter_collection files:
mean per dispatch of every counter, and the ratios that read a GEMM loop (MI355X_MICROARCH.md "rocprofv3 PMC slots"):
WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES (quad-cycles, disjoint), MFMA pipe occupancy =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv, sys
from collections import defaultdict
d = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "").split("(")[0].replace("void ", "")[-60:]
            d[(name, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(d.items()):
    n = max(len(v) for v in cs.values())
    if n < 3:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(f"{k[0]} grid {k[1]} wg {k[2]} n={n}")
    print("   " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(m.items())))
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("   of WAVE_CYCLES: " + "  ".join(f"{c[3:]}={m[c] / wc:.3f}" for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if c in m))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        print(f"   MFMA pipe occupancy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.1f} %   (GRBM_GUI_ACTIVE/8 = {m['GRBM_GUI_ACTIVE'] / 8:.0f} cycles)")
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS bank-conflict cycles / LDS active cycles = {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
