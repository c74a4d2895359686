"""Assemble the HBM traffic of ONE decode step from rocprofv3 --pmc passes over tools/pmc_workload.py.

inputs : FETCH_SIZE csv, WRITE_SIZE csv (rocprofv3 --pmc <counter> --output-format csv), decode steps profiled, batch
output : per-kernel means (text) + profiles/pmc_decode_step.json {fetch_bytes_per_step, write_bytes_per_step, ...}
Units  : FETCH_SIZE / WRITE_SIZE are reported in KiB.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads -> doubled here; WRITE_SIZE is left as reported (uncalibrated in the guide) and
cross-checked against the bytes the decode kernels are known to write.
Only kernels of the decode loop are summed (dec_gemm / dec_attn2 / rmsnorm2 / sample / advance), divided by the number of
profiled steps."""
import csv, json, os, sys
from collections import defaultdict

fetch_csv, write_csv, steps, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
precision = sys.argv[6] if len(sys.argv) > 6 else "bf16"
DEC = ("dec_gemm_kernel", "dec_attn2", "rmsnorm2_kernel", "sample_greedy_kernel", "sample_stochastic_kernel", "advance_kernel",
       "dec_gemm_f32_kernel", "dec_gemm_f32t_kernel", "dec_attn_f32")      # exact mode (decode_f32.hip); its rmsnorm_kernel<float> is shared with the prefill and left out (< 0.5 % of the step's bytes)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from controlar_amd import _lib
build_id = _lib.load().car_build_id().decode()


def load(path, counter):
    d = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if not any(k in name for k in DEC):
                continue
            short = name.split("(")[0].replace("void ", "")[:70]
            d[(short, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))].append(float(r["Counter_Value"]))
    return d


res = {}
for tag, path, counter, mult in (("fetch", fetch_csv, "FETCH_SIZE", 2.0), ("write", write_csv, "WRITE_SIZE", 1.0)):
    d = load(path, counter)
    tot = 0.0
    print(f"--- {counter} (KiB per dispatch as reported; x{mult} applied in the per-step sum)")
    for k, v in sorted(d.items(), key=lambda x: -sum(x[1])):
        print(f"{k[0]:72s} grid {k[1]:>8s} wg {k[2]:>4s} n {len(v):5d} mean {sum(v)/len(v):12.1f} KiB  sum {sum(v):14.1f}")
        tot += sum(v)
    res[tag + "_bytes_per_step"] = tot * 1024.0 * mult / steps
    print(f"{counter}: {res[tag + '_bytes_per_step'] / 1e9:.3f} GB per decode step ({steps} steps profiled)")
rec = {"model": "xl", "batch": batch, "cfg_scale": 1.0, "precision": precision, "weights_fp8": False, "build_id": build_id, "image_hw": [512, 512], "adapter_size": "small",
       "steps_profiled": steps, **res,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/pmc_workload.py at the mean decode position "
               "(CAR_DEBUG_SKIP_STEPS); FETCH_SIZE KiB x2 (gfx950 wide-read correction), WRITE_SIZE as reported; decode-loop kernels only"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
