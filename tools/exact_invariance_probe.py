"""GPU probe: where does the exact mode stop being batch-invariant?  Runs tests/test_parity_gpu.py::test_exact_mode_is_batch_invariant's inputs and reports the
first (step, row, column) at which the logits of a sequence decoded alone and inside a 9x batch differ."""
import os, sys
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from cases import load_case
from controlar_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "tiny_depth_cfg4"
cs = load_case(name)
eng = Engine(cs["cfg"], "fp32"); eng.load_state_dict(cs["gsd"]); eng.finalize()
B = cs["B"]
mask = cs["mask"].cuda() if cs.get("mask") is not None else None
for reps in (1, 2, 3, 9):
    img, emb = cs["img"].repeat(reps, 1, 1, 1), cs["emb"].repeat(reps, 1, 1)
    mk = cs["mask"].repeat(reps, 1).cuda() if cs.get("mask") is not None else None
    eng.encode_control(img.cuda())
    t, l = eng.generate(emb.cuda(), cs["n_new"], mk, cfg_scale=cs["cfg_scale"], control_strength=cs["control_strength"], return_logits=True)
    t, l = t.cpu(), l.cpu()
    if reps == 1:
        t1, l1 = t, l
        continue
    for r in range(reps):
        d = (l[r * B:(r + 1) * B] != l1)
        if d.any():
            idx = d.nonzero()[0].tolist()
            steps = sorted(set(d.nonzero()[:, 1].tolist()))
            print(f"reps={reps} copy {r}: first diff (row, step, col) = {idx}, |d| = {float((l[r*B:(r+1)*B] - l1).abs().max()):.3g}, steps with a diff: {steps[:8]}... ({len(steps)} of {l.shape[1]})")
        else:
            print(f"reps={reps} copy {r}: bit-identical")
