"""Round-6 diagnostic (needs the instrumented build of tools/history/r06_posdbg_instrumentation.diff at TWIN_LIB): every dec_gemm QKV workgroup reads *pos at kernel
entry (agent scope, with a timestamp) AND in the epilogue; a mismatch is recorded with the layer's K-cache pointer, the block, both values and both times; the advance
kernel records the time of every write.  Answers: which layer, which chain, which XCDs, and did the kernel start before the write."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from controlar_amd import config as C, synth
from controlar_amd import _lib
_lib.LIB_PATH = os.environ['TWIN_LIB']
from controlar_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 96
cfg = C.b_t2i(256, adapter_size="small", condition_type="canny")
gsd, _ = synth.path_state_dicts(cfg, seed=0)
img = synth.canny_like_control(B, 256, 256); emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
t = B // 2
img[t:], emb[t:], mask[t:] = img[:t].clone(), emb[:t].clone(), mask[:t].clone()      # chain 1 = a copy of chain 0, row by row
eng = Engine(cfg, "bf16")
eng.load_state_dict(gsd); eng.finalize()
eng.encode_control(img.cuda())
T = cfg.gpt.cls_token_num
seen = []
for it in range(int(os.environ.get("CALLS", "4"))):
    toks, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=1.0, return_logits=True)
    print(f"call {it}: NaN logits: chain 0 {int(torch.isnan(logits[:t]).any(dim=2).sum())}, chain 1 {int(torch.isnan(logits[t:]).any(dim=2).sum())} (row, step) pairs", flush=True)
    dl = torch.nan_to_num(logits[:t] - logits[t:], nan=1e9).abs().amax(dim=2)            # [rows, steps]
    nz = (dl > 0)
    per_step = nz.sum(dim=0).cpu().tolist()
    fs = [i for i, c in enumerate(per_step) if c]
    print(f"call {it}: rows with different logits per step (first 40 steps): {per_step[:40]}", flush=True)
    if fs:
        s0 = fs[0]; rr = nz[:, s0].nonzero().flatten().cpu().tolist()
        print(f"   first step with a difference: {s0}; rows {rr[:40]}; max |dlogit| there {float(dl[:, s0].max()):.4g}", flush=True)
    del logits, dl, nz
    a, b = tuple(toks[0].cpu().tolist()), tuple(toks[t].cpu().tolist())
    seen += [a, b]
    tc = toks.cpu(); d = (tc[:t] != tc[t:])
    rows = d.any(dim=1).nonzero().flatten().tolist()
    print(f"call {it}: twins equal {a == b}; rows of chain 1 differing from their chain-0 copy: {len(rows)} of {t}: {rows[:48]}", flush=True)
    if rows: print("   first differing token per such row:", [int(d[r].nonzero()[0]) for r in rows[:48]], flush=True)
    sys.stderr.flush()
    eng.lib.car_posdbg_dump()
    if it == 0:
        if hasattr(eng.lib, "car_posdbg_dump_adv"):
            for ch in range(2): eng.lib.car_posdbg_dump_adv(ch, T, T + n_new)
from collections import Counter
ref = Counter(seen).most_common(1)[0][0]
def fd(x): return next((i for i, (u, v) in enumerate(zip(x, ref)) if u != v), None)
print("first token differing from the consensus sequence, per call (chain 0 twin, chain 1 twin):", [(fd(seen[2 * i]), fd(seen[2 * i + 1])) for i in range(len(seen) // 2)])
eng.close()
