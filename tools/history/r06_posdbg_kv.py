"""Round-6 diagnostic: chain 1 = a row-by-row copy of chain 0; after one free-running generate() with the early-*pos build, compare the two halves of the packed KV cache
and report where (layer, K or V, position, rows, heads, dims) they first differ.  Needs car_posdbg_kv (alt build only)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from controlar_amd import config as C, synth
from controlar_amd import _lib
_lib.LIB_PATH = os.environ['TWIN_LIB']
from controlar_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 48
cfg = C.b_t2i(256, adapter_size="small", condition_type="canny")
gsd, _ = synth.path_state_dicts(cfg, seed=0)
img = synth.canny_like_control(B, 256, 256); emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
t = B // 2
img[t:], emb[t:], mask[t:] = img[:t].clone(), emb[:t].clone(), mask[:t].clone()
eng = Engine(cfg, "bf16")
eng.load_state_dict(gsd); eng.finalize()
eng.encode_control(img.cuda())
T = cfg.gpt.cls_token_num; L = cfg.gpt.n_layer; H = cfg.gpt.n_head
hip = ctypes.CDLL("libamdhip64.so")
for it in range(int(os.environ.get("CALLS", "2"))):
    toks = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=1.0)
    tc = toks.cpu(); dtok = (tc[:t] != tc[t:])
    first_tok = torch.where(dtok.any(1), dtok.float().argmax(1), torch.full((t,), 10**6))        # per row: first differing token index
    print(f"call {it}: rows with different tokens {int(dtok.any(1).sum())} of {t}", flush=True)
    p = ctypes.c_void_p(); cap = ctypes.c_ulonglong()
    eng.lib.car_posdbg_kv(eng._h, ctypes.byref(p), ctypes.byref(cap))
    SA = ((T + n_new + 7) // 8 * 8 + 31) // 32 * 32; SA64 = SA * 64          # engine_generate.hip: S_max = rup(T + n_new, 8), SA = rup(S_max, 32)
    n = 2 * L * B * H * SA64
    assert n * 2 <= cap.value, (n, cap.value)
    kv = torch.empty(n, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    assert hip.hipMemcpy(ctypes.c_void_p(kv.data_ptr()), p, ctypes.c_size_t(n * 2), 3) == 0
    kv = kv.view(2 * L, B, H, SA64)
    e = torch.arange(SA64, device="cuda")
    posK = (e // 1024) * 16 + ((e % 512) // 8) % 16
    dimK = ((e % 1024) // 512) * 32 + (((e % 512) // 8) // 16) * 8 + e % 8
    idx = e % 512; ev = idx % 8; qv = (idx // 8) // 16
    w = torch.where(ev < 4, qv * 4 + ev, 16 + qv * 4 + (ev - 4))
    posV = (e // 2048) * 32 + w
    dimV = ((e % 2048) // 512) * 16 + (idx // 8) % 16
    print(f"  SA {SA}, T {T}; first position with a K / V difference per layer (positions >= T were written by the decode steps):")
    events = []
    for l in range(L):
        out = []
        for isv, (pp, dd) in enumerate(((posK, dimK), (posV, dimV))):
            d = kv[2 * l + isv, :t] != kv[2 * l + isv, t:]                    # [t, H, SA64]
            if not bool(d.any()): out.append(None); continue
            pm = torch.where(d, pp.view(1, 1, -1).expand_as(d), torch.full_like(pp, 10**6).view(1, 1, -1).expand_as(d))
            rowmin = pm.amin(dim=(1, 2))                                       # per row: first differing position
            p0 = int(rowmin.min())
            sel = d & (pp.view(1, 1, -1) == p0)
            rows = sel.any(dim=(1, 2)).nonzero().flatten().tolist(); heads = sel.any(dim=(0, 2)).nonzero().flatten().tolist()
            dims = sorted(set(dd[sel.any(dim=(0, 1))].tolist()))
            out.append((p0, rows[:20], heads, dims[:64]))
            events.append((l, isv, rowmin.cpu()))
            del d, pm, sel
        print(f"   layer {l:2d}: K {out[0]}\n             V {out[1]}", flush=True)
    # per row: the earliest differing position over all layers, the lowest layer showing it, and whether the row's tokens had already diverged by then
    if events:
        allmin = torch.stack([r for _, _, r in events])                          # [events, t]
        rmin, arg = allmin.min(dim=0)
        bad = (rmin < 10**6).nonzero().flatten().tolist()
        rep = []
        for r in bad[:40]:
            p0 = int(rmin[r]); ls = sorted({(events[i][0], "KV"[events[i][1]]) for i in range(len(events)) if int(allmin[i, r]) == p0})
            rep.append((r, p0, ls[:4], "first token diff %d" % int(first_tok[r]) if int(first_tok[r]) < 10**6 else "tokens equal"))
        print("  per row: (row, first differing position, (layer, K/V) showing it, token status; token i is sampled at position T+i-1 and written to the KV at T+i):")
        for x in rep: print("    ", x)
    del kv
eng.close()
