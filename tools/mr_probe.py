"""GPU probe for BASELINE config 4 (SURVEY.md §8d'): the multi-resolution model (block_size 2304 = rope grid 48, sample_t2i_MR.py:73-78) in ONE context,
batch 1, cfg 4, both orientations (768x512: 48 x 32 tokens on the 48-wide rope grid — the linear-index quirk; 512x768: 32 x 48) and 512x512, alternating:
what a change of N between calls costs (KV cache re-size, resize / position-embedding tables, hipGraph re-capture) against a repeated call.  Not a test.
usage: mr_probe.py [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine

cfg = C.xl_t2i(2304, "small", "canny")
gsd, vsd = synth.path_state_dicts(cfg, 0)
eng = Engine(cfg, "bf16"); eng.load_state_dict(gsd, finalize=True)
vq = Engine(cfg, "bf16"); vq.load_state_dict(vsd, finalize=True)
emb, mask = synth.text_embeddings(1, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
emb, mask = emb.to(torch.bfloat16).cuda(), mask.cuda()
seq = [(768, 512), (768, 512), (512, 768), (512, 768), (512, 512), (512, 512), (768, 512), (512, 768), (512, 512)]
seen, rows, first_tokens = set(), [], {}
for (H, W) in seq:
    img = synth.canny_like_control(1, H, W).to(torch.bfloat16).cuda()
    gh, gw = H // 16, W // 16
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.encode_control(img); torch.cuda.synchronize(); t1 = time.perf_counter()
    toks = eng.generate(emb, gh * gw, mask, cfg_scale=4.0); torch.cuda.synchronize(); t2 = time.perf_counter()
    px = vq.vq_decode(toks, gh, gw); torch.cuda.synchronize(); t3 = time.perf_counter()
    st = eng.stats()
    key = (H, W)
    same = None
    if key in first_tokens:
        same = bool(torch.equal(first_tokens[key], toks.cpu()))
    else:
        first_tokens[key] = toks.cpu()
    rows.append(dict(HxW=f"{H}x{W}", tokens=gh * gw, first_call_at_this_shape=key not in seen, encode_ms=round((t1 - t0) * 1e3, 1), generate_ms=round((t2 - t1) * 1e3, 1),
                     vq_decode_ms=round((t3 - t2) * 1e3, 1), image_s=round(t3 - t0, 3), decode_ms_per_step=round(st["decode_ms"] / max(st["decode_steps"], 1), 4),
                     prefill_ms=round(st["prefill_ms"], 1), graph=st["graph_used"], same_tokens_as_first_call=same, finite=bool(torch.isfinite(px).all())))
    seen.add(key)
    print(json.dumps(rows[-1]), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
