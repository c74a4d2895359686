"""Two-chain determinism probe: identical inputs in rows 0 and B/2 (different decode chains) must give identical free-running tokens (what bench.py asserts)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from controlar_amd import config as C, synth
from controlar_amd import _lib
if os.environ.get('TWIN_LIB'): _lib.LIB_PATH = os.environ['TWIN_LIB']
from controlar_amd.engine import Engine
model = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_new = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = C.tiny_t2i(64, "canny") if model == "tiny" else C.b_t2i(256, adapter_size="small", condition_type="canny")
hw = 128 if model == "tiny" else 256
gsd, _ = synth.path_state_dicts(cfg, seed=0)
img = synth.canny_like_control(B, hw, hw); emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
t = B // 2
img[t], emb[t], mask[t] = img[0], emb[0], mask[0]
eng = Engine(cfg, "bf16", dev=bool(os.environ.get("CONTROLAR_DEV_LIB")))
eng.load_state_dict(gsd); eng.finalize()
eng.encode_control(img.cuda())
for it in range(3):
    toks, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=1.0, return_logits=True)
    a, b = toks[0].cpu(), toks[t].cpu()
    neq = (a != b).nonzero().flatten()
    dl = (logits[0] - logits[t]).abs().amax(dim=1).cpu()
    print(f"{model} B={B} call {it}: twin tokens equal {bool(torch.equal(a, b))}", "first diff at" if len(neq) else "", int(neq[0]) if len(neq) else "", "| first step with different logits:", int((dl > 0).nonzero()[0]) if (dl > 0).any() else None, "| kernels/step", eng.stats()["decode_kernels_per_step"])
eng.close()
