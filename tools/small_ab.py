"""GPU probe: A/B of the small-batch decode schedule knobs inside ONE process (same box, same clocks).  Not a test.
usage: small_ab.py 2,8,12,16,32 [n_new=1024] [cfg_scale=1.0]"""
import sys, os, json
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine

batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,8,16").split(",")]
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg_scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cfg = C.xl_t2i(1024)
gsd, _ = synth.path_state_dicts(cfg, 0)
eng = Engine(cfg, "bf16"); eng.load_state_dict(gsd); eng.finalize()
KNOBS = [("default", {}), ("split-KV attention", {"CAR_ATTN_SPLIT_SMALL": "1"}), ("separate norms", {"CAR_NO_SMALL_FUSE": "1"}),
         ("split-KV + separate norms", {"CAR_ATTN_SPLIT_SMALL": "1", "CAR_NO_SMALL_FUSE": "1"})]
for B in batches:
    img = synth.canny_like_control(B, 512, 512).to(torch.bfloat16).cuda()
    emb, mask = synth.text_embeddings(B, 120, 2048)
    emb = emb.to(torch.bfloat16).cuda(); mask = mask.cuda()
    ref = None
    for name, env in KNOBS:
        for k in ("CAR_ATTN_SPLIT_SMALL", "CAR_NO_SMALL_FUSE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for rep in range(2):
            eng.encode_control(img)
            toks = eng.generate(emb, n_new, mask, cfg_scale=cfg_scale)
            torch.cuda.synchronize()
            st = eng.stats()
        ms = st["decode_ms"] / st["decode_steps"]
        gbs = st["decode_algo_bytes"] / st["decode_steps"] / (ms * 1e-3) / 1e9
        print(json.dumps(dict(B=B, cfg_scale=cfg_scale, knobs=name, ms_per_step=round(ms, 4), frac=round(gbs / 8000, 4), kernels=st["decode_kernels_per_step"])), flush=True)
