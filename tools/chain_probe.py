"""GPU probe: two-chain overlap vs attention occupancy (CAR_ATTN_VARIANT / CAR_ATTN_LDS_PAD knobs).  Not a test."""
import sys, os, time, json
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = C.xl_t2i(1024)
gsd, _ = synth.path_state_dicts(cfg, 0)
eng = Engine(cfg, "bf16"); eng.load_state_dict(gsd); eng.finalize()
img = synth.canny_like_control(B, 512, 512).to(torch.bfloat16).cuda()
emb, mask = synth.text_embeddings(B, 120, 2048)
emb = emb.to(torch.bfloat16).cuda(); mask = mask.cuda()
eng.encode_control(img)
for chains in sys.argv[3].split(","):
    for var in sys.argv[4].split(","):
        for pad in sys.argv[5].split(","):
            os.environ["CAR_CHAINS"], os.environ["CAR_ATTN_VARIANT"], os.environ["CAR_ATTN_LDS_PAD"] = chains, var, pad
            for rep in range(2):
                eng.generate(emb, n_new, mask, cfg_scale=1.0); torch.cuda.synchronize()
                st = eng.stats()
            ms = st["decode_ms"] / st["decode_steps"]
            print(json.dumps(dict(B=B, chains=chains, attn=var, lds_pad=pad, ms_per_step=round(ms, 4), frac=round(st["decode_algo_bytes"] / st["decode_steps"] / (ms * 1e-3) / 8e12, 4))), flush=True)
