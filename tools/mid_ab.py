"""GPU probe: A/B of decode-schedule knobs at a given batch inside ONE process (same box, same clocks).  Not a test.
usage: mid_ab.py B cfg_scale n_new "K1=V1,K2=V2;K3=V3;..."   (knob sets separated by ';', the empty set = defaults)"""
import sys, os, json
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine
from controlar_amd import _lib
if os.environ.get('AB_LIB'): _lib.DEV_LIB_PATH = os.environ['AB_LIB']      # an experimental build of the development library

B = int(sys.argv[1]); cfg_scale = float(sys.argv[2]); n_new = int(sys.argv[3])
sets = [dict(kv.split("=") for kv in s.split(",") if kv) for s in sys.argv[4].split(";")]
model = sys.argv[5] if len(sys.argv) > 5 else "xl"
prec = sys.argv[6] if len(sys.argv) > 6 else "bf16"
cfg = C.xl_t2i(1024) if model == "xl" else C.b_t2i(1024)
gsd, _ = synth.path_state_dicts(cfg, 0)
eng = Engine(cfg, prec); eng.load_state_dict(gsd); eng.finalize()
img = synth.canny_like_control(B, 512, 512).to(torch.bfloat16).cuda()
emb, mask = synth.text_embeddings(B, 120, 2048)
emb = emb.to(torch.bfloat16).cuda(); mask = mask.cuda()
if prec == "fp32":
    img, emb = img.float(), emb.float()
ref = None
used = set()
for env in sets:
    for k in used:
        os.environ.pop(k, None)
    os.environ.update(env); used |= set(env)
    for rep in range(2):
        eng.encode_control(img)
        toks = eng.generate(emb, n_new, mask, cfg_scale=cfg_scale)
        torch.cuda.synchronize()
        st = eng.stats()
    same = None
    if ref is None:
        ref = toks.clone()
    else:
        same = bool(torch.equal(ref, toks))
    ms = st["decode_ms"] / st["decode_steps"]
    gbs = st["decode_algo_bytes"] / st["decode_steps"] / (ms * 1e-3) / 1e9
    print(json.dumps(dict(B=B, cfg_scale=cfg_scale, knobs=env, ms_per_step=round(ms, 4), us_per_layer=round(ms * 1e3 / cfg.gpt.n_layer, 2), frac=round(gbs / 8000, 4),
                          kernels=st["decode_kernels_per_step"], tokens_equal_first=same)), flush=True)
