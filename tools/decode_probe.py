"""GPU probe: decode-step time vs batch / chain count for the XL model (weights loaded once).  Not a test.
usage: decode_probe.py xl 256,128 1024 1,2,4 [cfg_scale] [fp8|fp8mfma]"""
import sys, os, time, json
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine

model = sys.argv[1] if len(sys.argv) > 1 else "xl"
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,8,16,32,64").split(",")]
n_new = int(sys.argv[3]) if len(sys.argv) > 3 else 256
chains = [x for x in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if x] or [None]
cfg_scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
fp8 = {"fp8": True, "fp8mfma": "mfma"}.get(sys.argv[6], False) if len(sys.argv) > 6 else False   # fp8 = weight-only e4m3, fp8mfma = W8A8 on the fp8 MFMA
kv8 = len(sys.argv) > 6 and sys.argv[6] == "kv8"                   # opt-in e4m3 KV cache
cfg = C.xl_t2i(1024) if model == "xl" else C.b_t2i(1024)
t0 = time.time()
gsd, _ = synth.path_state_dicts(cfg, 0)
eng = Engine(cfg, "bf16", weights_fp8=fp8, kv_fp8=kv8); eng.load_state_dict(gsd); eng.finalize()
print("load %.1fs" % (time.time() - t0), flush=True)
for B, ch in [(B, ch) for B in batches for ch in chains]:
    if ch is not None:
        os.environ["CAR_CHAINS"] = ch
    img = synth.canny_like_control(B, 512, 512).to(torch.bfloat16).cuda()
    emb, mask = synth.text_embeddings(B, 120, 2048)
    emb = emb.to(torch.bfloat16).cuda(); mask = mask.cuda()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        eng.encode_control(img); torch.cuda.synchronize(); t1 = time.time()
        eng.generate(emb, n_new, mask, cfg_scale=cfg_scale); torch.cuda.synchronize(); t2 = time.time()
        st = eng.stats()
    ms = st["decode_ms"] / st["decode_steps"]
    gbs = st["decode_algo_bytes"] / st["decode_steps"] / (ms * 1e-3) / 1e9
    print(json.dumps(dict(B=B, cfg_scale=cfg_scale, fp8=fp8, kv8=kv8, chains=ch, n_new=n_new, enc_ms=round((t1 - t0) * 1e3, 1), gen_ms=round((t2 - t1) * 1e3, 1), prefill_ms=round(st["prefill_ms"], 1),
                          ms_per_step=round(ms, 4), algo_GBps=round(gbs, 1), frac=round(gbs / 8000, 4), kernels=st["decode_kernels_per_step"])), flush=True)
