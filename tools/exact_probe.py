"""GPU probe (not a test): the EXACT-mode decode step of the XL model at the mean KV position, by batch and chain count.
The loop is started `skip` positions late (CAR_DEBUG_SKIP_STEPS: the skipped cache rows hold stale / zero rows, tokens are meaningless) so a handful of
graph replays see the KV prefix of the middle of an image; weights are loaded once.
usage: exact_probe.py 384,192 1,2 [skip=509] [steps=20] [extra env assignments K=V ...]"""
import sys, os, json
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine

batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "384").split(",")]
chains = [x for x in (sys.argv[2] if len(sys.argv) > 2 else "2").split(",")]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 509
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
for kv in sys.argv[5:]:
    k, v = kv.split("="); os.environ[k] = v
os.environ["CAR_DEBUG_SKIP_STEPS"] = str(skip)
cfg = C.xl_t2i(1024)
gsd, _ = synth.path_state_dicts(cfg, 0)
eng = Engine(cfg, "fp32"); eng.load_state_dict(gsd); eng.finalize()
for B in batches:
    img = synth.canny_like_control(B, 512, 512).float().cuda()
    emb, mask = synth.text_embeddings(B, 120, 2048)
    emb = emb.float().cuda(); mask = mask.cuda()
    eng.encode_control(img); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.encode_control(img); e1.record(); torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1)
    for ch in chains:
        os.environ["CAR_CHAINS"] = ch
        for rep in range(2):
            eng.generate(emb, skip + steps + 1, mask, cfg_scale=1.0); torch.cuda.synchronize()
            st = eng.stats()
        ms = st["decode_ms"] / st["decode_steps"]
        print(json.dumps(dict(B=B, chains=ch, pos0=120 + skip, steps=st["decode_steps"], ms_per_step=round(ms, 4), prefill_ms=round(st["prefill_ms"], 1), encode_ms=round(enc_ms, 1),
                              kernels=st["decode_kernels_per_step"], graph=st.get("graph_used"))), flush=True)
