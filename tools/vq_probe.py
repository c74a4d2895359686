"""GPU probe: VQ-16 decode (32x32 tokens -> 512x512) and control-encoder time per batch.  Not a test.
usage: vq_probe.py [B=64] [prec=bf16]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = C.xl_t2i(1024)
gsd, vsd = synth.path_state_dicts(cfg, 0)
vq = Engine(cfg, prec); vq.load_state_dict(vsd, finalize=True)
toks = torch.randint(0, 16384, (B, 1024), dtype=torch.int32).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    px = vq.vq_decode(toks, 32, 32); torch.cuda.synchronize(); t1 = time.time()
print(json.dumps(dict(stage="vq_decode", B=B, ms=round((t1 - t0) * 1e3, 2), ms_per_image=round((t1 - t0) * 1e3 / B, 3), tflops=round(1.017 * B / (t1 - t0), 1))), flush=True)
eng = Engine(cfg, prec); eng.load_state_dict(gsd, finalize=True)
img = synth.canny_like_control(B, 512, 512).to(torch.bfloat16).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    eng.encode_control(img); torch.cuda.synchronize(); t1 = time.time()
print(json.dumps(dict(stage="encode_control", B=B, ms=round((t1 - t0) * 1e3, 2), ms_per_image=round((t1 - t0) * 1e3 / B, 3))), flush=True)
