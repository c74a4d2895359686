"""Minimal workload for rocprofv3 --pmc passes (counter collection serialises and slows every dispatch):
XL weights, B sequences, encode + prefill + a few eager decode steps.  CAR_DEBUG_SKIP_STEPS=<n> starts the loop n positions
late (engine_generate.hip) so that the few profiled steps run over a long KV prefix.
usage: pmc_workload.py B n_new [skip] [bf16|fp32]   -> decode steps profiled = n_new - 1 - skip at positions 120+skip .."""
import os, sys
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CAR_NO_GRAPH", "1")      # PMC collection cannot follow graph replays: eager launches, chains back to back
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(sys.argv) > 3:
    os.environ["CAR_DEBUG_SKIP_STEPS"] = sys.argv[3]
cfg = C.xl_t2i(1024)
gsd, _ = synth.path_state_dicts(cfg, 0)
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
dt = torch.bfloat16 if prec == "bf16" else torch.float32
eng = Engine(cfg, prec); eng.load_state_dict(gsd); eng.finalize()
img = synth.canny_like_control(B, 512, 512).to(dt).cuda()
emb, mask = synth.text_embeddings(B, 120, 2048)
eng.encode_control(img)
eng.generate(emb.to(dt).cuda(), n_new, mask.cuda(), cfg_scale=1.0)
torch.cuda.synchronize()
print("done", eng.stats())
