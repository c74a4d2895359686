"""GPU probe (one GPU): the N > 1 start-up path of bench.py at model size — context A builds the XL weights and exports the packed images (what rank 0 does),
context B imports them (what every other rank does); both must decode the same tokens.  Prints the times of the three steps."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlar_amd import config as C, synth
from controlar_amd.engine import Engine

cfg = C.xl_t2i(1024)
t0 = time.time(); gsd, vsd = synth.path_state_dicts(cfg, 0); t_syn = time.time() - t0
a = Engine(cfg, "bf16"); t0 = time.time(); a.load_state_dict(gsd, finalize=True); torch.cuda.synchronize(); t_load = time.time() - t0
d = tempfile.mkdtemp(prefix="car_pk_"); f = os.path.join(d, "gpt.carpk")
t0 = time.time(); a._check(a.lib.car_export_packed(a._h, f.encode()), "car_export_packed"); t_exp = time.time() - t0
b = Engine(cfg, "bf16"); t0 = time.time(); b._check(b.lib.car_import_packed(b._h, f.encode()), "car_import_packed"); torch.cuda.synchronize(); t_imp = time.time() - t0
img = synth.canny_like_control(4, 512, 512).to(torch.bfloat16).cuda(); emb, mask = synth.text_embeddings(4, 120, 2048)
outs = []
for e in (a, b):
    e.encode_control(img); outs.append(e.generate(emb.to(torch.bfloat16).cuda(), 24, mask, cfg_scale=1.0).cpu())
print(dict(synth_s=round(t_syn, 1), load_pack_s=round(t_load, 1), export_s=round(t_exp, 1), import_s=round(t_imp, 1), file_GB=round(os.path.getsize(f) / 1e9, 2), tokens_equal=bool(torch.equal(outs[0], outs[1]))))
os.remove(f); os.rmdir(d)
