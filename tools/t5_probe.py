#!/usr/bin/env python3
"""Time car_t5_encode (Flan-T5-XL, bf16, synthetic weights) at the batch sizes of the bench: usage t5_probe.py [B ...].
Prints ms per call, prompts/s and achieved dense bf16 TFLOP/s (2*params_linear*tokens + attention matmuls)."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from controlar_amd import config as C  # noqa: E402
from controlar_amd import synth  # noqa: E402
from controlar_amd.t5 import T5Embedder  # noqa: E402

cfg = C.flan_t5_xl()
t0 = time.time()
sd = {k: v.to(torch.bfloat16) for k, v in synth.t5_state_dict(cfg).items()}
print(f"weights {time.time() - t0:.1f}s", flush=True)
t0 = time.time()
emb = T5Embedder("cuda", config=cfg, state_dict=sd, torch_dtype=torch.bfloat16)
torch.cuda.synchronize()
print(f"load {time.time() - t0:.1f}s", flush=True)
T = cfg.model_max_length
inner = cfg.num_heads * cfg.d_kv
lin = cfg.num_layers * (4 * inner * cfg.d_model + 3 * cfg.d_ff * cfg.d_model)
for B in [int(a) for a in sys.argv[1:]] or [1, 16, 64, 256]:
    ids, mask = synth.t5_tokens(B, cfg)
    ids, mask = ids.cuda(), mask.cuda()
    for _ in range(2):
        emb.encode_ids(ids, mask)
    torch.cuda.synchronize()
    n = 5
    t0 = time.time()
    for _ in range(n):
        emb.encode_ids(ids, mask)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    flops = 2.0 * lin * B * T + cfg.num_layers * 4.0 * B * cfg.num_heads * T * T * cfg.d_kv
    print(f"B={B}: {ms:.2f} ms  {B / ms * 1e3:.0f} prompts/s  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)
