"""Shared case table for the parity tests (same seeds as tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from controlar_amd import config as C, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (config factory, control kind, cfg_interval)
CASES = {
    "tiny_canny_cfg1": (lambda: C.tiny_t2i(64, "canny"), "canny", -1),
    "tiny_depth_cfg4": (lambda: C.tiny_t2i(64, "depth"), "smooth", -1),
    "tiny_mr_192x128": (lambda: C.tiny_t2i(144, "canny"), "canny", -1),
    "tiny_mr_128x192": (lambda: C.tiny_t2i(144, "canny"), "canny", -1),
    "tiny_cfg_interval": (lambda: C.tiny_t2i(64, "canny"), "canny", 20),
    "tiny_hed_base_cfg1p5": (lambda: C.tiny_t2i_base(64, "hed"), "smooth", -1),
    "tiny_mask_edges": (lambda: C.tiny_t2i(64, "canny"), "canny", -1),
    "tiny_no_mask": (lambda: C.tiny_t2i(64, "canny"), "canny", -1),
}


def load_case(name):
    mk, control, interval = CASES[name]
    cfg = mk()
    gold = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    B, H, W, seed, _threads = [int(x) for x in gold["meta"]]
    img = synth.canny_like_control(B, H, W) if control == "canny" else synth.smooth_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    if name == "tiny_mask_edges":       # one real token | no padding | typical
        emb, mask = synth.text_embeddings_with_lengths([1, 120, 40], cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    elif name == "tiny_no_mask":        # generate(..., emb_masks=None): plain causal mask (generate.py:184)
        mask = None
    gsd, vsd = synth.path_state_dicts(cfg, seed=seed)
    return dict(cfg=cfg, gold=gold, B=B, H=H, W=W, img=img, emb=emb, mask=mask, gsd=gsd, vsd=vsd,
                cfg_scale=float(gold["cfg_scale"]), control_strength=float(gold["control_strength"]),
                cfg_interval=interval, n_new=(H // 16) * (W // 16))


def record_measured(name, **vals):
    """Append a measured deviation to gpurun_out/parity_measured.jsonl (merged back from the GPU box; the round's copy is committed under profiles/):
    the tolerances of the bf16 pixel tests are 1.5 x what these lines say."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}}) + "\n")
    except OSError:
        pass
