"""generate() with the reference signature (autoregressive/models/generate.py:134-204), running the whole
loop in libcontrolar_hip.so.  Callers that keep working unchanged: sample_t2i.py:163, sample_t2i_MR.py:184,
test_t2i.py:220, demo/model.py:156,250."""
from __future__ import annotations

import torch

from .models import Transformer, _PendingControl


def _call_seed(kw) -> int:
    """Seed of the library's counter-based RNG for this call.  The reference draws from torch's global generator
    (torch.multinomial, generate.py:72), so consecutive calls differ and torch.manual_seed() reproduces a run; the same holds
    here: without an explicit `seed` kwarg (an extension) one 62-bit value is drawn from torch's CPU generator per call."""
    if "seed" in kw and kw["seed"] is not None:
        return int(kw["seed"])
    return int(torch.randint(0, 2 ** 62, (1,)).item())


@torch.no_grad()
def generate(model: Transformer, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, condition=None,
             condition_null=None, condition_token_nums=0, control_strength=1, **sampling_kwargs):
    """Returns int32 [B, max_new_tokens] on cond.device.  sampling_kwargs: temperature, top_k (default 2000 as sample(),
    generate.py:59), top_p, sample_logits (default True).  `condition` is the control image [B,3,H,W] in [-1,1] (or None)."""
    if model.model_type not in ("t2i", "c2i"):
        raise Exception("please check model type")
    eng = model.engine
    if condition is not None:
        if isinstance(condition, _PendingControl):
            condition = condition.img
        eng.encode_control(condition)                       # model.adapter + model.adapter_mlp (generate.py:136-138)
    if emb_masks is not None:
        assert emb_masks.shape[0] == cond.shape[0]          # generate.py:185-186
        assert emb_masks.shape[-1] == (1 + condition_token_nums if model.model_type == "c2i" else cond.shape[1])
    if model.model_type == "c2i" and condition_token_nums != 0:
        raise RuntimeError("c2i: condition_token_nums must be 0 (the only value the reference samplers pass, sample_c2i.py:55)")
    out = eng.generate(cond, int(max_new_tokens), emb_masks, cfg_scale=float(cfg_scale), cfg_interval=int(cfg_interval),
                       use_control=condition is not None, control_strength=float(control_strength),
                       temperature=float(sampling_kwargs.get("temperature", 1.0)), top_k=int(sampling_kwargs.get("top_k", 2000) or 0),
                       top_p=float(sampling_kwargs.get("top_p", 1.0)), sample_logits=bool(sampling_kwargs.get("sample_logits", True)),
                       seed=_call_seed(sampling_kwargs))
    return out.to(cond.device)
