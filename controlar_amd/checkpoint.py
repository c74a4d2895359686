"""Checkpoint I/O of the path: the file formats the reference's samplers read, the key bookkeeping of
``load_state_dict(strict=...)``, and an on-disk cache of the PACKED weight images.

Reference behaviour mirrored here:
  * ``sample_t2i.py:64-83`` — GPT weights come from a ``.safetensors`` file (``safetensors.torch.load_file``) or from a
    ``torch.load`` checkpoint whose state dict sits under ``"model"`` (DDP), ``"module"`` (DeepSpeed) or ``"state_dict"``;
    anything else raises ``Exception("please check model weight")``; ``load_state_dict(..., strict=False)``.
  * ``sample_t2i.py:48-49`` — the VQ tokenizer is ``torch.load(...)["model"]``, loaded strictly.
  * ``demo/model.py:66-75`` re-reads the safetensors file on every request; the packed-image cache removes that cost:
    the images the HIP kernels stream (MFMA-fragment / e4m3 decode linears, implicit-GEMM conv layouts, ...) are written once
    per (checkpoint content, car_config, library build) and restored by plain copies (car_export_packed / car_import_packed).
"""
from __future__ import annotations

import hashlib
import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .config import PathConfig

PACK_FORMAT = "CARPK03"          # bump together with engine_weights.hip kPackMagic


def _torch_load(path: str, trust: bool = False):
    """``torch.load`` for the checkpoints the reference's training scripts write:
    ``{"model", "optimizer", "steps", "args": argparse.Namespace}`` (train_t2i_canny.py:208, train_c2i_depth.py:252).
    The weights-only unpickler stays on; ``argparse.Namespace`` (a plain attribute bag) is allow-listed for it.  Anything else the
    unpickler refuses is an error unless the caller states that the file is trusted (``trust=True`` -> full pickle)."""
    import argparse
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        if not trust:
            raise
        return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(path: str, *, vq: bool = False, trust_pickle: bool = False) -> Dict[str, torch.Tensor]:
    """The state dict stored in `path`, by the reference's rules (sample_t2i.py:48-49 for ``vq=True``, :64-83 otherwise).
    `trust_pickle`: fall back to the full unpickler for a file whose non-tensor entries the weights-only loader refuses."""
    _, ext = os.path.splitext(path)
    if ext.lower() == ".safetensors":
        from safetensors.torch import load_file
        return load_file(path)
    checkpoint = _torch_load(path, trust=trust_pickle)
    if vq:
        return checkpoint["model"]
    if "model" in checkpoint:            # ddp
        return checkpoint["model"]
    if "module" in checkpoint:           # deepspeed
        return checkpoint["module"]
    if "state_dict" in checkpoint:
        return checkpoint["state_dict"]
    raise Exception("please check model weight")


# ---------------------------------------------------------------------------------------------- expected keys
def expected_gpt_keys(cfg: PathConfig) -> List[str]:
    """Parameter names of gpt_t2i.Transformer / gpt.Transformer (incl. the HF encoder under ``adapter.model.``) that the path reads
    (SURVEY.md §8b weight contract; engine_weights.hip car_finalize_weights)."""
    g, v = cfg.gpt, cfg.vit
    keys = ["tok_embeddings.weight", "norm.weight", "output.weight", "adapter_mlp.fc1.weight", "adapter_mlp.fc2.weight",
            "condition_mlp.cap_proj.fc1.weight", "condition_mlp.cap_proj.fc2.weight"]
    if g.model_type == "c2i":
        keys.append("cls_embedding.embedding_table.weight")
    else:
        keys += ["cls_embedding.cap_proj.fc1.weight", "cls_embedding.cap_proj.fc2.weight", "cls_embedding.uncond_embedding"]
    for k in range(3):
        keys += [f"condition_layers.{k}.fc1.weight", f"condition_layers.{k}.fc2.weight"]
    for i in range(g.n_layer):
        p = f"layers.{i}."
        keys += [p + s for s in ("attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight", "feed_forward.w3.weight",
                                 "feed_forward.w2.weight", "attention_norm.weight", "ffn_norm.weight")]
    a = "adapter.model."
    keys += [a + s for s in ("embeddings.cls_token", "embeddings.position_embeddings", "embeddings.patch_embeddings.projection.weight",
                             "embeddings.patch_embeddings.projection.bias", "layernorm.weight", "layernorm.bias")]
    vit16 = getattr(v, "variant", "dinov2") == "vit"
    for i in range(v.layers):
        p = f"{a}encoder.layer.{i}."
        keys += [p + s for s in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias",
                                 "attention.attention.query.weight", "attention.attention.query.bias", "attention.attention.key.weight",
                                 "attention.attention.key.bias", "attention.attention.value.weight", "attention.attention.value.bias",
                                 "attention.output.dense.weight", "attention.output.dense.bias",
                                 "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
        if not vit16:
            keys += [p + "layer_scale1.lambda1", p + "layer_scale2.lambda1"]
    return keys


def expected_vq_keys(vq, side: str = "decoder") -> List[str]:
    """Parameter names of VQModel's decode side (quantize + post_quant_conv + decoder, vq_model.py:28-39,129-169) or of its
    encode side (encoder + quant_conv) that the path reads."""
    from .synth import vq_decoder_layout, vq_encoder_layout

    def mod(layout):
        out = []
        for item in layout:
            if item[0] == "res":
                _, name, cin, cout = item
                for m in ("norm1", "conv1", "norm2", "conv2") + (("nin_shortcut",) if cin != cout else ()):
                    out += [f"{name}.{m}.weight", f"{name}.{m}.bias"]
            elif item[0] == "attn":
                for m in ("norm", "q", "k", "v", "proj_out"):
                    out += [f"{item[1]}.{m}.weight", f"{item[1]}.{m}.bias"]
            else:
                out += [f"{item[1]}.conv.weight", f"{item[1]}.conv.bias"]
        return out
    if side == "decoder":
        keys = ["quantize.embedding.weight", "post_quant_conv.weight", "post_quant_conv.bias", "decoder.conv_in.weight", "decoder.conv_in.bias"]
        keys += mod(vq_decoder_layout(vq)[0])
        return keys + ["decoder.norm_out.weight", "decoder.norm_out.bias", "decoder.conv_out.weight", "decoder.conv_out.bias"]
    keys = ["encoder.conv_in.weight", "encoder.conv_in.bias"] + mod(vq_encoder_layout(vq)[0])
    return keys + ["encoder.norm_out.weight", "encoder.norm_out.bias", "encoder.conv_out.weight", "encoder.conv_out.bias", "quant_conv.weight", "quant_conv.bias"]


# reference parameters/buffers that exist in the checkpoints but that inference never reads (engine_weights.hip car_load_tensor)
IGNORED_GPT = ("condition_embeddings.weight", "condition_mlp.uncond_embedding", "adapter.model.embeddings.mask_token",
               "condition_norm.weight", "freqs_cis", "causal_mask")


def _canon_adapter_key(k: str) -> str:
    """HF ViT/Dinov2 key spellings of transformers 5.x -> the 4.x checkpoint names (engine_weights.hip canon_name)."""
    if not k.startswith("adapter.model."):
        return k
    k = k.replace("adapter.model.layers.", "adapter.model.encoder.layer.")
    for a, b in ((".attention.q_proj.", ".attention.attention.query."), (".attention.k_proj.", ".attention.attention.key."),
                 (".attention.v_proj.", ".attention.attention.value."), (".attention.o_proj.", ".attention.output.dense."),
                 (".layernorm_before.", ".norm1."), (".layernorm_after.", ".norm2."), (".intermediate.dense.", ".mlp.fc1.")):
        k = k.replace(a, b)
    if ".attention.output.dense." not in k:
        k = k.replace(".output.dense.", ".mlp.fc2.")
    return k


def key_report(expected: Iterable[str], provided: Iterable[str], ignored: Iterable[str] = ()) -> Tuple[List[str], List[str]]:
    """(missing_keys, unexpected_keys) as torch's load_state_dict reports them."""
    exp, ign = list(expected), tuple(ignored)
    got = {_canon_adapter_key(k) for k in provided}
    missing = [k for k in exp if k not in got]
    es = set(exp)
    unexpected = [k for k in provided if _canon_adapter_key(k) not in es and not any(_canon_adapter_key(k) == i or _canon_adapter_key(k).endswith(i) or
                                                                                   _canon_adapter_key(k).startswith("adapter.model.pooler.") for i in ign)]
    return missing, unexpected


# ---------------------------------------------------------------------------------------------- packed-image cache
def content_key(paths: Iterable[str], cfg_bytes: bytes, extra: str = "") -> str:
    """blake2b over the checkpoint files' CONTENT + the car_config bytes + the pack format tag + `extra` (precision, library build id)."""
    h = hashlib.blake2b(digest_size=16)
    h.update(PACK_FORMAT.encode()); h.update(cfg_bytes); h.update(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            while True:
                b = f.read(1 << 24)
                if not b:
                    break
                h.update(b)
    return h.hexdigest()


def default_cache_dir() -> str:
    return os.environ.get("CONTROLAR_PACK_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "controlar_amd"))


def load_engine_from_checkpoints(engine, gpt_path: Optional[str] = None, vq_path: Optional[str] = None,
                                 cache_dir: Optional[str] = None, use_cache: bool = True) -> dict:
    """Fill `engine` (controlar_amd.engine.Engine) from checkpoint files, through the packed-image cache.
    Returns {"cache": "hit" | "miss" | "off", "file": path-or-None}."""
    import ctypes as C
    paths = [p for p in (gpt_path, vq_path) if p]
    if not paths:
        raise ValueError("no checkpoint given")
    info = {"cache": "off", "file": None}
    cfile = None
    if use_cache:
        cdir = cache_dir or default_cache_dir()
        os.makedirs(cdir, exist_ok=True)
        # the images are a private format of one library build: its id is part of the key (and of the file header, engine_weights.hip)
        build = engine.lib.car_build_id().decode()
        cfile = os.path.join(cdir, content_key(paths, bytes(engine._cc), engine.precision + "|" + build) + ".carpk")
        info["file"] = cfile
        if os.path.exists(cfile):
            rc = engine.lib.car_import_packed(engine._h, cfile.encode())
            if rc == 0:
                info["cache"] = "hit"
                return info
            os.remove(cfile)                       # stale / foreign file: fall through to a normal load
    if gpt_path:
        engine.load_state_dict(load_checkpoint(gpt_path))
    if vq_path:
        engine.load_state_dict(load_checkpoint(vq_path, vq=True))
    engine.finalize()
    if cfile:
        tmp = cfile + f".tmp{os.getpid()}"
        engine._check(engine.lib.car_export_packed(engine._h, tmp.encode()), "car_export_packed")
        os.replace(tmp, cfile)
        info["cache"] = "miss"
    return info
