"""Deterministic synthetic weights and inputs (no checkpoints or datasets exist offline).

The recipe is SURVEY.md §8(d): reference-shaped state dicts keyed by the reference's own
parameter names (gpt_t2i.Transformer, HF Dinov2Model under ``adapter.model.``, VQModel),
drawn from a seeded torch CPU generator so the imported reference (golden script), the CPU
oracle and the HIP path all consume bit-identical fp32 tensors.

Why not the reference's own init: ``output.weight`` and every ``MLP`` are zero-initialised
(gpt_t2i.py:377,174-175) so logits and control tokens would be identically 0.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .config import PathConfig, GPTConfig, ViTConfig, VQConfig, T5Config


class _Rng:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)

    def normal(self, *shape, std=1.0, mean=0.0):
        return torch.randn(*shape, generator=self.g, dtype=torch.float32) * std + mean

    def uniform(self, *shape, lo=-1.0, hi=1.0):
        return torch.rand(*shape, generator=self.g, dtype=torch.float32) * (hi - lo) + lo


def gpt_state_dict(cfg: GPTConfig, vit: ViTConfig, seed: int = 0, ctrl_gain: float = 1.1) -> Dict[str, torch.Tensor]:
    """Names follow gpt_t2i.Transformer.state_dict() (reference gpt_t2i.py:310-349).
    SURVEY §8(d) recipe: embeddings N(0,1), transformer linears N(0,0.03^2), output.weight
    N(0,0.2^2); the text/control MLPs are fan-in scaled so control tokens are O(0.3) at every size."""
    r = _Rng(seed)
    D, H = cfg.dim, cfg.ffn_hidden
    lin = 0.03
    ctrl = ctrl_gain * D ** -0.5
    sd: Dict[str, torch.Tensor] = {}
    sd["tok_embeddings.weight"] = r.normal(cfg.vocab_size, D, std=1.0)
    if cfg.model_type == "c2i":      # LabelEmbedder (gpt.py:66-96): num_classes + 1 rows, last = CFG null class
        sd["cls_embedding.embedding_table.weight"] = r.normal(cfg.num_classes + 1, D, std=1.0)
    else:
        sd["cls_embedding.cap_proj.fc1.weight"] = r.normal(D, cfg.caption_dim, std=3.0 * cfg.caption_dim ** -0.5)
        sd["cls_embedding.cap_proj.fc2.weight"] = r.normal(D, D, std=3.0 * D ** -0.5)
        sd["cls_embedding.uncond_embedding"] = r.normal(cfg.cls_token_num, cfg.caption_dim, std=cfg.caption_dim ** -0.5)
    sd["adapter_mlp.fc1.weight"] = r.normal(D, vit.hidden, std=ctrl_gain * vit.hidden ** -0.5)
    sd["adapter_mlp.fc2.weight"] = r.normal(D, D, std=ctrl)
    sd["condition_mlp.cap_proj.fc1.weight"] = r.normal(D, D, std=ctrl)
    sd["condition_mlp.cap_proj.fc2.weight"] = r.normal(D, D, std=ctrl)
    for k in range(3):
        sd[f"condition_layers.{k}.fc1.weight"] = r.normal(D, D, std=ctrl)
        sd[f"condition_layers.{k}.fc2.weight"] = r.normal(D, D, std=ctrl)
    for i in range(cfg.n_layer):
        p = f"layers.{i}."
        sd[p + "attention.wqkv.weight"] = r.normal(3 * D, D, std=lin)
        sd[p + "attention.wo.weight"] = r.normal(D, D, std=lin)
        sd[p + "feed_forward.w1.weight"] = r.normal(H, D, std=lin)
        sd[p + "feed_forward.w3.weight"] = r.normal(H, D, std=lin)
        sd[p + "feed_forward.w2.weight"] = r.normal(D, H, std=lin)
        sd[p + "attention_norm.weight"] = r.normal(D, std=0.1, mean=1.0)
        sd[p + "ffn_norm.weight"] = r.normal(D, std=0.1, mean=1.0)
    sd["norm.weight"] = r.normal(D, std=0.1, mean=1.0)
    sd["output.weight"] = r.normal(cfg.vocab_size, D, std=0.2)
    return sd


def vit16_state_dict(cfg: ViTConfig, seed: int = 1, prefix: str = "adapter.model.") -> Dict[str, torch.Tensor]:
    """Names follow HF ViTModel.state_dict() (transformers 5.15.0 modeling_vit.py: layers.{i}.attention.{q,k,v,o}_proj ...)."""
    r = _Rng(seed)
    D, M = cfg.hidden, cfg.mlp
    n_pos = cfg.pos_grid * cfg.pos_grid + 1
    sd: Dict[str, torch.Tensor] = {}
    e = prefix + "embeddings."
    sd[e + "cls_token"] = r.normal(1, 1, D, std=0.02)
    sd[e + "position_embeddings"] = r.normal(1, n_pos, D, std=0.2)
    fan_in = 3 * cfg.patch * cfg.patch
    sd[e + "patch_embeddings.projection.weight"] = r.normal(D, 3, cfg.patch, cfg.patch, std=fan_in ** -0.5)
    sd[e + "patch_embeddings.projection.bias"] = r.normal(D, std=0.02)
    for i in range(cfg.layers):
        p = f"{prefix}layers.{i}."
        sd[p + "layernorm_before.weight"] = r.normal(D, std=0.1, mean=1.0)
        sd[p + "layernorm_before.bias"] = r.normal(D, std=0.05)
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"attention.{n}.weight"] = r.normal(D, D, std=(0.7 if n == "o_proj" else 1.0) * D ** -0.5)
            sd[p + f"attention.{n}.bias"] = r.normal(D, std=0.02)
        sd[p + "layernorm_after.weight"] = r.normal(D, std=0.1, mean=1.0)
        sd[p + "layernorm_after.bias"] = r.normal(D, std=0.05)
        sd[p + "mlp.fc1.weight"] = r.normal(M, D, std=D ** -0.5)
        sd[p + "mlp.fc1.bias"] = r.normal(M, std=0.02)
        sd[p + "mlp.fc2.weight"] = r.normal(D, M, std=0.7 * M ** -0.5)
        sd[p + "mlp.fc2.bias"] = r.normal(D, std=0.02)
    sd[prefix + "layernorm.weight"] = r.normal(D, std=0.1, mean=1.0)
    sd[prefix + "layernorm.bias"] = r.normal(D, std=0.05)
    sd[prefix + "pooler.dense.weight"] = r.normal(D, D, std=D ** -0.5)      # present in the checkpoint, unused on the path
    sd[prefix + "pooler.dense.bias"] = torch.zeros(D)
    return sd


def vit_state_dict(cfg: ViTConfig, seed: int = 1, prefix: str = "adapter.model.") -> Dict[str, torch.Tensor]:
    """Names follow HF Dinov2Model.state_dict() (transformers 5.15.0 modeling_dinov2.py)."""
    if cfg.variant == "vit":
        return vit16_state_dict(cfg, seed, prefix)
    r = _Rng(seed)
    D, M = cfg.hidden, cfg.mlp
    n_pos = cfg.pos_grid * cfg.pos_grid + 1
    sd: Dict[str, torch.Tensor] = {}
    e = prefix + "embeddings."
    sd[e + "cls_token"] = r.normal(1, 1, D, std=0.02)
    sd[e + "mask_token"] = torch.zeros(1, D)
    sd[e + "position_embeddings"] = r.normal(1, n_pos, D, std=0.2)
    fan_in = 3 * cfg.patch * cfg.patch
    sd[e + "patch_embeddings.projection.weight"] = r.normal(D, 3, cfg.patch, cfg.patch, std=fan_in ** -0.5)
    sd[e + "patch_embeddings.projection.bias"] = r.normal(D, std=0.02)
    for i in range(cfg.layers):
        p = f"{prefix}encoder.layer.{i}."
        sd[p + "norm1.weight"] = r.normal(D, std=0.1, mean=1.0)
        sd[p + "norm1.bias"] = r.normal(D, std=0.05)
        for n in ("query", "key", "value"):
            sd[p + f"attention.attention.{n}.weight"] = r.normal(D, D, std=D ** -0.5)
            sd[p + f"attention.attention.{n}.bias"] = r.normal(D, std=0.02)
        sd[p + "attention.output.dense.weight"] = r.normal(D, D, std=D ** -0.5)
        sd[p + "attention.output.dense.bias"] = r.normal(D, std=0.02)
        sd[p + "layer_scale1.lambda1"] = r.normal(D, std=0.05, mean=0.5)
        sd[p + "norm2.weight"] = r.normal(D, std=0.1, mean=1.0)
        sd[p + "norm2.bias"] = r.normal(D, std=0.05)
        sd[p + "mlp.fc1.weight"] = r.normal(M, D, std=D ** -0.5)
        sd[p + "mlp.fc1.bias"] = r.normal(M, std=0.02)
        sd[p + "mlp.fc2.weight"] = r.normal(D, M, std=M ** -0.5)
        sd[p + "mlp.fc2.bias"] = r.normal(D, std=0.02)
        sd[p + "layer_scale2.lambda1"] = r.normal(D, std=0.05, mean=0.5)
    sd[prefix + "layernorm.weight"] = r.normal(D, std=0.1, mean=1.0)
    sd[prefix + "layernorm.bias"] = r.normal(D, std=0.05)
    return sd


def _conv(r: _Rng, sd, name, cout, cin, k):
    bound = 1.0 / math.sqrt(cin * k * k)
    # gain ~sqrt(3): keeps activations O(1) through 30 conv layers with swish
    sd[name + ".weight"] = r.uniform(cout, cin, k, k, lo=-bound, hi=bound) * 1.7
    sd[name + ".bias"] = r.uniform(cout, lo=-bound, hi=bound)


def _gn(r: _Rng, sd, name, c):
    sd[name + ".weight"] = r.normal(c, std=0.1, mean=1.0)
    sd[name + ".bias"] = r.normal(c, std=0.1)


def vq_decoder_layout(cfg: VQConfig):
    """Yields the decode-side module list in execution order (reference vq_model.py:129-195).
    Items: ("res", name, cin, cout) | ("attn", name, c) | ("up", name, c)."""
    nres = len(cfg.ch_mult)
    block_in = cfg.ch * cfg.ch_mult[nres - 1]
    out = [("res", "decoder.mid.0", block_in, block_in), ("attn", "decoder.mid.1", block_in),
           ("res", "decoder.mid.2", block_in, block_in)]
    for idx, i_level in enumerate(reversed(range(nres))):
        block_out = cfg.ch * cfg.ch_mult[i_level]
        for j in range(cfg.num_res_blocks + 1):
            out.append(("res", f"decoder.conv_blocks.{idx}.res.{j}", block_in, block_out))
            block_in = block_out
            if i_level == nres - 1:
                out.append(("attn", f"decoder.conv_blocks.{idx}.attn.{j}", block_in))
        if i_level != 0:
            out.append(("up", f"decoder.conv_blocks.{idx}.upsample", block_in))
    return out, block_in


def vq_encoder_layout(cfg: VQConfig):
    """Encoder module list in execution order (reference vq_model.py:62-126).
    Items: ("res", name, cin, cout) | ("attn", name, c) | ("down", name, c)."""
    nres = len(cfg.ch_mult)
    in_mult = (1,) + tuple(cfg.ch_mult)
    out = []
    block_in = cfg.ch
    for i_level in range(nres):
        block_in = cfg.ch * in_mult[i_level]
        block_out = cfg.ch * cfg.ch_mult[i_level]
        for j in range(cfg.num_res_blocks):
            out.append(("res", f"encoder.conv_blocks.{i_level}.res.{j}", block_in, block_out))
            block_in = block_out
            if i_level == nres - 1:
                out.append(("attn", f"encoder.conv_blocks.{i_level}.attn.{j}", block_in))
        if i_level != nres - 1:
            out.append(("down", f"encoder.conv_blocks.{i_level}.downsample", block_in))
    out += [("res", "encoder.mid.0", block_in, block_in), ("attn", "encoder.mid.1", block_in), ("res", "encoder.mid.2", block_in, block_in)]
    return out, block_in


def vq_state_dict(cfg: VQConfig, seed: int = 2) -> Dict[str, torch.Tensor]:
    """Decode-side names of VQModel.state_dict() (reference vq_model.py:28-39,129-169)."""
    r = _Rng(seed)
    sd: Dict[str, torch.Tensor] = {}
    cb = r.uniform(cfg.codebook_size, cfg.codebook_embed_dim, lo=-1.0, hi=1.0)
    sd["quantize.embedding.weight"] = torch.nn.functional.normalize(cb, p=2, dim=-1) * r.uniform(cfg.codebook_size, 1, lo=0.5, hi=1.5)
    _conv(r, sd, "post_quant_conv", cfg.z_channels, cfg.codebook_embed_dim, 1)
    layout, last = vq_decoder_layout(cfg)
    block_in = cfg.ch * cfg.ch_mult[-1]
    _conv(r, sd, "decoder.conv_in", block_in, cfg.z_channels, 3)
    for item in layout:
        if item[0] == "res":
            _, name, cin, cout = item
            _gn(r, sd, name + ".norm1", cin)
            _conv(r, sd, name + ".conv1", cout, cin, 3)
            _gn(r, sd, name + ".norm2", cout)
            _conv(r, sd, name + ".conv2", cout, cout, 3)
            if cin != cout:
                _conv(r, sd, name + ".nin_shortcut", cout, cin, 1)
        elif item[0] == "attn":
            _, name, c = item
            _gn(r, sd, name + ".norm", c)
            for n in ("q", "k", "v", "proj_out"):
                _conv(r, sd, f"{name}.{n}", c, c, 1)
        else:
            _, name, c = item
            _conv(r, sd, name + ".conv", c, c, 3)
    _gn(r, sd, "decoder.norm_out", last)
    _conv(r, sd, "decoder.conv_out", 3, last, 3)
    # ---- encode side (vq_model.py:62-126 Encoder, :39 quant_conv).  Appended AFTER the decode side so the decoder tensors
    # (and every golden minted from them) keep their values.
    enc, elast = vq_encoder_layout(cfg)
    _conv(r, sd, "encoder.conv_in", cfg.ch, 3, 3)
    for item in enc:
        if item[0] == "res":
            _, name, cin, cout = item
            _gn(r, sd, name + ".norm1", cin)
            _conv(r, sd, name + ".conv1", cout, cin, 3)
            _gn(r, sd, name + ".norm2", cout)
            _conv(r, sd, name + ".conv2", cout, cout, 3)
            if cin != cout:
                _conv(r, sd, name + ".nin_shortcut", cout, cin, 1)
        elif item[0] == "attn":
            _, name, c = item
            _gn(r, sd, name + ".norm", c)
            for n in ("q", "k", "v", "proj_out"):
                _conv(r, sd, f"{name}.{n}", c, c, 1)
        else:
            _, name, c = item
            _conv(r, sd, name + ".conv", c, c, 3)
    _gn(r, sd, "encoder.norm_out", elast)
    _conv(r, sd, "encoder.conv_out", cfg.z_channels, elast, 3)
    _conv(r, sd, "quant_conv", cfg.codebook_embed_dim, cfg.z_channels, 1)
    return sd


def path_state_dicts(cfg: PathConfig, seed: int = 0):
    gpt = gpt_state_dict(cfg.gpt, cfg.vit, seed=seed)
    gpt.update(vit_state_dict(cfg.vit, seed=seed + 1))
    vq = vq_state_dict(cfg.vq, seed=seed + 2)
    return gpt, vq


# ----------------------------------------------------------------------------- inputs
def canny_like_control(batch: int, H: int, W: int, seed: int = 1234, density: float = 0.08, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Binary edge-like map in {-1,+1}, 3 identical channels, as sample_t2i.py:123-125,141
    produces from cv2.Canny (uint8 {0,255} -> 2*(x/255-0.5)).  One seeded generator per image (image i of any batch is the
    same map); written straight into the output in `dtype` (both values are exact in bf16)."""
    out = torch.empty(batch, 3, H, W, dtype=dtype)
    one, neg = torch.tensor(1.0, dtype=dtype), torch.tensor(-1.0, dtype=dtype)
    for i in range(batch):
        g = torch.Generator().manual_seed(seed + i)
        m = torch.rand(H, W, generator=g) > (1.0 - density)
        out[i] = torch.where(m, one, neg)            # broadcast over the 3 channels
    return out


def smooth_control(batch: int, H: int, W: int, seed: int = 1234) -> torch.Tensor:
    """Smooth map in [-1,1] (depth / soft-edge stand-in): low-pass filtered noise, 3 channels."""
    out = []
    for i in range(batch):
        g = torch.Generator().manual_seed(seed + i)
        lo = torch.rand(1, 1, max(H // 32, 2), max(W // 32, 2), generator=g)
        m = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=True)[0, 0]
        out.append((2.0 * m - 1.0)[None].repeat(3, 1, 1))
    return torch.stack(out)


def class_labels(batch: int, num_classes: int, seed: int = 1234) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, num_classes, (batch,), generator=g, dtype=torch.int64)


def text_embeddings_with_lengths(lengths, T: int = 120, caption_dim: int = 2048, seed: int = 1234):
    """As text_embeddings but with explicit valid lengths (edge cases: 1 = a single real token, T = no padding at all)."""
    embs, masks = [], []
    for i, L in enumerate(lengths):
        g = torch.Generator().manual_seed(seed + i)
        e = 0.2 * torch.randn(T, caption_dim, generator=g)
        m = torch.zeros(T, dtype=torch.int64)
        m[T - int(L):] = 1
        embs.append(e * m[:, None])
        masks.append(m)
    return torch.stack(embs), torch.stack(masks)


def text_embeddings(batch: int, T: int = 120, caption_dim: int = 2048, seed: int = 1234):
    """0.2*randn caption features, valid length L~U{8..40}, LEFT-padded exactly as
    sample_t2i.py:146-160: valid tokens occupy the last L slots, features multiplied by mask."""
    embs, masks = [], []
    for i in range(batch):
        g = torch.Generator().manual_seed(seed + i)
        e = 0.2 * torch.randn(T, caption_dim, generator=g)
        L = int(torch.randint(8, min(41, T + 1), (1,), generator=g).item())
        m = torch.zeros(T, dtype=torch.int64)
        m[T - L:] = 1
        embs.append(e * m[:, None])
        masks.append(m)
    return torch.stack(embs), torch.stack(masks)


def t5_state_dict(cfg: T5Config, seed: int = 7) -> Dict[str, torch.Tensor]:
    """HF T5EncoderModel.state_dict() names (shared.weight, encoder.block.N.layer.{0,1}.*, encoder.final_layer_norm.weight).
    Scales follow T5PreTrainedModel._init_weights (q: (d_model*d_kv)^-0.5 because attention is unscaled; k, v, wi: d_model^-0.5;
    o: (heads*d_kv)^-0.5; wo: d_ff^-0.5) except that norms are 1 + 0.1 N and the relative bias is N(0, 0.5^2) so both matter."""
    r = _Rng(seed)
    D, kv, H, F = cfg.d_model, cfg.d_kv, cfg.num_heads, cfg.d_ff
    inner = H * kv
    sd = {"shared.weight": r.normal(cfg.vocab_size, D)}
    for i in range(cfg.num_layers):
        a = f"encoder.block.{i}.layer.0."
        sd[a + "SelfAttention.q.weight"] = r.normal(inner, D, std=(D * kv) ** -0.5 * 2.0)
        sd[a + "SelfAttention.k.weight"] = r.normal(inner, D, std=D ** -0.5)
        sd[a + "SelfAttention.v.weight"] = r.normal(inner, D, std=D ** -0.5)
        sd[a + "SelfAttention.o.weight"] = r.normal(D, inner, std=inner ** -0.5)
        if i == 0:
            sd[a + "SelfAttention.relative_attention_bias.weight"] = r.normal(cfg.relative_attention_num_buckets, H, std=0.5)
        sd[a + "layer_norm.weight"] = 1.0 + 0.1 * r.normal(D)
        f = f"encoder.block.{i}.layer.1."
        sd[f + "DenseReluDense.wi_0.weight"] = r.normal(F, D, std=D ** -0.5)
        sd[f + "DenseReluDense.wi_1.weight"] = r.normal(F, D, std=D ** -0.5)
        sd[f + "DenseReluDense.wo.weight"] = r.normal(D, F, std=F ** -0.5)
        sd[f + "layer_norm.weight"] = 1.0 + 0.1 * r.normal(D)
    sd["encoder.final_layer_norm.weight"] = 1.0 + 0.1 * r.normal(D)
    return sd


def t5_tokens(batch: int, cfg: T5Config, seed: int = 4321, lengths=None):
    """Tokenizer-shaped inputs: ids RIGHT-padded with 0 to model_max_length (HF T5 tokenizer pads right; the reference left-pads
    the EMBEDDINGS afterwards, sample_t2i.py:146-160), last valid id = 1 (</s>), attention_mask 1 on valid positions."""
    T = cfg.model_max_length
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(batch, T, dtype=torch.int64)
    mask = torch.zeros(batch, T, dtype=torch.int64)
    for b in range(batch):
        L = int(lengths[b]) if lengths is not None else int(torch.randint(4, min(41, T + 1), (1,), generator=g).item())
        ids[b, :L - 1] = torch.randint(2, cfg.vocab_size, (L - 1,), generator=g)
        ids[b, L - 1] = 1
        mask[b, :L] = 1
    return ids, mask
