"""CPU oracle for the ControlAR conditional-decoding hot path.  TEST INFRASTRUCTURE ONLY.

A functional restatement (torch CPU tensors used as the array library; no nn.Module, no HF
transformers, no import from /root/reference) of the reference algorithm for the path
  DINOv2 control encoder -> LlamaGen AR decode with per-token control fusion -> VQGAN decoder.
Every function cites the reference file:line it follows.  ``HF:`` = transformers 5.15.0
``models/dinov2/modeling_dinov2.py`` (third-party arithmetic pinned by SURVEY.md §8c).

Parity pin: the reference itself holds NO golden vectors for this path (SURVEY.md §4/§8c:
"parity unpinned" by the reference's own tests).  This oracle is therefore pinned against
outputs of the unmodified reference modules run in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product (controlar_amd) never does: it fails loudly if the HIP library is missing.

``dtype=torch.float32`` is the thread-stable ground truth ("exact" contract);
``dtype=torch.bfloat16`` reproduces the reference's default --precision bf16 rounding points
 because the same torch CPU element-wise/matmul kernels round once per op.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# resize in front of the encoder   (reference: autoregressive/models/dinov2_adapter.py:16-24)
# --------------------------------------------------------------------------------------
def nearest_src_index(out_size: int, in_size: int) -> Tensor:
    """F.interpolate(mode='nearest') source index: floor(dst * (in/out)) computed in fp32,
    clamped to in-1 (ATen nearest_neighbor_compute_source_index)."""
    scale = torch.tensor(float(in_size) / float(out_size), dtype=torch.float32)
    dst = torch.arange(out_size, dtype=torch.float32)
    return torch.clamp(torch.floor(dst * scale).to(torch.int64), max=in_size - 1)


def _cubic_weights(t: Tensor) -> Tensor:
    """Cubic-convolution coefficients, A = -0.75 (ATen get_cubic_upsample_coefficients)."""
    A = -0.75
    x0 = t + 1.0
    w0 = ((A * x0 - 5 * A) * x0 + 8 * A) * x0 - 4 * A
    w1 = ((A + 2) * t - (A + 3)) * t * t + 1
    x2 = 1.0 - t
    w2 = ((A + 2) * x2 - (A + 3)) * x2 * x2 + 1
    x3 = 2.0 - t
    w3 = ((A * x3 - 5 * A) * x3 + 8 * A) * x3 - 4 * A
    return torch.stack([w0, w1, w2, w3], dim=-1)


def bicubic_table(out_size: int, in_size: int, align_corners: bool) -> Tuple[Tensor, Tensor]:
    """Per-output-index 4 clamped source indices and 4 fp32 weights of F.interpolate(bicubic)."""
    dst = torch.arange(out_size, dtype=torch.float32)
    if align_corners:
        scale = (in_size - 1) / (out_size - 1) if out_size > 1 else 0.0
        src = dst * torch.tensor(scale, dtype=torch.float32)
    else:
        scale = in_size / out_size
        src = (dst + 0.5) * torch.tensor(scale, dtype=torch.float32) - 0.5
    ix = torch.floor(src)
    t = src - ix
    idx = torch.stack([torch.clamp(ix.to(torch.int64) - 1 + k, 0, in_size - 1) for k in range(4)], dim=-1)
    return idx, _cubic_weights(t)


def bicubic_resize(x: Tensor, out_h: int, out_w: int, align_corners: bool) -> Tensor:
    """Separable bicubic on [..., H, W] in fp32 (x then y, as ATen's nested cubic_interp1d)."""
    H, W = x.shape[-2:]
    iy, wy = bicubic_table(out_h, H, align_corners)
    ix, wx = bicubic_table(out_w, W, align_corners)
    xf = x.float()
    gx = xf[..., :, ix]                                   # [..., H, out_w, 4]
    rows = (gx * wx).sum(-1)                              # [..., H, out_w]
    gy = rows[..., iy, :]                                 # [..., out_h, 4, out_w]
    return (gy * wy[:, :, None]).sum(-2)


def to_patch14(img: Tensor, condition_type: str, patch: int = 14) -> Tensor:
    """reference: dinov2_adapter.py:16-24 — (H//16*14, W//16*14); nearest for canny/seg,
    bicubic align_corners=True otherwise.  Output keeps the input dtype."""
    H, W = img.shape[2:]
    nh, nw = (H // 16) * patch, (W // 16) * patch
    if condition_type in ("canny", "seg"):
        iy, ix = nearest_src_index(nh, H), nearest_src_index(nw, W)
        return img[:, :, iy][:, :, :, ix]
    return bicubic_resize(img, nh, nw, align_corners=True).to(img.dtype)


# --------------------------------------------------------------------------------------
# DINOv2 encoder   (HF: modeling_dinov2.py:38-149 embeddings, :182-235 attention, :342-381 layer, :451-469)
# --------------------------------------------------------------------------------------
def dinov2_pos_embed(sd: Dict[str, Tensor], prefix: str, gh: int, gw: int, dtype) -> Tensor:
    """HF: Dinov2Embeddings.interpolate_pos_encoding (:57-95): bicubic, align_corners=False,
    fp32, to size (gh, gw); skipped when the grid equals the native square grid."""
    pe = sd[prefix + "embeddings.position_embeddings"]
    n = pe.shape[1] - 1
    g = int(round(math.sqrt(n)))
    if gh * gw == n and gh == gw:
        return pe.to(dtype)
    cls_pe, patch_pe = pe[:, :1], pe[:, 1:]
    D = pe.shape[-1]
    grid = patch_pe.reshape(1, g, g, D).permute(0, 3, 1, 2).float()
    out = bicubic_resize(grid, gh, gw, align_corners=False).to(pe.dtype)
    out = out.permute(0, 2, 3, 1).reshape(1, gh * gw, D)
    return torch.cat([cls_pe, out], dim=1).to(dtype)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w.to(x.dtype), b.to(x.dtype), eps)


def dinov2_forward(sd: Dict[str, Tensor], vit, x: Tensor, prefix: str = "adapter.model.") -> Tensor:
    """HF Dinov2Model.forward -> last_hidden_state (CLS kept).  x: [B,3,H,W] already resized."""
    dtype = x.dtype
    B, _, H, W = x.shape
    p = vit.patch
    gh, gw = H // p, W // p
    D, nh = vit.hidden, vit.heads
    hd = D // nh
    # patch conv 14x14 stride 14 == matmul over unfolded patches (HF :119-149)
    w = sd[prefix + "embeddings.patch_embeddings.projection.weight"].to(dtype).reshape(D, -1)
    bias = sd[prefix + "embeddings.patch_embeddings.projection.bias"].to(dtype)
    patches = x.reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, 3 * p * p)
    tok = F.linear(patches, w, bias)
    cls = sd[prefix + "embeddings.cls_token"].to(dtype).expand(B, -1, -1)
    h = torch.cat([cls, tok], dim=1) + dinov2_pos_embed(sd, prefix, gh, gw, dtype)
    for i in range(vit.layers):
        q = f"{prefix}encoder.layer.{i}."
        g = lambda n: sd[q + n].to(dtype)  # noqa: E731
        y = layer_norm(h, g("norm1.weight"), g("norm1.bias"), vit.ln_eps)
        T = y.shape[1]
        qh = F.linear(y, g("attention.attention.query.weight"), g("attention.attention.query.bias")).view(B, T, nh, hd).transpose(1, 2)
        kh = F.linear(y, g("attention.attention.key.weight"), g("attention.attention.key.bias")).view(B, T, nh, hd).transpose(1, 2)
        vh = F.linear(y, g("attention.attention.value.weight"), g("attention.attention.value.bias")).view(B, T, nh, hd).transpose(1, 2)
        att = torch.softmax((qh.float() @ kh.float().transpose(-1, -2)) * hd ** -0.5, dim=-1)
        ctx = (att @ vh.float()).to(dtype).transpose(1, 2).reshape(B, T, D)
        a = F.linear(ctx, g("attention.output.dense.weight"), g("attention.output.dense.bias"))
        h = a * g("layer_scale1.lambda1") + h
        y = layer_norm(h, g("norm2.weight"), g("norm2.bias"), vit.ln_eps)
        m = F.linear(F.gelu(F.linear(y, g("mlp.fc1.weight"), g("mlp.fc1.bias"))), g("mlp.fc2.weight"), g("mlp.fc2.bias"))
        h = m * g("layer_scale2.lambda1") + h
    return layer_norm(h, sd[prefix + "layernorm.weight"].to(dtype), sd[prefix + "layernorm.bias"].to(dtype), vit.ln_eps)


def vit16_forward(sd: Dict[str, Tensor], vit, x: Tensor, prefix: str = "adapter.model.") -> Tensor:
    """HF ViTModel.forward(x, interpolate_pos_encoding=True) -> last_hidden_state (transformers 5.15.0
    models/vit/modeling_vit.py:75-160 embeddings + bicubic pos-emb interpolation, :257-287 ViTLayer —
    pre-LN, no LayerScale, erf-GELU MLP, final LayerNorm :348,:384).  Called by vit_adapter.py:13-15."""
    dtype = x.dtype
    B, _, H, W = x.shape
    p = vit.patch
    gh, gw = H // p, W // p
    D, nh = vit.hidden, vit.heads
    hd = D // nh
    w = sd[prefix + "embeddings.patch_embeddings.projection.weight"].to(dtype).reshape(D, -1)
    bias = sd[prefix + "embeddings.patch_embeddings.projection.bias"].to(dtype)
    patches = x.reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, 3 * p * p)
    tok = F.linear(patches, w, bias)
    cls = sd[prefix + "embeddings.cls_token"].to(dtype).expand(B, -1, -1)
    h = torch.cat([cls, tok], dim=1) + dinov2_pos_embed(sd, prefix, gh, gw, dtype)     # same bicubic(align_corners=False) rule
    for i in range(vit.layers):
        q = f"{prefix}layers.{i}."
        g = lambda n: sd[q + n].to(dtype)  # noqa: E731
        y = layer_norm(h, g("layernorm_before.weight"), g("layernorm_before.bias"), vit.ln_eps)
        T = y.shape[1]
        qh = F.linear(y, g("attention.q_proj.weight"), g("attention.q_proj.bias")).view(B, T, nh, hd).transpose(1, 2)
        kh = F.linear(y, g("attention.k_proj.weight"), g("attention.k_proj.bias")).view(B, T, nh, hd).transpose(1, 2)
        vh = F.linear(y, g("attention.v_proj.weight"), g("attention.v_proj.bias")).view(B, T, nh, hd).transpose(1, 2)
        att = torch.softmax((qh.float() @ kh.float().transpose(-1, -2)) * hd ** -0.5, dim=-1)
        ctx = (att @ vh.float()).to(dtype).transpose(1, 2).reshape(B, T, D)
        h = F.linear(ctx, g("attention.o_proj.weight"), g("attention.o_proj.bias")) + h
        y = layer_norm(h, g("layernorm_after.weight"), g("layernorm_after.bias"), vit.ln_eps)
        h = F.linear(F.gelu(F.linear(y, g("mlp.fc1.weight"), g("mlp.fc1.bias"))), g("mlp.fc2.weight"), g("mlp.fc2.bias")) + h
    return layer_norm(h, sd[prefix + "layernorm.weight"].to(dtype), sd[prefix + "layernorm.bias"].to(dtype), vit.ln_eps)


def control_encoder(sd, cfg, img: Tensor) -> Tensor:
    """t2i — dinov2_adapter.py:26-29: resize, DINOv2, drop CLS.  c2i — vit_adapter.py:13-15: HF ViT-S/16 on the
    unresized map with interpolated position embeddings, drop CLS.  -> [B, (H/16)(W/16), vit.hidden]"""
    if getattr(cfg.vit, "variant", "dinov2") == "vit":
        return vit16_forward(sd, cfg.vit, img)[:, 1:]
    x = to_patch14(img, cfg.gpt.condition_type, cfg.vit.patch)
    return dinov2_forward(sd, cfg.vit, x)[:, 1:]


# --------------------------------------------------------------------------------------
# LlamaGen pieces   (reference: autoregressive/models/gpt_t2i.py)
# --------------------------------------------------------------------------------------
def mlp(x: Tensor, fc1: Tensor, fc2: Tensor) -> Tensor:
    """reference: gpt_t2i.py:165-181 — fc2(gelu_tanh(fc1 x)), bias-free."""
    return F.linear(F.gelu(F.linear(x, fc1.to(x.dtype)), approximate="tanh"), fc2.to(x.dtype))


def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """reference: gpt_t2i.py:193-198 — fp32 normalise, cast to x.dtype, then * weight."""
    xf = x.float()
    n = (xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)).type_as(x)
    return n * w.to(x.dtype)


def rope_table_2d(grid: int, head_dim: int, base: float, cls_token_num: int) -> Tensor:
    """reference: gpt_t2i.py:506-519 — [cls+grid^2, head_dim/2, 2] fp32; first cls rows are ZERO."""
    half = head_dim // 2
    freqs = 1.0 / (base ** (torch.arange(0, half, 2)[: half // 2].float() / half))
    t = torch.arange(grid, dtype=torch.float32)
    fr = torch.outer(t, freqs)
    fg = torch.cat([fr[:, None, :].expand(-1, grid, -1), fr[None, :, :].expand(grid, -1, -1)], dim=-1)
    cache = torch.stack([torch.cos(fg), torch.sin(fg)], dim=-1).flatten(0, 1)
    return torch.cat([torch.zeros(cls_token_num, head_dim // 2, 2), cache])


def apply_rope(x: Tensor, fc: Tensor) -> Tensor:
    """reference: gpt_t2i.py:522-532 — x [b,s,h,d]; fc [s,d/2,2]; adjacent (2i,2i+1) pairs, fp32."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    fc = fc.view(1, xs.size(1), 1, xs.size(3), 2)
    out = torch.stack([xs[..., 0] * fc[..., 0] - xs[..., 1] * fc[..., 1],
                       xs[..., 1] * fc[..., 0] + xs[..., 0] * fc[..., 1]], dim=-1)
    return out.flatten(3).type_as(x)


class GPTState:
    """KV caches + cached control tokens for one generate() call
    (reference: gpt_t2i.py:220-235 KVCache, :391-405 setup_caches, :433-442 control cache)."""

    def __init__(self, sd, cfg, b: int, s_max: int, dtype):
        g = cfg.gpt
        self.sd, self.g, self.dtype, self.b = sd, g, dtype, b
        self.s_max = s_max
        self.k = [torch.zeros(b, g.n_head, s_max, g.head_dim, dtype=dtype) for _ in range(g.n_layer)]
        self.v = [torch.zeros(b, g.n_head, s_max, g.head_dim, dtype=dtype) for _ in range(g.n_layer)]
        self.rope = rope_table_2d(g.grid, g.head_dim, g.rope_base, g.cls_token_num)
        self.mask = torch.tril(torch.ones(s_max, s_max, dtype=torch.bool)).unsqueeze(0).repeat(b, 1, 1)
        self.ctrl: Optional[List[Tensor]] = None
        self.control_strength = 1.0
        # model of the library's W8A8 decode mode (car_config.decode_weight_fp8 = 2; NOT a reference feature): the inputs of the five
        # decode linears are rounded to OCP e4m3 (unit scale, clamped to +-448) on single-token steps; the prefill stays as is
        self.act_fp8_decode = False
        # model of the library's opt-in e4m3 KV cache (car_config.kv_cache_fp8; NOT a reference feature): rotated K and V are rounded to OCP e4m3
        # (unit scale, clamped to +-448) when they are STORED; the prefill's own attention runs on the unrounded rows it has just computed, every
        # later step reads the rounded ones
        self.kv_fp8 = False

    def lin(self, x: Tensor, name: str) -> Tensor:
        if self.act_fp8_decode and x.shape[1] == 1:
            x = x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype)
        return F.linear(x, self.w(name))

    def w(self, name):
        return self.sd[name].to(self.dtype)


def fold_pad_mask(st: GPTState, emb_masks: Tensor, T: int):
    """reference: generate.py:184-193 — text-pad columns masked, diagonal forced on."""
    st.mask[:, :, :T] = st.mask[:, :, :T] * emb_masks.to(torch.bool).unsqueeze(1)
    eye = torch.eye(st.s_max, dtype=torch.bool)
    st.mask[:] = st.mask | eye


def transformer_forward(st: GPTState, h: Tensor, input_pos: Tensor) -> Tensor:
    """reference: gpt_t2i.py:446-470 (inference branches) + TransformerBlock :294-307,
    Attention :254-291, FeedForward :216-217.  h: [b,s,dim]; returns fp32 logits [b,s,V]."""
    g = st.g
    b, s, D = h.shape
    mask = st.mask[:b, None, input_pos]                     # [b,1,s,S]
    fc = st.rope[input_pos]
    for i in range(g.n_layer):
        if i % g.layer_internal == 0 and st.ctrl is not None:
            c = st.ctrl[i // g.layer_internal]
            if s > 1:    # prefill: control token 0 onto the LAST prefix row (:463)
                h = h.clone()
                h[:, -1:] = h[:, -1:] + st.control_strength * c[:, 0:1]
            else:        # decode at position p: control token p - T + 1 (:466)
                h = h + st.control_strength * c[:, input_pos - g.cls_token_num + 1]
        p = f"layers.{i}."
        x = rms_norm(h, st.w(p + "attention_norm.weight"), g.norm_eps)
        qkv = st.lin(x, p + "attention.wqkv.weight")
        xq, xk, xv = qkv.split([D, D, D], dim=-1)
        xq = apply_rope(xq.view(b, s, g.n_head, g.head_dim), fc).transpose(1, 2)
        xk = apply_rope(xk.view(b, s, g.n_head, g.head_dim), fc).transpose(1, 2)
        xv = xv.view(b, s, g.n_head, g.head_dim).transpose(1, 2)
        if st.kv_fp8:      # library-side model, not the reference: K / V are stored as e4m3; the prefill attends to the rows it has just computed
            rq = lambda t: t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype)
            st.k[i][:, :, input_pos] = rq(xk)
            st.v[i][:, :, input_pos] = rq(xv)
            kk, vv = st.k[i], st.v[i]
            if s > 1:
                kk, vv = kk.clone(), vv.clone()
                kk[:, :, input_pos] = xk; vv[:, :, input_pos] = xv
        else:
            st.k[i][:, :, input_pos] = xk
            st.v[i][:, :, input_pos] = xv
            kk, vv = st.k[i], st.v[i]
        sc = (xq.float() @ kk.float().transpose(-1, -2)) * (g.head_dim ** -0.5)
        sc = sc.masked_fill(~mask, float("-inf"))
        att = (torch.softmax(sc, dim=-1) @ vv.float()).to(h.dtype)
        att = att.transpose(1, 2).reshape(b, s, D)
        h = h + st.lin(att, p + "attention.wo.weight")
        x = rms_norm(h, st.w(p + "ffn_norm.weight"), g.norm_eps)
        ff = st.lin(F.silu(st.lin(x, p + "feed_forward.w1.weight")) * st.lin(x, p + "feed_forward.w3.weight"), p + "feed_forward.w2.weight")
        h = h + ff
    h = rms_norm(h, st.w("norm.weight"), g.norm_eps)
    return st.lin(h, "output.weight").float()


def top_k_top_p_filtering(logits: Tensor, top_k: int = 0, top_p: float = 1.0) -> Tensor:
    """reference: generate.py:17-56."""
    if top_k > 0:
        top_k = min(max(top_k, 1), logits.size(-1))
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits = logits.masked_fill(logits < kth, -float("inf"))
    if top_p < 1.0:
        sl, si = torch.sort(logits, descending=True)
        cp = torch.cumsum(F.softmax(sl, dim=-1), dim=-1)
        rm = cp > top_p
        rm[..., 1:] = rm[..., :-1].clone()
        rm[..., 0] = 0
        logits = logits.masked_fill(rm.scatter(1, si, rm), -float("inf"))
    return logits


def sample(logits_last: Tensor, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False,
           generator: Optional[torch.Generator] = None) -> Tensor:
    """reference: generate.py:59-74 — greedy = topk(softmax, 1): ties -> lowest index."""
    lg = logits_last / max(temperature, 1e-5)
    if top_k > 0 or top_p < 1.0:
        lg = top_k_top_p_filtering(lg, top_k, top_p)
    probs = F.softmax(lg, dim=-1)
    if sample_logits:
        return torch.multinomial(probs, 1, generator=generator)
    return torch.topk(probs, 1, dim=-1)[1]


def generate(sd, cfg, cond: Tensor, max_new_tokens: int, emb_masks: Optional[Tensor] = None,
             cfg_scale: float = 1.0, cfg_interval: int = -1, condition: Optional[Tensor] = None,
             control_strength: float = 1.0, dtype=torch.float32, forced_tokens: Optional[Tensor] = None,
             return_logits: bool = False, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False,
             generator=None, return_stages: bool = False, act_fp8_decode: bool = False, kv_fp8: bool = False):
    """reference: generate.py:134-204 (t2i branch) incl. prefill :85-94, decode_one_token :97-110,
    decode_n_tokens :113-131.  ``forced_tokens`` [B,N] switches to the teacher-forced protocol of
    SURVEY.md Appendix G (token fed back = forced token; logits still recorded)."""
    g = cfg.gpt
    c2i = g.model_type == "c2i"
    # c2i (reference gpt.py + generate.py:139-154): cond = int64 class labels [B]; prefix length 1 (condition_token_nums = 0,
    # sample_c2i.py:55); CFG null class = num_classes; no pad mask; no control_strength.  gpt.py:427 hard-casts the control
    # tokens to bf16 (the reference cannot run c2i in fp32); this restatement keeps `dtype` throughout.
    B = cond.shape[0]
    T = 1 if c2i else cond.shape[1]
    stages = {}
    if not c2i:
        cond = cond.to(dtype)
    ctrl_in = None
    if condition is not None:
        a = control_encoder(sd, cfg, condition.to(dtype))
        ctrl_in = mlp(a, sd["adapter_mlp.fc1.weight"], sd["adapter_mlp.fc2.weight"])         # generate.py:136-138
        stages["adapter_out"], stages["adapter_mlp_out"] = a, ctrl_in
    use_cfg = cfg_scale > 1.0
    if use_cfg:                                                                           # generate.py:140-146 / :155-164
        if c2i:
            cond = torch.cat([cond, torch.ones_like(cond) * g.num_classes])
        else:
            cond = torch.cat([cond, torch.zeros_like(cond) + sd["cls_embedding.uncond_embedding"].to(dtype)])
        if ctrl_in is not None:
            ctrl_in = torch.cat([ctrl_in, torch.zeros_like(ctrl_in)])
    b = cond.shape[0]
    s_max = ((T + max_new_tokens + 7) // 8) * 8                                           # gpt_t2i.py:395
    st = GPTState(sd, cfg, b, s_max, dtype)
    st.act_fp8_decode = act_fp8_decode
    st.kv_fp8 = kv_fp8
    if emb_masks is not None:
        fold_pad_mask(st, torch.cat([emb_masks, emb_masks]) if use_cfg else emb_masks, T)
    # ---- prefill (gpt_t2i.py:433-442): text embed + control-token cache
    st.control_strength = (control_strength if use_cfg else 1.0) if not c2i else 1.0      # generate.py:87-92 quirk
    if c2i:     # LabelEmbedder.forward (gpt.py:89-96)
        h = sd["cls_embedding.embedding_table.weight"].to(dtype)[cond.long()].unsqueeze(1)
    else:
        h = mlp(cond, sd["cls_embedding.cap_proj.fc1.weight"], sd["cls_embedding.cap_proj.fc2.weight"])[:, :g.cls_token_num]
    if ctrl_in is not None:
        ce = mlp(ctrl_in, sd["condition_mlp.cap_proj.fc1.weight"], sd["condition_mlp.cap_proj.fc2.weight"])
        st.ctrl = [mlp(ce, sd[f"condition_layers.{k}.fc1.weight"], sd[f"condition_layers.{k}.fc2.weight"]) for k in range(3)]
        stages["ctrl"] = st.ctrl
    logits = transformer_forward(st, h, torch.arange(T))

    def mix(lg, flag=True):
        if use_cfg:
            c, u = torch.split(lg, b // 2, dim=0)
            return u + (c - u) * cfg_scale if flag else c
        return lg

    all_logits = []
    lg = mix(logits)[:, -1]
    all_logits.append(lg)
    nxt = sample(lg, temperature, top_k, top_p, sample_logits, generator)
    toks = [nxt]
    cfg_flag = True
    for i in range(max_new_tokens - 1):
        if cfg_interval > -1 and i > cfg_interval:
            cfg_flag = False
        cur = forced_tokens[:, i:i + 1].to(torch.int64) if forced_tokens is not None else nxt
        x = torch.cat([cur, cur]) if use_cfg else cur
        h = sd["tok_embeddings.weight"].to(dtype)[x.view(-1)].view(b, 1, g.dim)
        logits = transformer_forward(st, h, torch.tensor([T + i]))
        lg = mix(logits, cfg_flag)[:, -1]
        all_logits.append(lg)
        nxt = sample(lg, temperature, top_k, top_p, sample_logits, generator)
        toks.append(nxt)
    out = torch.cat(toks, dim=1).to(torch.int32)
    res = [out]
    if return_logits:
        res.append(torch.stack(all_logits, dim=1))
    if return_stages:
        res.append(stages)
    return res[0] if len(res) == 1 else tuple(res)


# --------------------------------------------------------------------------------------
# VQGAN decoder   (reference: tokenizer/tokenizer_image/vq_model.py)
# --------------------------------------------------------------------------------------
def _gn_swish(x, sd, name, cfg, swish=True):
    y = F.group_norm(x, cfg.gn_groups, sd[name + ".weight"], sd[name + ".bias"], cfg.gn_eps)   # :360-363
    return y * torch.sigmoid(y) if swish else y                                                  # :355-357


def _conv(x, sd, name, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def _resblock(x, sd, name, cin, cout, cfg):
    """reference: vq_model.py:300-315."""
    h = _conv(_gn_swish(x, sd, name + ".norm1", cfg), sd, name + ".conv1", 1)
    h = _conv(_gn_swish(h, sd, name + ".norm2", cfg), sd, name + ".conv2", 1)
    if cin != cout:
        x = _conv(x, sd, name + ".nin_shortcut", 0)
    return x + h


def _attnblock(x, sd, name, cfg):
    """reference: vq_model.py:328-352 — single head over h*w positions, scale C^-0.5."""
    h_ = _gn_swish(x, sd, name + ".norm", cfg, swish=False)
    q, k, v = (_conv(h_, sd, f"{name}.{n}", 0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    v = v.reshape(b, c, hh * ww)
    o = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(o, sd, name + ".proj_out", 0)


def _decoder_layout(cfg):
    """Decoder module order (reference: vq_model.py:129-169 __init__, :174-195 forward)."""
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
    return out


def vq_decode_code(sd, cfg, codes: Tensor, shape) -> Tensor:
    """reference: vq_model.py:53-56 decode_code -> :262-277 get_codebook_entry (L2-normalised
    codebook, channel_first) -> :48-51 post_quant_conv + Decoder.forward :174-195.  fp32."""
    B, C, h, w = shape
    emb = F.normalize(sd["quantize.embedding.weight"].float(), p=2, dim=-1)
    z = emb[codes.reshape(-1).long()].reshape(B, h, w, C).permute(0, 3, 1, 2).contiguous()
    x = _conv(z, sd, "post_quant_conv", 0)
    x = _conv(x, sd, "decoder.conv_in", 1)
    for item in _decoder_layout(cfg):
        if item[0] == "res":
            x = _resblock(x, sd, item[1], item[2], item[3], cfg)
        elif item[0] == "attn":
            x = _attnblock(x, sd, item[1], cfg)
        else:   # Upsample: nearest x2 then conv3x3 (:375-379)
            x = _conv(x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3), sd, item[1] + ".conv", 1)
    x = _gn_swish(x, sd, "decoder.norm_out", cfg)
    return _conv(x, sd, "decoder.conv_out", 1)


# --------------------------------------------------------------------------------------
# VQGAN encoder + quantizer ("next" row §8f-4)   (reference: vq_model.py:41-46 encode, :62-126 Encoder, :216-246 quantizer)
# --------------------------------------------------------------------------------------
def _encoder_layout(cfg):
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
    return out


def vq_encode(sd, cfg, img: Tensor):
    """VQModel.encode -> min_encoding_indices [B, h*w] (int64) and the pre-quantisation z [B, cd, h, w].
    Encoder.forward (:107-126); Downsample = F.pad (0,1,0,1) + conv3x3 stride 2 (:382-396); quant_conv (:39,:44);
    VectorQuantizer.forward (:216-232): l2-normalise z and codebook, d = |z|^2 + |e|^2 - 2 z.e, argmin (first minimum)."""
    x = _conv(img.float(), sd, "encoder.conv_in", 1)
    for item in _encoder_layout(cfg):
        if item[0] == "res":
            x = _resblock(x, sd, item[1], item[2], item[3], cfg)
        elif item[0] == "attn":
            x = _attnblock(x, sd, item[1], cfg)
        else:
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[item[1] + ".conv.weight"], sd[item[1] + ".conv.bias"], stride=2)
    x = _conv(_gn_swish(x, sd, "encoder.norm_out", cfg), sd, "encoder.conv_out", 1)
    z = _conv(x, sd, "quant_conv", 0)
    B, C, h, w = z.shape
    zf = F.normalize(z.permute(0, 2, 3, 1).reshape(-1, C), p=2, dim=-1)
    emb = F.normalize(sd["quantize.embedding.weight"].float(), p=2, dim=-1)
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - 2 * (zf @ emb.t())
    return torch.argmin(d, dim=1).view(B, h * w), z
