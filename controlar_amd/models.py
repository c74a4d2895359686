"""Drop-in model objects mirroring the reference factories
(autoregressive/models/gpt_t2i.py:566-569 ``GPT_models``; tokenizer/tokenizer_image/vq_model.py:422-424
``VQ_models``).  They keep the attributes ``generate()`` and the sampler scripts touch
(generate.py:137-182; sample_t2i.py:43-83,163-176) but hold no torch parameters: weights live,
packed, inside the HIP context (controlar_amd/engine.py)."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .config import GPTConfig, PathConfig, ViTConfig, VQConfig
from .engine import Engine

_SIZES = {  # reference: gpt_t2i.py:541-563
    "GPT-B": dict(n_layer=12, n_head=12, dim=768), "GPT-L": dict(n_layer=24, n_head=16, dim=1024),
    "GPT-XL": dict(n_layer=36, n_head=20, dim=1280), "GPT-XXL": dict(n_layer=48, n_head=24, dim=1536),
    "GPT-1B": dict(n_layer=22, n_head=32, dim=2048), "GPT-3B": dict(n_layer=24, n_head=32, dim=3200),
}


class _PendingControl:
    """What ``model.adapter(x)`` returns: the control image waiting for ``model.adapter_mlp``."""

    def __init__(self, img):
        self.img = img


class Transformer:
    """Stands in for gpt_t2i.Transformer on the inference path (t2i)."""

    def __init__(self, gpt: GPTConfig, vit: Optional[ViTConfig] = None):
        if gpt.model_type not in ("t2i", "c2i"):
            raise Exception("please check model type")            # generate.py:173 wording
        if vit is None:
            if gpt.model_type == "c2i":                            # gpt.py:319: ViT_Adapter() = HF ViT-S/16
                from .config import vit_small16
                vit = vit_small16()
            else:
                vit = ViTConfig() if gpt.adapter_size == "small" else ViTConfig(hidden=768, heads=12)
        self.cfg = PathConfig(gpt=gpt, vit=vit, vq=VQConfig())
        self.config = gpt
        self.model_type, self.num_classes = gpt.model_type, gpt.num_classes
        self.vocab_size, self.block_size, self.cls_token_num, self.n_layer = gpt.vocab_size, gpt.block_size, gpt.cls_token_num, gpt.n_layer
        self._dtype = torch.bfloat16
        self._device = None
        self._sd: Dict[str, torch.Tensor] = {}
        self._engine: Optional[Engine] = None
        self._ckpt = None
        self.tok_embeddings = SimpleNamespace(weight=SimpleNamespace(dtype=self._dtype))
        self.cls_embedding = SimpleNamespace(uncond_embedding=None)
        self.max_batch_size = self.max_seq_length = -1
        self.training = False

    # ---- torch.nn.Module look-alikes used by the sampler scripts
    def load_state_dict(self, sd, strict: bool = True):
        """torch.nn.Module.load_state_dict semantics (the samplers pass strict=False, sample_t2i.py:69,81): returns the real
        missing / unexpected key lists against the parameters this path reads; strict=True raises on either."""
        from .checkpoint import IGNORED_GPT, expected_gpt_keys, key_report
        missing, unexpected = key_report(expected_gpt_keys(self.cfg), list(sd.keys()), IGNORED_GPT)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Transformer: Missing key(s): {missing[:8]}{'...' if len(missing) > 8 else ''} "
                               f"Unexpected key(s): {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        self._sd.update({k: v for k, v in sd.items() if torch.is_tensor(v)})
        if "cls_embedding.uncond_embedding" in sd:
            self.cls_embedding.uncond_embedding = sd["cls_embedding.uncond_embedding"]
        self._engine = None
        self._ckpt = None
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def load_checkpoint(self, path: str, cache_dir: Optional[str] = None, use_cache: bool = True):
        """sample_t2i.py:64-83 in one call: .safetensors or .pt with model|module|state_dict, strict=False, and the packed-image
        cache of controlar_amd/checkpoint.py (the second start from the same file restores the HIP weight images by plain copies)."""
        from .checkpoint import load_checkpoint
        self._ckpt = (path, cache_dir, use_cache)
        sd = load_checkpoint(path)
        res = self.load_state_dict(sd, strict=False)
        self._ckpt = (path, cache_dir, use_cache)
        return res

    def to(self, *args, **kw):
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.dtype):
                self._dtype = a
            elif isinstance(a, (str, torch.device)):
                self._device = torch.device(a)
        self.tok_embeddings.weight.dtype = self._dtype
        self._engine = None
        return self

    def eval(self):
        return self

    def setup_caches(self, max_batch_size, max_seq_length, dtype):
        """gpt_t2i.py:391-405 re-allocates KV cache + bool mask every call; here the library owns them."""
        self.max_batch_size, self.max_seq_length = max_batch_size, max_seq_length

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            prec = "bf16" if self._dtype == torch.bfloat16 else "fp32"
            # library-side opt-in modes that the reference does not have (tolerance-graded, never the parity path): set e.g.
            # `gpt.engine_options = {"kv_fp8": True}` (e4m3 KV cache) or `{"weights_fp8": True}` before the first generate()
            self._engine = Engine(self.cfg, prec, device=self._device, **getattr(self, "engine_options", {}))
            ck = getattr(self, "_ckpt", None)
            if ck is not None:
                from .checkpoint import load_engine_from_checkpoints
                self.cache_info = load_engine_from_checkpoints(self._engine, gpt_path=ck[0], cache_dir=ck[1], use_cache=ck[2])
            else:
                self._engine.load_state_dict(self._sd, finalize=True)
        return self._engine

    # generate.py:136-138 calls these two in sequence
    def adapter(self, x):
        return _PendingControl(x)

    def adapter_mlp(self, pending):
        return self.engine.encode_control(pending.img, want_output=True)


def _gpt_factory(name):
    def make(**kw):
        kw = dict(kw)
        kw.pop("condition_token_num", None); kw.pop("image_size", None)      # gpt.py ModelArgs extras (sample_c2i.py:55-56)
        known = {f for f in GPTConfig.__dataclass_fields__}
        extra = {k: kw.pop(k) for k in list(kw) if k not in known}      # training-only ModelArgs (dropouts, ...) are accepted and ignored
        del extra
        return Transformer(GPTConfig(**_SIZES[name], **kw))
    return make


GPT_models = {k: _gpt_factory(k) for k in _SIZES}


class VQModel:
    """Stands in for vq_model.VQModel on the decode side (decode_code only; the encoder is training-time)."""

    def __init__(self, vq: VQConfig):
        self.vq = vq
        self.config = vq
        self._dtype = torch.float32            # the reference never casts the VQ model (sample_t2i.py:43-47)
        self._device = None
        self._sd: Dict[str, torch.Tensor] = {}
        self._engine: Optional[Engine] = None

    def load_state_dict(self, sd, strict: bool = True):
        """The reference loads the tokenizer strictly (sample_t2i.py:49).  Required here: the decode side; the encode side
        (encoder.*, quant_conv.*) is optional as a group (it enables encode_indices) and complete if present."""
        from .checkpoint import expected_vq_keys, key_report
        exp = expected_vq_keys(self.vq, "decoder")
        if any(k.startswith(("encoder.", "quant_conv.")) for k in sd):
            exp = exp + expected_vq_keys(self.vq, "encoder")
        missing, unexpected = key_report(exp, list(sd.keys()), ("quantize.codebook_used",))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for VQModel: Missing key(s): {missing[:8]} Unexpected key(s): {unexpected[:8]}")
        self._sd.update({k: v for k, v in sd.items() if torch.is_tensor(v)})
        self._engine = None
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def to(self, *args, **kw):
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.dtype):
                self._dtype = a
            elif isinstance(a, (str, torch.device)):
                self._device = torch.device(a)
        self._engine = None
        return self

    def eval(self):
        return self

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            from .config import tiny_t2i
            cfg = tiny_t2i()                   # GPT/ViT halves of the context stay empty
            cfg.vq = self.vq
            self._engine = Engine(cfg, "bf16" if self._dtype == torch.bfloat16 else "fp32", device=self._device)
            self._engine.load_state_dict(self._sd, finalize=True)
        return self._engine

    def decode_code(self, code_b, shape=None, channel_first=True):
        """vq_model.py:53-56: code_b [B, h*w] (or flat), shape [B, C, h, w] -> fp32 [B,3,16h,16w].
        The other two argument forms fail in the reference as well, and fail the same way here:
          * shape=None      -> get_codebook_entry returns a 2-D [N, C] tensor (vq_model.py:262-277) and post_quant_conv rejects it;
          * channel_first=False, shape=(B, h, w, C) -> the [B, h, w, C] tensor reaches the NCHW convolution with h "channels": a channel-count
            error unless h happens to equal the codebook dimension (then the reference decodes a transposed latent, which no caller on the path does)."""
        C = self.vq.codebook_embed_dim
        if shape is None:
            dims = [int(d) for d in code_b.shape] + [C]
            if len(dims) == 2:
                raise RuntimeError(f"Expected 3D (unbatched) or 4D (batched) input to conv2d, but got input of size: {dims} "
                                   "(decode_code(shape=None) hands the raw codebook gather to post_quant_conv, vq_model.py:48-56,262-277)")
            if len(dims) == 3 and dims[0] != C:      # [B, N, C] is taken for ONE unbatched image with B channels
                raise RuntimeError(f"Given groups=1, weight of size [{self.vq.z_channels}, {C}, 1, 1], expected input[1, {dims[0]}, {dims[1]}, {dims[2]}] to have {C} channels, "
                                   f"but got {dims[0]} channels instead (decode_code(shape=None), vq_model.py:48-56,262-277)")
            raise NotImplementedError("decode_code(shape=None) only 'works' in the reference when the batch size equals the codebook dimension (it then decodes "
                                      "the batch as one unbatched latent); not on the path")
        if not channel_first:
            B, h, w, c_last = shape
            if h != C:
                raise RuntimeError(f"Given groups=1, weight of size [{self.vq.z_channels}, {C}, 1, 1], expected input[{B}, {h}, {w}, {c_last}] to have {C} channels, "
                                   f"but got {h} channels instead (decode_code(channel_first=False) feeds a [B,h,w,C] tensor to an NCHW convolution, vq_model.py:271-276)")
            raise NotImplementedError("decode_code(channel_first=False) with h == codebook_embed_dim decodes a transposed latent in the reference; not on the path")
        B, _, h, w = shape
        return self.engine.vq_decode(code_b.reshape(B, h * w), h, w)

    def encode_indices(self, x):
        """min_encoding_indices of VQModel.encode(x) (vq_model.py:41-46 -> info[2]) as int32 [B, h*w]."""
        return self.engine.vq_encode(x)



def VQ_16(**kw):
    return VQModel(VQConfig(ch_mult=(1, 1, 2, 2, 4), **kw))


def VQ_8(**kw):
    return VQModel(VQConfig(ch_mult=(1, 2, 2, 4), **kw))


VQ_models = {"VQ-16": VQ_16, "VQ-8": VQ_8}
