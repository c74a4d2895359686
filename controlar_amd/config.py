"""Model/shape configuration for the ControlAR conditional-decoding hot path.

Mirrors the reference's hyper-parameter dataclasses for the components on the path
(reference: autoregressive/models/gpt_t2i.py:31-60 ModelArgs, :556-563 GPT_XL/GPT_B;
tokenizer/tokenizer_image/vq_model.py:12-24 ModelArgs, :422 VQ_16; HF Dinov2Config as
instantiated by autoregressive/models/dinov2_adapter.py:13).  Nothing here computes.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List, Tuple


def find_multiple(n: int, k: int) -> int:
    """reference: autoregressive/models/gpt_t2i.py:26-29"""
    return n if n % k == 0 else n + k - (n % k)


def ffn_hidden_dim(dim: int, multiple_of: int = 256) -> int:
    """reference: autoregressive/models/gpt_t2i.py:204-209 (FeedForward.__init__)"""
    hidden = int(2 * (4 * dim) / 3)
    return find_multiple(hidden, multiple_of)


@dataclass
class GPTConfig:
    dim: int = 1280
    n_layer: int = 36
    n_head: int = 20
    vocab_size: int = 16384
    cls_token_num: int = 120          # text prefix length T (t2i)
    block_size: int = 1024            # image tokens; rope grid = sqrt(block_size)
    caption_dim: int = 2048
    norm_eps: float = 1e-5
    rope_base: float = 10000.0
    multiple_of: int = 256
    model_type: str = "t2i"
    num_classes: int = 1000
    adapter_size: str = "small"
    condition_type: str = "canny"

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_head

    @property
    def ffn_hidden(self) -> int:
        return ffn_hidden_dim(self.dim, self.multiple_of)

    @property
    def grid(self) -> int:
        g = int(self.block_size ** 0.5)
        assert g * g == self.block_size
        return g

    @property
    def layer_internal(self) -> int:
        return self.n_layer // 3


@dataclass
class ViTConfig:
    """Control encoder as the reference instantiates it: DINOv2-S/B (t2i) or ViT-S/16 (c2i)."""
    hidden: int = 384
    layers: int = 12
    heads: int = 6
    mlp_ratio: int = 4
    patch: int = 14
    image_size: int = 518             # native pos-emb grid = image_size // patch = 37
    ln_eps: float = 1e-6
    variant: str = "dinov2"           # "dinov2" (LayerScale, resize to patch-14 grid) | "vit" (HF ViTModel, no LayerScale, no resize)

    @property
    def mlp(self) -> int:
        return self.hidden * self.mlp_ratio

    @property
    def pos_grid(self) -> int:
        return self.image_size // self.patch


@dataclass
class VQConfig:
    codebook_size: int = 16384
    codebook_embed_dim: int = 8
    z_channels: int = 256
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    gn_groups: int = 32
    gn_eps: float = 1e-6


@dataclass
class T5Config:
    """HF T5Config fields the encoder reads (language/t5.py:58-79 loads flan-t5-xl / t5-v1_1-xxl: feed_forward_proj 'gated-gelu')."""
    vocab_size: int = 32128
    d_model: int = 2048
    d_kv: int = 64
    num_heads: int = 32
    d_ff: int = 5120
    num_layers: int = 24
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    model_max_length: int = 120          # T5Embedder(model_max_length=120)


def flan_t5_xl() -> T5Config:
    """google/flan-t5-xl, the encoder every t2i sampler of the reference loads (sample_t2i.py:99-107)."""
    return T5Config()


def tiny_t5() -> T5Config:
    return T5Config(vocab_size=512, d_model=64, d_kv=32, num_heads=2, d_ff=128, num_layers=2)


def small_t5() -> T5Config:
    """t5-small-shaped gated variant (8 heads x 64): mid-size parity case."""
    return T5Config(vocab_size=4096, d_model=512, d_kv=64, num_heads=8, d_ff=1024, num_layers=4)


@dataclass
class PathConfig:
    gpt: GPTConfig = field(default_factory=GPTConfig)
    vit: ViTConfig = field(default_factory=ViTConfig)
    vq: VQConfig = field(default_factory=VQConfig)

    def to_dict(self):
        return asdict(self)


def xl_t2i(block_size: int = 1024, adapter_size: str = "small", condition_type: str = "canny") -> PathConfig:
    """BASELINE configs 2-5: GPT-XL t2i (reference: gpt_t2i.py:556, sample_t2i.py:56-62)."""
    vit = ViTConfig() if adapter_size == "small" else ViTConfig(hidden=768, heads=12)
    return PathConfig(gpt=GPTConfig(block_size=block_size, adapter_size=adapter_size,
                                    condition_type=condition_type), vit=vit, vq=VQConfig())


def b_t2i(block_size: int = 256, adapter_size: str = "small", condition_type: str = "canny") -> PathConfig:
    """GPT-B sized t2i (reference: gpt_t2i.py:562)."""
    vit = ViTConfig() if adapter_size == "small" else ViTConfig(hidden=768, heads=12)
    return PathConfig(gpt=GPTConfig(dim=768, n_layer=12, n_head=12, block_size=block_size,
                                    adapter_size=adapter_size, condition_type=condition_type),
                      vit=vit, vq=VQConfig())


def vit_small16() -> ViTConfig:
    """WinKawaks/vit-small-patch16-224 as gpt.py:319 / vit_adapter.py:11 load it."""
    return ViTConfig(hidden=384, layers=12, heads=6, mlp_ratio=4, patch=16, image_size=224, ln_eps=1e-12, variant="vit")


def b_c2i(block_size: int = 256) -> PathConfig:
    """BASELINE config 1: LlamaGen-B class-conditional + ViT-S/16 control (reference gpt.py:542, sample_c2i.py:49-57)."""
    return PathConfig(gpt=GPTConfig(dim=768, n_layer=12, n_head=12, block_size=block_size, cls_token_num=1, model_type="c2i",
                                    condition_type="canny"), vit=vit_small16(), vq=VQConfig())


def l_c2i(block_size: int = 256) -> PathConfig:
    """LlamaGen-L class-conditional (reference gpt.py:545 GPT_L: 24 layers, 16 heads, dim 1024) + ViT-S/16 control — the size of the released
    ControlAR c2i depth / canny checkpoints next to GPT-B."""
    return PathConfig(gpt=GPTConfig(dim=1024, n_layer=24, n_head=16, block_size=block_size, cls_token_num=1, model_type="c2i",
                                    condition_type="depth"), vit=vit_small16(), vq=VQConfig())


def tiny_c2i(block_size: int = 64, vocab_size: int = 1024, num_classes: int = 10) -> PathConfig:
    return PathConfig(
        gpt=GPTConfig(dim=256, n_layer=6, n_head=4, vocab_size=vocab_size, block_size=block_size, cls_token_num=1,
                      model_type="c2i", num_classes=num_classes, condition_type="canny"),
        vit=ViTConfig(hidden=128, layers=3, heads=2, mlp_ratio=4, patch=16, image_size=224, ln_eps=1e-12, variant="vit"),
        vq=VQConfig(codebook_size=vocab_size, z_channels=64, ch=32),
    )


def tiny_t2i_base(block_size: int = 64, condition_type: str = "hed", vocab_size: int = 1024) -> PathConfig:
    """tiny_t2i with a 'base'-shaped control encoder (3 heads x 64, like DINOv2-base's 12 x 64) and a bicubic condition type."""
    cfg = tiny_t2i(block_size, condition_type, vocab_size)
    cfg.vit = ViTConfig(hidden=192, layers=3, heads=3)
    return cfg


def tiny_t2i(block_size: int = 64, condition_type: str = "canny", vocab_size: int = 1024) -> PathConfig:
    """Small parity-test configuration: same graph, every dimension shrunk so the CPU oracle
    (and the imported reference) finish in seconds.  head_dim stays 64 as in every LlamaGen size."""
    return PathConfig(
        gpt=GPTConfig(dim=256, n_layer=6, n_head=4, vocab_size=vocab_size, block_size=block_size,
                      condition_type=condition_type),
        vit=ViTConfig(hidden=128, layers=3, heads=2),
        vq=VQConfig(codebook_size=vocab_size, z_channels=64, ch=32),
    )
