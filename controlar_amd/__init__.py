"""controlar_amd — MI355X-native drop-in for ControlAR's conditional-decoding hot path
(DINOv2 control encoder -> LlamaGen AR decode with per-token control fusion -> VQGAN decoder)."""
from .config import PathConfig, GPTConfig, ViTConfig, VQConfig, xl_t2i, b_t2i, tiny_t2i  # noqa: F401
