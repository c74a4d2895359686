"""T5Embedder with the reference's interface (language/t5.py:14-79, get_text_embeddings :185-201), the encoder running in
libcontrolar_hip.so (car_t5_encode).  Callers that keep working: sample_t2i.py:99-118, sample_t2i_MR.py, demo/model.py.

What stays outside the boundary: the sentencepiece tokenizer and the caption cleaning (`clean_caption`: ftfy / BeautifulSoup /
regex string work on the CPU).  The constructor keeps the reference's keywords (language/t5.py:19-20): `local_cache` / `cache_dir`
resolve the checkpoint directory as `os.path.join(cache_dir, dir_or_name)`; `use_text_preprocessing` defaults to True as in the
reference and then needs the reference's own `clean_caption` — taken from `language.t5` when that module imports (the drop-in lives in
the reference tree; it needs ftfy + bs4), or from a `text_preprocessing` callable; when neither is available `text_preprocessing()` /
`get_text_embeddings()` RAISE instead of silently cleaning captions differently.  `use_text_preprocessing=False` is the reference's `text.lower().strip()` branch."""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional

import torch

from .config import T5Config, flan_t5_xl, tiny_t2i
from .engine import Engine


def t5_config_from_hf(d: dict) -> T5Config:
    """config.json of an HF T5 checkpoint -> T5Config (only the gated-gelu family the reference uses is supported)."""
    ff = d.get("feed_forward_proj", "relu")
    if ff != "gated-gelu":
        raise ValueError(f"feed_forward_proj={ff!r}: only 'gated-gelu' (flan-t5 / t5-v1_1) is supported")
    return T5Config(vocab_size=d["vocab_size"], d_model=d["d_model"], d_kv=d["d_kv"], num_heads=d["num_heads"], d_ff=d["d_ff"],
                    num_layers=d["num_layers"], relative_attention_num_buckets=d.get("relative_attention_num_buckets", 32),
                    relative_attention_max_distance=d.get("relative_attention_max_distance", 128),
                    layer_norm_epsilon=d.get("layer_norm_epsilon", 1e-6))


def load_t5_dir(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of an HF checkpoint directory (safetensors or pytorch_model*.bin shards)."""
    sd: Dict[str, torch.Tensor] = {}
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors") or (f.startswith("pytorch_model") and f.endswith(".bin")))
    if not files:
        raise Exception("please check model weight")
    for f in files:
        fp = os.path.join(path, f)
        if f.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd.update(load_file(fp))
        else:
            sd.update(torch.load(fp, map_location="cpu", weights_only=True))
    return sd


class T5Embedder:
    available_models = ["t5-v1_1-xxl", "t5-v1_1-xl", "flan-t5-xl"]

    def __init__(self, device, dir_or_name: Optional[str] = None, *, local_cache: bool = False, cache_dir: Optional[str] = None,
                 hf_token=None, use_text_preprocessing: bool = True, t5_model_kwargs=None, torch_dtype=None, use_offload_folder=None,
                 model_max_length: int = 120, config: Optional[T5Config] = None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, tokenizer: Optional[Callable] = None,
                 text_preprocessing: Optional[Callable[[str], str]] = None, engine: Optional[Engine] = None):
        """Either `dir_or_name` = a local HF checkpoint directory (config.json + weights [+ tokenizer files]; with
        `local_cache=True` it is looked up under `cache_dir` as the reference does, language/t5.py:33-35; nothing is
        downloaded — the reference's hf_hub_download branch needs a network), or `config` + `state_dict`.
        `hf_token`, `t5_model_kwargs`, `use_offload_folder` configure HF loading in the reference and have nothing to act on here."""
        self.device = torch.device(device)
        self.use_text_preprocessing = use_text_preprocessing
        self.hf_token = hf_token
        self.cache_dir = cache_dir or os.path.expanduser("~/.cache/IF_")
        self.dir_or_name = dir_or_name
        if local_cache and dir_or_name is not None:
            dir_or_name = os.path.join(self.cache_dir, dir_or_name)
        try:
            self._prep, self._prep_error = self._resolve_preprocessing(use_text_preprocessing, text_preprocessing), None
        except RuntimeError as exc:                  # raised by text_preprocessing() / get_text_embeddings(): encode_ids() needs no cleaner
            self._prep, self._prep_error = None, exc
        self.torch_dtype = torch_dtype or torch.bfloat16
        precision = {torch.bfloat16: "bf16", torch.float32: "fp32"}.get(self.torch_dtype)
        if precision is None:
            raise TypeError("torch_dtype must be torch.bfloat16 or torch.float32")
        self.model_max_length = model_max_length
        if dir_or_name is not None and state_dict is None:
            if not os.path.isdir(dir_or_name):
                raise FileNotFoundError(f"{dir_or_name}: not a local checkpoint directory (no network: nothing is downloaded)")
            import json
            with open(os.path.join(dir_or_name, "config.json")) as fh:
                config = t5_config_from_hf(json.load(fh))
            state_dict = load_t5_dir(dir_or_name)
            if tokenizer is None:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(dir_or_name)
        if state_dict is None:
            raise ValueError("T5Embedder needs dir_or_name or (config, state_dict)")
        self.config = config or flan_t5_xl()
        self.tokenizer = tokenizer
        # the encoder lives in its own context unless one is shared (car_t5_configure works on any context)
        self.engine = engine or Engine(tiny_t2i(), precision=precision, device=self.device)
        if self.engine.dtype != self.torch_dtype:
            raise TypeError("shared engine precision differs from torch_dtype")
        self.engine.t5_configure(self.config)
        self.engine.load_t5_state_dict(state_dict, finalize=True)

    @staticmethod
    def _resolve_preprocessing(use_text_preprocessing: bool, user_fn: Optional[Callable[[str], str]]) -> Callable[[str], str]:
        if user_fn is not None:
            return user_fn
        if not use_text_preprocessing:
            return lambda t: t.lower().strip()              # language/t5.py:88
        try:                                                 # the reference's own cleaner, applied twice as at language/t5.py:83-86
            from language.t5 import T5Embedder as _RefEmbedder
        except Exception as exc:
            raise RuntimeError(
                "T5Embedder(use_text_preprocessing=True) needs the reference's clean_caption (language/t5.py:96-182: ftfy + bs4 + regex "
                "string work that stays on the host).  `language.t5` did not import here (" + type(exc).__name__ + ": " + str(exc) + "). "
                "Pass text_preprocessing=<callable>, or use_text_preprocessing=False for the reference's lower().strip() branch.") from exc
        ref = _RefEmbedder.__new__(_RefEmbedder)             # clean_caption only touches the class-level regex
        return lambda t: ref.clean_caption(ref.clean_caption(t))

    def text_preprocessing(self, text: str) -> str:
        if self._prep is None:
            raise self._prep_error
        return self._prep(text)

    @torch.no_grad()
    def encode_ids(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.engine.t5_encode(input_ids, attention_mask)

    @torch.no_grad()
    def get_text_embeddings(self, texts):
        """-> (last_hidden_state [B, model_max_length, d_model], attention_mask [B, model_max_length]) on self.device."""
        if self.tokenizer is None:
            raise RuntimeError("T5Embedder.get_text_embeddings needs a tokenizer (pass tokenizer=... or a checkpoint directory with spiece.model)")
        texts = [self.text_preprocessing(t) for t in texts]
        tk = self.tokenizer(texts, max_length=self.model_max_length, padding="max_length", truncation=True,
                            return_attention_mask=True, add_special_tokens=True, return_tensors="pt")
        mask = tk["attention_mask"].to(self.device)
        embs = self.engine.t5_encode(tk["input_ids"].to(self.device), mask)
        return embs, mask
