"""``Model.process_edge`` / ``Model.process_depth`` with the reference's signatures (demo/model.py:92-105, :192-203), running the
hot path (control encoder -> generate -> decode_code) in libcontrolar_hip.so.

What sits in FRONT of the path in the reference demo is injected, because it is outside this library's scope (SURVEY.md §2:
condition extractors and the Flan-T5 encoder are upstream producers):
  * ``preprocessor(name, image, **kw) -> PIL.Image | np.ndarray`` — the reference's external ``Preprocessor`` (Canny / HED / Lineart /
    depth).  ``'No preprocess'`` (a choice of the reference UI, demo/model.py:123-124) needs none: the image IS the control map;
    ``'Canny'`` falls back to the built-in GPU extractor (``controlar_amd.condition.CannyDetector``) when nothing is injected.
  * ``text_encoder(prompts) -> (caption_embs [B,120,2048], emb_masks [B,120])`` — ``T5Embedder.get_text_embeddings`` (language/t5.py:58-79).
    A prompt may also be given directly as such a pair (precomputed features, as the training pipeline stores them).
The rest — resize to 512x512, ``2*(x/255-0.5)``, left-padding of the caption, ``generate(..., sample_logits=True)``, ``decode_code``,
control map prepended, uint8 PIL images — follows demo/model.py:127-187 line by line.

Deviations, all benign: weights are loaded once (the reference re-reads the safetensors file per request, demo/model.py:132); the
``seed`` argument seeds the library's sampler (the reference accepts and ignores it)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .generate import generate
from .models import Transformer, VQModel

TextFeatures = Tuple[torch.Tensor, torch.Tensor]


def left_pad_caption(caption_embs: torch.Tensor, emb_masks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """demo/model.py:141-152 (= sample_t2i.py:146-160): valid tokens move to the END of the 120 slots, the mask is flipped,
    and the embeddings are multiplied by the new mask."""
    new_masks = torch.flip(emb_masks, dims=[-1])
    rows = []
    for emb, m in zip(caption_embs, emb_masks):
        valid = int(m.sum().item())
        rows.append(torch.cat([emb[valid:], emb[:valid]]))
    new_embs = torch.stack(rows)
    return new_embs * new_masks[:, :, None].to(new_embs.dtype), new_masks


class Model:
    def __init__(self, gpt_edge: Optional[Transformer] = None, gpt_depth: Optional[Transformer] = None, vq_model: Optional[VQModel] = None,
                 text_encoder: Optional[Callable[[Sequence[str]], TextFeatures]] = None,
                 preprocessor: Optional[Callable[..., object]] = None, device: str = "cuda"):
        self.gpt = {"edge": gpt_edge, "depth": gpt_depth}
        self.vq_model = vq_model
        self.text_encoder = text_encoder
        self.preprocessor = preprocessor
        self.device = torch.device(device)

    # ------------------------------------------------------------------ shared tail of both entry points (demo/model.py:127-187)
    def _run(self, kind: str, condition_img, prompt, cfg_scale, temperature, top_k, top_p, seed, control_strength) -> list:
        from PIL import Image
        gpt, vq = self.gpt[kind], self.vq_model
        if gpt is None or vq is None:
            raise RuntimeError(f"Model: no GPT model for '{kind}' / no VQ model was given")
        if isinstance(condition_img, np.ndarray):
            condition_img = Image.fromarray(condition_img)
        condition_img = condition_img.resize((512, 512))
        W, H = condition_img.size
        arr = np.array(condition_img)
        if arr.ndim == 2:
            arr = np.repeat(arr[:, :, None], 3, axis=2)
        cond = torch.from_numpy(arr[:, :, :3].copy()).unsqueeze(0).permute(0, 3, 1, 2).to(self.device)
        cond = 2 * (cond / 255 - 0.5)
        if isinstance(prompt, (tuple, list)) and len(prompt) == 2 and torch.is_tensor(prompt[0]):
            caption_embs, emb_masks = prompt
        else:
            if self.text_encoder is None:
                raise RuntimeError("Model: a text prompt needs the text_encoder callable (T5Embedder.get_text_embeddings, language/t5.py:58-79); "
                                   "alternatively pass (caption_embs, emb_masks) as the prompt")
            caption_embs, emb_masks = self.text_encoder([prompt] * 1)
        c_indices, c_emb_masks = left_pad_caption(caption_embs.to(self.device), emb_masks.to(self.device))
        qzshape = [len(c_indices), 8, H // 16, W // 16]
        index_sample = generate(gpt, c_indices, (H // 16) * (W // 16), c_emb_masks, condition=cond, cfg_scale=cfg_scale, temperature=temperature,
                                top_k=top_k, top_p=top_p, sample_logits=True, control_strength=control_strength, seed=int(seed))
        samples = vq.decode_code(index_sample, qzshape)                     # in [-1, 1]
        samples = torch.cat((cond[0:1].to(samples.dtype), samples), dim=0)
        samples = 255 * (samples * 0.5 + 0.5)
        return [Image.fromarray(s.permute(1, 2, 0).cpu().detach().numpy().clip(0, 255).astype(np.uint8)) for s in samples]

    def _preprocess(self, name: str, image, **kw):
        if name == "No preprocess":
            return image
        if self.preprocessor is None and name == "Canny":
            # built-in: cv2.Canny on the GPU (car_canny, condition/canny.py:6-14).  The reference's external Preprocessor (demo/model.py:15,32 — not part
            # of its repository) resizes the photo to `detect_resolution` BEFORE detecting edges, by the rule of condition/utils.py:28-38
            # (shorter side -> detect_resolution, both sides rounded to multiples of 64, Lanczos when enlarging / area when shrinking); the same
            # rule is applied here with PIL's filters (cv2 is not available: LANCZOS / BOX are its counterparts, not bit-equal to cv2.resize).
            # The caller then resizes the edge map to 512x512 exactly as demo/model.py:127 does.
            from PIL import Image
            from .condition import CannyDetector
            if getattr(self, "_canny", None) is None:
                self._canny = CannyDetector(self.device)
            img = image.convert("RGB")
            res = kw.get("detect_resolution")
            if res:
                W0, H0 = img.size
                k = float(res) / min(H0, W0)
                H1, W1 = int(np.round(H0 * k / 64.0)) * 64, int(np.round(W0 * k / 64.0)) * 64
                if (W1, H1) != (W0, H0) and H1 > 0 and W1 > 0:
                    img = img.resize((W1, H1), Image.LANCZOS if k > 1 else Image.BOX)
            return self._canny(np.array(img), kw.get("low_threshold", 100), kw.get("high_threshold", 200))
        if self.preprocessor is None:
            raise RuntimeError(f"Model: preprocessor '{name}' needs the preprocessor callable (the reference's external Preprocessor, demo/model.py:15,32); "
                               "'No preprocess' takes the image as the control map")
        return self.preprocessor(name, image, **kw)

    @torch.no_grad()
    def process_edge(self, image: np.ndarray, prompt: str, cfg_scale: float, temperature: float, top_k: int, top_p: int, seed: int,
                     low_threshold: int, high_threshold: int, control_strength: float, preprocessor_name: str) -> list:
        """demo/model.py:92-188.  preprocessor_name in {'Canny', 'Hed', 'Lineart', 'No preprocess'}."""
        from PIL import Image
        if isinstance(image, np.ndarray):
            image = Image.fromarray(image)
        if preprocessor_name == "Canny":
            cond = self._preprocess("Canny", image, low_threshold=low_threshold, high_threshold=high_threshold, detect_resolution=512)
        elif preprocessor_name in ("Hed", "Lineart"):
            cond = self._preprocess({"Hed": "HED", "Lineart": "Lineart"}[preprocessor_name], image, image_resolution=512, detect_resolution=512)
        elif preprocessor_name == "No preprocess":
            cond = image
        else:
            raise ValueError(f"unknown preprocessor_name {preprocessor_name!r}")
        return self._run("edge", cond, prompt, cfg_scale, temperature, top_k, top_p, seed, control_strength)

    @torch.no_grad()
    def process_depth(self, image: np.ndarray, prompt: str, cfg_scale: float, temperature: float, top_k: int, top_p: int, seed: int,
                      control_strength: float, preprocessor_name: str) -> list:
        """demo/model.py:192-284.  preprocessor_name in {'depth', 'No preprocess'}."""
        from PIL import Image
        if isinstance(image, np.ndarray):
            image = Image.fromarray(image)
        if preprocessor_name == "depth":
            cond = self._preprocess("Depth", image, image_resolution=512, detect_resolution=512)
        elif preprocessor_name == "No preprocess":
            cond = image
        else:
            raise ValueError(f"unknown preprocessor_name {preprocessor_name!r}")
        return self._run("depth", cond, prompt, cfg_scale, temperature, top_k, top_p, seed, control_strength)
