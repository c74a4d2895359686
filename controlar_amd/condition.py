"""Condition extractors of the path's front end.  ``CannyDetector`` keeps the call shape of the reference's
``condition/canny.py:6-14`` (array or tensor (H, W, 3) in, array (H, W) out) and runs ``car_canny`` on the GPU."""
from __future__ import annotations

import numpy as np
import torch

from .config import tiny_t2i
from .engine import Engine


class CannyDetector:
    def __init__(self, device=None):
        self._eng = Engine(tiny_t2i(), "bf16", device=device)        # the kernel needs no weights: any context serves

    def __call__(self, img, low_threshold=100, high_threshold=200):
        """input: array or tensor (H,W,3)   output: array (H,W)   (condition/canny.py:7-14)"""
        if torch.is_tensor(img):
            img = img.cpu().detach().numpy().astype(np.uint8)
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.uint8)))
        return self._eng.canny(x, low_threshold, high_threshold).cpu().numpy()
