"""CPU restatement of ``cv2.Canny(img, low, high)`` as the reference calls it (condition/canny.py:6-14: 8-bit H x W x 3 input,
default apertureSize = 3, L2gradient = False) — TEST INFRASTRUCTURE ONLY (checker of ``car_canny``; never imported by the product).

PARITY UNPINNED.  The arithmetic lives in a third-party dependency that is absent from /root/reference and from this image:
``opencv-python==4.9.0.80`` (requirements.txt:2), ``cv::Canny`` in ``modules/imgproc/src/canny.cpp``.  No (photo, edge map) pair
exists in the reference tree (condition/example/c2i/canny/*.png are edge maps without their source photos), so this restatement
of the published algorithm cannot be checked against OpenCV here; it is anchored on the call site above and on known-answer cases
derived by hand (tests/test_canny_cpu.py).  The algorithm, as implemented by OpenCV 4.x for this call:

 1. dx, dy = 3x3 Sobel of every channel, 16-bit signed, BORDER_REPLICATE.
 2. per pixel the channel with the largest L1 magnitude |dx| + |dy| is kept (first channel wins ties).
 3. non-maximum suppression on mag = |dx| + |dy| (zero outside the image) for pixels with mag > low, direction quantised with
    the fixed-point tangents TG22 = round(tan(22.5 deg) * 2^15) and tan(67.5 deg) = TG22 + 2:
      |dy| * 2^15 <  |dx| * TG22              : horizontal gradient, keep if m >  left  and m >= right
      |dy| * 2^15 >  |dx| * TG22 + |dx| * 2^16 : vertical gradient,   keep if m >  above and m >= below
      otherwise (diagonal), s = -1 if dx, dy have opposite signs else +1: keep if m > prev_row[x - s] and m > next_row[x + s]
      (the two neighbours ACROSS the edge: up-right / down-left when the gradient points right-and-up, y growing downwards)
    kept pixels with m > high are strong seeds, the other kept pixels are weak candidates.
 4. hysteresis: weak candidates 8-connected (through candidates) to a strong seed become edges.  5. output 255 on edges, else 0.
"""
from __future__ import annotations

import numpy as np

TG22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)        # 13573


def sobel3(ch: np.ndarray):
    """3x3 Sobel derivatives of one uint8 channel, int32, BORDER_REPLICATE."""
    p = np.pad(ch.astype(np.int32), 1, mode="edge")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return dx, dy


def gradient(img: np.ndarray):
    """(dx, dy, mag) after the per-pixel channel selection; img uint8 [H, W] or [H, W, C]."""
    if img.ndim == 2:
        img = img[:, :, None]
    dxs, dys = zip(*[sobel3(img[:, :, c]) for c in range(img.shape[2])])
    dx, dy = dxs[0].copy(), dys[0].copy()
    mag = np.abs(dx) + np.abs(dy)
    for c in range(1, img.shape[2]):
        m = np.abs(dxs[c]) + np.abs(dys[c])
        take = m > mag                               # strict: the first channel wins ties
        dx[take], dy[take], mag[take] = dxs[c][take], dys[c][take], m[take]
    return dx, dy, mag


def nms_map(dx, dy, mag, low: int, high: int) -> np.ndarray:
    """0 = weak candidate, 1 = not an edge, 2 = strong seed."""
    H, W = mag.shape
    mp = np.pad(mag, 1, mode="constant")             # magnitudes outside the image are zero
    c = mp[1:-1, 1:-1]
    x = np.abs(dx).astype(np.int64); y = np.abs(dy).astype(np.int64) << 15
    tg22x = x * TG22
    tg67x = tg22x + (x << 16)
    horiz = y < tg22x
    vert = (~horiz) & (y > tg67x)
    diag = ~(horiz | vert)
    s_neg = (dx ^ dy) < 0                             # opposite signs -> s = -1
    yy, xx = np.mgrid[0:H, 0:W]
    left, right = mp[1:-1, :-2], mp[1:-1, 2:]
    above, below = mp[:-2, 1:-1], mp[2:, 1:-1]
    s = np.where(s_neg, -1, 1)
    prev_d = mp[yy, xx + 1 - s]                       # prev_row[x - s]  (row index y-1 -> padded row y)
    next_d = mp[yy + 2, xx + 1 + s]                   # next_row[x + s]
    keep = (horiz & (c > left) & (c >= right)) | (vert & (c > above) & (c >= below)) | (diag & (c > prev_d) & (c > next_d))
    keep &= c > low
    out = np.ones((H, W), dtype=np.uint8)
    out[keep & (c > high)] = 2
    out[keep & ~(c > high)] = 0
    return out


def hysteresis(m: np.ndarray) -> np.ndarray:
    from scipy import ndimage
    lab, n = ndimage.label(m != 1, structure=np.ones((3, 3), dtype=np.int8))
    if n == 0:
        return np.zeros_like(m, dtype=bool)
    has_seed = np.zeros(n + 1, dtype=bool)
    has_seed[np.unique(lab[m == 2])] = True
    has_seed[0] = False
    return has_seed[lab]


def canny(img: np.ndarray, low_threshold: float = 100, high_threshold: float = 200) -> np.ndarray:
    """uint8 [H, W] edge map (255 / 0) of a uint8 image [H, W] or [H, W, 3]."""
    low, high = float(low_threshold), float(high_threshold)
    if low > high:
        low, high = high, low
    dx, dy, mag = gradient(np.asarray(img, dtype=np.uint8))
    m = nms_map(dx, dy, mag, int(np.floor(low)), int(np.floor(high)))
    return (hysteresis(m) * 255).astype(np.uint8)
