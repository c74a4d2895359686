"""car_canny on the GPU (pytest -m gpu) against the CPU restatement oracle/canny_oracle.py — integer arithmetic, so the edge maps
must be identical bit for bit; plus the control tensor of sample_t2i.py:125,141 and the CannyDetector call shape (condition/canny.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _photo(seed, H, W):
    """photo-like uint8 RGB: smooth low-frequency colour field + a few hard-edged shapes + sensor noise"""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 3), np.float32)
    for c in range(3):
        for _ in range(4):
            fx, fy, ph = g.uniform(0.002, 0.03), g.uniform(0.002, 0.03), g.uniform(0, 6.28)
            img[:, :, c] += g.uniform(20, 60) * np.sin(fx * xx + fy * yy + ph)
    img += 128
    for _ in range(8):
        cx, cy, r = g.uniform(0, W), g.uniform(0, H), g.uniform(5, min(H, W) / 4)
        m = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        img[m] = img[m] * 0.3 + g.uniform(0, 255, size=3)
    x0, y0 = int(g.uniform(0, W / 2)), int(g.uniform(0, H / 2))
    img[y0:y0 + H // 3, x0:x0 + W // 3] += 70
    img += g.normal(0, 3, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("H,W,B", [(512, 512, 3), (97, 131, 2), (33, 40, 1), (768, 512, 1)])
def test_canny_bit_exact_vs_oracle(H, W, B):
    from controlar_amd import config as C
    from controlar_amd.engine import Engine
    from oracle import canny_oracle as K
    eng = Engine(C.tiny_t2i(), "bf16")
    imgs = np.stack([_photo(10 + i, H, W) for i in range(B)])
    for low, high in ((100, 200), (30.7, 90.2), (250, 120)):
        edges, ctrl = eng.canny(torch.from_numpy(imgs), low, high, want_control=True)
        e = edges.cpu().numpy()
        for i in range(B):
            want = K.canny(imgs[i], low, high)
            assert np.array_equal(e[i], want), (H, W, i, low, high, int((e[i] != want).sum()))
        assert 0.005 < (e > 0).mean() < 0.5                       # a sane edge density on photo-like input
        c = ctrl.float().cpu().numpy()
        assert np.array_equal(c[:, 0], 2 * (e.astype(np.float32) / 255 - 0.5)) and np.array_equal(c[:, 0], c[:, 1]) and np.array_equal(c[:, 1], c[:, 2])
    eng.close()


def test_canny_detector_call_shape_and_feeds_the_path():
    """CannyDetector()(array (H,W,3)) -> array (H,W), as condition/canny.py:7-14; its control tensor drives encode_control."""
    from controlar_amd.condition import CannyDetector
    from oracle import canny_oracle as K
    from tests.cases import load_case
    from controlar_amd.engine import Engine
    img = _photo(3, 128, 128)
    det = CannyDetector()
    out = det(img)
    assert isinstance(out, np.ndarray) and out.shape == (128, 128) and out.dtype == np.uint8 and set(np.unique(out)) <= {0, 255}
    assert np.array_equal(out, K.canny(img)) and np.array_equal(det(torch.from_numpy(img), 50, 150), K.canny(img, 50, 150))
    cs = load_case("tiny_canny_cfg1")
    eng = Engine(cs["cfg"], "fp32"); eng.load_state_dict(cs["gsd"]); eng.finalize()
    _, ctrl = eng.canny(torch.from_numpy(np.stack([img, img])), want_control=True)
    a = eng.encode_control(ctrl, want_output=True)
    assert tuple(a.shape) == (2, 64, cs["cfg"].gpt.dim) and bool(torch.isfinite(a).all())
    eng.close()
