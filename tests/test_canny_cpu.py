"""oracle/canny_oracle.py (restatement of cv2.Canny as condition/canny.py:6-14 calls it; parity UNPINNED: opencv-python is absent and
the reference tree holds no photo/edge pair) against known-answer cases worked out by hand from the algorithm's definition."""
import numpy as np

from oracle import canny_oracle as K


def test_vertical_step_edge_is_one_pixel_wide_and_left_of_the_tie():
    img = np.zeros((16, 16, 3), np.uint8); img[:, 8:] = 255
    e = K.canny(img)
    # Sobel gives |dx| = 1020 in columns 7 and 8; horizontal NMS keeps m > left and m >= right: column 7 only
    assert (e[:, 7] == 255).all() and e.sum() == 255 * 16


def test_flat_image_and_subthreshold_ramp_have_no_edges():
    assert K.canny(np.full((9, 9, 3), 77, np.uint8)).sum() == 0
    ramp = np.tile(np.arange(32, dtype=np.uint8)[None, :, None] * 4, (8, 1, 3))      # dx = 4*2*4 = 32 < 100
    assert K.canny(ramp).sum() == 0


def test_channel_of_maximum_gradient_is_used_and_first_channel_wins_ties():
    img = np.zeros((8, 12, 3), np.uint8); img[:, 6:, 2] = 200          # edge only in channel 2
    assert (K.canny(img)[:, 5] == 255).all()
    a = np.zeros((8, 12, 3), np.uint8); a[:, 6:, 0] = 100; a[:, 6:, 1] = 100
    dx, dy, mag = K.gradient(a)
    assert mag[4, 5] == 400 and dx[4, 5] == 400


def test_hysteresis_keeps_weak_pixels_only_when_connected_to_a_strong_seed():
    m = np.ones((5, 9), np.uint8)
    m[2, 1:5] = 0; m[2, 4] = 2          # weak run touching a seed
    m[0, 7] = 0                         # isolated weak pixel
    keep = K.hysteresis(m)
    assert keep[2, 1:5].all() and not keep[0, 7] and keep.sum() == 4


def test_thresholds_are_floored_and_swapped_like_opencv():
    img = np.zeros((8, 8, 3), np.uint8); img[:, 4:] = 30               # |dx| = 120
    assert K.canny(img, 100, 119.9).sum() > 0 and K.canny(img, 100, 120).sum() == 0     # strong needs m > floor(high)
    assert np.array_equal(K.canny(img, 119.9, 100), K.canny(img, 100, 119.9))


def test_diagonal_edge_direction_test():
    img = np.zeros((24, 24, 3), np.uint8)
    for y in range(24):
        img[y, y:] = 255                                               # 45-degree edge
    e = K.canny(img)
    ys, xs = np.nonzero(e[4:20])
    assert len(ys) > 0 and np.all(np.abs(xs - (ys + 4)) <= 1)           # a thin line along the diagonal
