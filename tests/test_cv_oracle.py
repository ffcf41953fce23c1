"""oracle/cv_oracle.py against hand-computed vectors of OpenCV's published INTER_LINEAR resize / remap algorithms
(cv2 is absent here; when it is installed the last test compares with it directly)."""
import numpy as np
import pytest

from oracle import cv_oracle as C

f32 = np.float32


def test_resize_coefficients_are_float32_of_the_double_position():
    """resize.cpp: fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx); fx -= sx; edge taps clamp with weight 1."""
    s, s1, a0, a1 = C._linear_coeffs(4, 3)                       # scale 4/3
    assert list(s) == [0, 1, 2] and list(s1) == [1, 2, 3]
    exp = [f32(0.5 * (4.0 / 3.0) - 0.5), f32(f32(1.5 * (4.0 / 3.0) - 0.5) - f32(1.0)), f32(f32(2.5 * (4.0 / 3.0) - 0.5) - f32(2.0))]
    assert [float(v) for v in a1] == [float(v) for v in exp]
    assert [float(v) for v in a0] == [float(f32(1.0) - v) for v in exp]
    # upsampling 2 -> 5: first and last positions fall outside the source and clamp
    s, s1, a0, a1 = C._linear_coeffs(2, 5)
    # dx = 0: 0.5*0.4-0.5 = -0.3 -> sx = -1 -> (0, 0);  dx = 3: 0.9 -> sx 0;  dx = 4: 1.3 -> sx = 1 = src-1 -> (1, 0)
    assert list(s) == [0, 0, 0, 0, 1] and float(a1[0]) == 0.0 and float(a1[4]) == 0.0 and float(a0[4]) == 1.0
    assert float(a1[1]) == float(f32(1.5 * 0.4 - 0.5)) and float(a1[2]) == float(f32(2.5 * 0.4 - 0.5)) and float(a1[3]) == float(f32(3.5 * 0.4 - 0.5))


def test_resize_float32_source_uses_float_arithmetic():
    row = np.array([[0.1, 10.3, 20.7, 30.9]], f32)
    out = C.cv_resize_linear(row, 3, 1)
    assert out.dtype == np.float32 and out.shape == (1, 3)
    for dx in range(3):
        fx = f32((dx + 0.5) * (4.0 / 3.0) - 0.5); sx = int(np.floor(fx)); fx = f32(fx - f32(sx))
        exp = f32(f32(row[0, sx] * f32(f32(1.0) - fx)) + f32(row[0, sx + 1] * fx))          # t = S[sx]*a0 + S[sx+1]*a1 in float
        assert float(out[0, dx]) == float(exp)                  # single source row: b0 = 1, b1 = 0 leaves t unchanged


def test_resize_float64_source_keeps_float32_coefficients():
    """CV_64F images (frames / 255): double work type, but the alpha / beta tables are floats (AT = float)."""
    img = (np.arange(12, dtype=np.float64).reshape(3, 4) * 7 % 11) / 255.0
    out = C.cv_resize_linear(img, 3, 2)
    assert out.dtype == np.float64
    fy = f32(0.5 * 1.5 - 0.5); b1 = float(fy); b0 = float(f32(1.0) - fy)                       # dy = 0: rows 0, 1
    fx = f32(1.5 * (4.0 / 3.0) - 0.5); fx = f32(fx - f32(1.0)); a1 = float(fx); a0 = float(f32(1.0) - fx)   # dx = 1: cols 1, 2
    t0 = img[0, 1] * a0 + img[0, 2] * a1
    t1 = img[1, 1] * a0 + img[1, 2] * a1
    assert out[0, 1] == t0 * b0 + t1 * b1
    exact = (img[0, 1] * (1 - 0.25) + img[0, 2] * 0.25)         # (same here: 0.25 and 0.5 are exact) — a case where they differ:
    s, s1, a0v, a1v = C._linear_coeffs(7, 3)
    assert float(a1v[0]) != (0.5 * (7.0 / 3.0) - 0.5)           # float32(2/3) != 2/3: double-weight restatements differ in the last bits


def test_resize_identity_copies():
    img = np.random.default_rng(0).random((5, 6, 3))
    assert np.array_equal(C.cv_resize_linear(img, 6, 5), img)


def test_remap_quantises_positions_to_one_32nd_pixel():
    img = np.array([[[1.0], [3.0], [7.0]], [[2.0], [5.0], [11.0]], [[4.0], [9.0], [13.0]]], f32)
    def at(x, y):
        return float(C.cv_remap_linear(img, np.array([[[x, y]]], f32))[0, 0, 0])
    assert at(0.49, 0.0) == 0.5 * 1.0 + 0.5 * 3.0               # 0.49*32 = 15.68 -> 16 -> frac 0.5
    assert at(0.515625, 0.0) == 0.5 * 1.0 + 0.5 * 3.0           # 16.5 -> 16 (half to even)
    assert at(0.546875, 0.0) == (1 - 0.5625) * 1.0 + 0.5625 * 3.0   # 17.5 -> 18
    assert at(1.0, 1.0) == 5.0
    assert at(0.25, 0.5) == 0.5 * (0.75 * 1.0 + 0.25 * 3.0) + 0.5 * (0.75 * 2.0 + 0.25 * 5.0)
    # constant-0 border, tap by tap: x = -0.25 -> fixed point -8 -> pixel -1, fraction 24/32
    assert at(-0.25, 0.0) == 0.75 * 1.0
    assert at(2.5, 2.0) == 0.5 * 13.0                           # right neighbour outside
    assert at(-1.5, 0.0) == 0.0 and at(0.0, 3.0) == 0.0         # all four taps outside


def test_consistency_mask_rule():
    """unwrap_utils.py:10-23,151-159: exact opposite flows are consistent in the interior, inconsistent where the match leaves the frame."""
    h, w = 6, 8
    f12 = np.zeros((h, w, 2), f32); f12[..., 0] = 1.5; f12[..., 1] = 0.5
    c = C.cv_compute_consistency(f12, -f12)
    assert c.shape == (h, w) and np.all(c[:h - 1, :w - 2] < 1e-6)
    assert np.all(c[:, w - 1] > 1.0)                             # x + 1.5 leaves the frame: the warped flow is the zero border


def test_against_cv2_when_available():
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(3)
    img = rng.random((37, 53, 3))
    assert np.array_equal(cv2.resize(img, (20, 11)), C.cv_resize_linear(img, 20, 11))
    fl = rng.standard_normal((37, 53, 2)).astype(f32) * 3
    m = fl.copy(); m[:, :, 0] += np.arange(53); m[:, :, 1] += np.arange(37)[:, None]
    assert np.array_equal(cv2.remap(fl[::-1].copy(), m, None, cv2.INTER_LINEAR), C.cv_remap_linear(fl[::-1].copy(), m))
    r = cv2.resize(fl, (20, 11), interpolation=cv2.INTER_LINEAR)
    assert np.abs(r - C.cv_resize_linear(fl, 20, 11)).max() <= 4e-7 * np.abs(r).max()     # FMA-dispatch builds may move one ulp
