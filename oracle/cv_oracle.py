"""CPU ORACLE (test infrastructure, not product code) for the two OpenCV calls inside the reference's input builder
(src/models/stage_1/unwrap_utils.py): `cv2.resize(..., INTER_LINEAR)` (:35, :68-70, :131) and
`cv2.remap(img, map, None, cv2.INTER_LINEAR)` (:22).  OpenCV is a third-party dependency that is absent here
(environment.yml pins opencv-python 4.x; no cv2 wheel in this image), so this file restates the PUBLISHED algorithm of
OpenCV 4.x `modules/imgproc/src/resize.cpp` (resizeGeneric_ / HResizeLinear / VResizeLinear) and
`modules/imgproc/src/imgwarp.cpp` (cv::remap float-map conversion + remapBilinear), in the scalar (non-SIMD) form:

resize, INTER_LINEAR, no anti-aliasing:
  * per destination column  fx = (float)((dx + 0.5) * scale_x - 0.5)  with scale_x = (double) src_w / dst_w;
    sx = floor(fx); fx -= sx (float); sx < 0 -> (sx, fx) = (0, 0); sx >= src_w - 1 -> (src_w - 1, 0)  [one clamped tap];
    the two coefficients are the FLOATS  1.f - fx  and  fx  (alpha table type AT = float, also for CV_64F images);
  * rows likewise with fy; horizontal pass first:  t = S[sx] * a0 + S[sx + 1] * a1,  then  D = t0 * b0 + t1 * b1,
    both in the work type WT = double for CV_64F sources (the frames and masks: `astype(float64) / 255`), float for
    CV_32F sources (the flows).
remap, INTER_LINEAR, CV_32FC2 map, BORDER_CONSTANT 0:
  * the sampling position is converted to fixed point with INTER_BITS = 5:  sx = cvRound(mapx * 32)  (round half to
    even), pixel = sx >> 5, fraction = (sx & 31) / 32 — i.e. positions are quantised to 1/32 px;
  * weights from the bilinear table  w = {(1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx}  (floats, exact products of k/32);
  * D = v00*w0 + v01*w1 + v10*w2 + v11*w3  in float, left to right; taps outside the image contribute the border value 0.
Known build-dependent detail NOT restated: OpenCV's AVX2/FMA3 dispatch of the float vertical pass fuses one multiply-add
(`v_muladd`), which can move a CV_32F resize result by one ulp.

Pinning: cv2 cannot be imported here, so this restatement is pinned by hand-computed vectors of the published
algorithm (tests/test_cv_oracle.py); with cv2 installed the same test file compares against it directly
(`pytest.importorskip("cv2")`).  Only tests/ import this module.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def _linear_coeffs(src, dst):
    """resize.cpp: per destination index the first source tap and the two float coefficients."""
    scale = float(src) / float(dst)                           # double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)           # (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                           # cvFloor
    f = (f - s.astype(np.float32)).astype(np.float32)          # fx -= sx   (float)
    lo = s < 0
    s[lo] = 0; f[lo] = 0.0
    hi = s >= src - 1
    s[hi] = src - 1; f[hi] = 0.0
    a0 = (np.float32(1.0) - f).astype(np.float32)
    return s, np.minimum(s + 1, src - 1), a0, f


def cv_resize_linear(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) with the default INTER_LINEAR.  img: (H, W[, C]) float32 or float64."""
    img = np.asarray(img)
    assert img.dtype in (np.float32, np.float64)
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img.copy()                                      # cv::resize copies when the sizes agree
    wt = img.dtype.type                                        # work type: float for CV_32F, double for CV_64F
    sx, sx1, a0, a1 = _linear_coeffs(w, new_w)
    sy, sy1, b0, b1 = _linear_coeffs(h, new_h)
    shp = (1, new_w) + (1,) * (img.ndim - 2)
    a0, a1 = a0.astype(wt).reshape(shp), a1.astype(wt).reshape(shp)
    rows0 = img[sy][:, sx] * a0 + img[sy][:, sx1] * a1         # horizontal pass on the two source rows of each output row
    rows1 = img[sy1][:, sx] * a0 + img[sy1][:, sx1] * a1
    shp = (new_h, 1) + (1,) * (img.ndim - 2)
    out = rows0 * b0.astype(wt).reshape(shp) + rows1 * b1.astype(wt).reshape(shp)
    return out.astype(img.dtype)


def cv_resize_flow(flow, newh, neww):
    """unwrap_utils.py:33-38 with cv2.resize restated (CV_32FC2: float arithmetic)."""
    oldh, oldw = flow.shape[:2]
    out = cv_resize_linear(np.asarray(flow, np.float32), neww, newh)
    out[:, :, 0] *= np.float32(newh / oldh)
    out[:, :, 1] *= np.float32(neww / oldw)
    return out


def cv_remap_linear(img, mapxy):
    """cv2.remap(img, mapxy, None, cv2.INTER_LINEAR): img (H, W, C) float32, mapxy (h, w, 2) float32 (x, y); constant-0 border."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    q = np.rint(np.asarray(mapxy, np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)    # cvRound: half to even
    ix, iy = q[..., 0] >> INTER_BITS, q[..., 1] >> INTER_BITS
    fx = ((q[..., 0] & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE))[..., None]
    fy = ((q[..., 1] & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE))[..., None]
    one = np.float32(1.0)
    w0, w1, w2, w3 = (one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[..., None], v, np.float32(0.0)).astype(np.float32)

    return ((tap(iy, ix) * w0 + tap(iy, ix + 1) * w1) + tap(iy + 1, ix) * w2) + tap(iy + 1, ix + 1) * w3


def cv_compute_consistency(flow12, flow21):
    """unwrap_utils.py:10-23: || flow12 + warp_flow(flow21, flow12) || with cv2.remap restated."""
    flow12 = np.asarray(flow12, np.float32)
    h, w = flow12.shape[:2]
    m = flow12.copy()
    m[:, :, 0] += np.arange(w)                                 # float32 += int64 (one rounding of the exact sum)
    m[:, :, 1] += np.arange(h)[:, np.newaxis]
    diff = flow12 + cv_remap_linear(np.asarray(flow21, np.float32), m)
    return (diff[:, :, 0] ** 2 + diff[:, :, 1] ** 2) ** .5
