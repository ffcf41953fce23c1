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
(`pytest.importorskip("cv2")`) — PARITY UNPINNED against the real library until then.  Only tests/ import this module.

Written as SCALAR loops in the order of OpenCV's own code (coefficient tables, a horizontal pass into row buffers, a vertical
pass; a per-pixel fixed-point remap) with numpy scalar types doing the roundings — deliberately NOT the vectorised expressions of
the product's host loader (all-in-one-deflicker_amd/stage1.py), so that the bit-equality the tests assert between the two
(and the device kernels) compares two independent implementations.  Slow: test sizes only.
"""
import math

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
F32 = np.float32


def _linear_coeffs(src, dst):
    """resize.cpp (resizeGeneric_ set-up loop): per destination index the first source tap, the second (clamped) tap and the
    two float coefficients.  Returns (ofs, ofs1, a0, a1) as arrays."""
    scale = float(src) / float(dst)                            # double scale_x = (double)ssize.width / dsize.width
    ofs, ofs1, a0, a1 = [], [], [], []
    for d in range(dst):
        f = F32((d + 0.5) * scale - 0.5)                       # fx = (float)((dx + 0.5) * scale_x - 0.5)
        s = int(math.floor(float(f)))                          # sx = cvFloor(fx)
        f = F32(f - F32(s))                                    # fx -= sx
        if s < 0:
            s, f = 0, F32(0.0)
        if s >= src - 1:
            s, f = src - 1, F32(0.0)
        ofs.append(s); ofs1.append(min(s + 1, src - 1)); a0.append(F32(F32(1.0) - f)); a1.append(f)
    return np.array(ofs, np.int64), np.array(ofs1, np.int64), np.array(a0, np.float32), np.array(a1, np.float32)


def cv_resize_linear(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) with the default INTER_LINEAR.  img: (H, W[, C]) float32 or float64."""
    img = np.asarray(img)
    assert img.dtype in (np.float32, np.float64)
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img.copy()                                      # cv::resize copies when the sizes agree
    wt = img.dtype.type                                        # work type WT: float for CV_32F, double for CV_64F
    src = img.reshape(h, w, -1)
    cn = src.shape[2]
    xofs, xofs1, alpha0, alpha1 = _linear_coeffs(w, new_w)
    yofs, yofs1, beta0, beta1 = _linear_coeffs(h, new_h)
    out = np.empty((new_h, new_w, cn), img.dtype)
    rows = {}                                                  # HResizeLinear results of the source rows in use (the ring of two row buffers)

    def hresize(sy):
        if sy not in rows:
            buf = np.empty((new_w, cn), img.dtype)
            for dx in range(new_w):
                a0, a1 = wt(alpha0[dx]), wt(alpha1[dx])
                for c in range(cn):
                    buf[dx, c] = wt(src[sy, xofs[dx], c]) * a0 + wt(src[sy, xofs1[dx], c]) * a1
            rows[sy] = buf
        return rows[sy]

    for dy in range(new_h):
        r0, r1 = hresize(int(yofs[dy])), hresize(int(yofs1[dy]))
        b0, b1 = wt(beta0[dy]), wt(beta1[dy])
        for dx in range(new_w):
            for c in range(cn):
                out[dy, dx, c] = wt(r0[dx, c]) * b0 + wt(r1[dx, c]) * b1          # VResizeLinear
        for k in [k for k in rows if k < int(yofs[dy])]:
            del rows[k]
    return out.reshape((new_h, new_w) + img.shape[2:])


def cv_resize_flow(flow, newh, neww):
    """unwrap_utils.py:33-38 with cv2.resize restated (CV_32FC2: float arithmetic)."""
    oldh, oldw = flow.shape[:2]
    out = cv_resize_linear(np.asarray(flow, np.float32), neww, newh)
    su, sv = F32(newh / oldh), F32(neww / oldw)
    for y in range(newh):
        for x in range(neww):
            out[y, x, 0] = F32(out[y, x, 0] * su)
            out[y, x, 1] = F32(out[y, x, 1] * sv)
    return out


def _cv_round(v):
    """cvRound of a float: round half to even (SSE cvtss2si / lrint)."""
    return int(np.rint(v))


def cv_remap_linear(img, mapxy):
    """cv2.remap(img, mapxy, None, cv2.INTER_LINEAR): img (H, W, C) float32, mapxy (h, w, 2) float32 (x, y); constant-0 border."""
    img = np.asarray(img, np.float32)
    mapxy = np.asarray(mapxy, np.float32)
    h, w, cn = img.shape
    oh, ow = mapxy.shape[:2]
    out = np.empty((oh, ow, cn), np.float32)
    one, tab = F32(1.0), F32(INTER_TAB_SIZE)
    for y in range(oh):
        for x in range(ow):
            sx = _cv_round(F32(mapxy[y, x, 0] * tab)); sy = _cv_round(F32(mapxy[y, x, 1] * tab))      # fixed point, INTER_BITS = 5
            ix, iy = sx >> INTER_BITS, sy >> INTER_BITS
            fx = F32(F32(sx & (INTER_TAB_SIZE - 1)) / tab); fy = F32(F32(sy & (INTER_TAB_SIZE - 1)) / tab)
            w0 = F32(F32(one - fy) * F32(one - fx)); w1 = F32(F32(one - fy) * fx); w2 = F32(fy * F32(one - fx)); w3 = F32(fy * fx)
            for c in range(cn):
                def tap(yy, xx):
                    return img[yy, xx, c] if (0 <= xx < w and 0 <= yy < h) else F32(0.0)
                v = F32(F32(tap(iy, ix) * w0) + F32(tap(iy, ix + 1) * w1))
                v = F32(v + F32(tap(iy + 1, ix) * w2))
                out[y, x, c] = F32(v + F32(tap(iy + 1, ix + 1) * w3))
    return out


def cv_compute_consistency(flow12, flow21):
    """unwrap_utils.py:10-23: || flow12 + warp_flow(flow21, flow12) || with cv2.remap restated."""
    flow12 = np.asarray(flow12, np.float32)
    h, w = flow12.shape[:2]
    m = np.empty_like(flow12)
    for y in range(h):
        for x in range(w):
            m[y, x, 0] = F32(flow12[y, x, 0] + F32(x))         # flow[:, :, 0] += np.arange(w): one rounding of the exact sum
            m[y, x, 1] = F32(flow12[y, x, 1] + F32(y))
    warped = cv_remap_linear(np.asarray(flow21, np.float32), m)
    out = np.empty((h, w), np.float32)
    for y in range(h):
        for x in range(w):
            du = F32(flow12[y, x, 0] + warped[y, x, 0]); dv = F32(flow12[y, x, 1] + warped[y, x, 1])
            out[y, x] = np.sqrt(F32(F32(du * du) + F32(dv * dv)))      # numpy evaluates `array ** .5` as a correctly rounded sqrt (a scalar `**` goes through powf)
    return out
