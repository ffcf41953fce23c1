"""The pin that row (f)2 of the scope table still lacks: vectors written by the REAL OpenCV for the two calls inside the reference's
input builder (src/models/stage_1/unwrap_utils.py:10-38,68-70,127-131,143-162).  No image of this project carries cv2, so nothing
in the evidence chain has ever been produced by OpenCV itself; `oracle/cv_oracle.py` restates the published algorithm and is pinned
by hand-computed vectors only ("parity unpinned").  Run this script ONCE on any machine with opencv-python 4.x:

    python oracle/make_golden_cv2.py            # writes tests/golden/cv2_vectors.npz (inputs AND cv2's outputs, ~60 KB)

From then on tests/test_cv2_vectors.py holds, without cv2, (CPU) the restatement and the host loader's arithmetic and (GPU) the
device kernels `k_resize_bilinear` / `k_flow_consistency` against what OpenCV itself computed.  The cases are chosen where
implementations differ: up- and down-scales whose source taps clamp at both edges, 1-pixel-wide / 1-pixel-high sources, float64
frames (`astype(float64) / 255`) against float32 flows (work type double vs float), remap positions on exact 1/64 ties (cvRound is
round-half-to-even), taps straddling every border (constant 0), `resize_flow`'s per-channel rescale, and the consistency mask of a
random flow pair with norms straddling 1.  The FMA-dispatch ulp of OpenCV's AVX2 vertical pass (cv_oracle.py docstring) is RECORDED,
not assumed: the fixture stores the largest distance between cv2 and the restatement per case at generation time.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "cv2_vectors.npz")


def cases(rng):
    """name -> (kind, inputs...) ; everything the fixture needs to replay the call without cv2."""
    c = {}
    u8 = lambda h, w, ch=3: rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    # frames: uint8 -> float64 / 255 -> cv2.resize (unwrap_utils.py:127-131)
    for name, (h, w, nh, nw) in {"frame_down": (37, 53, 11, 20), "frame_up": (9, 7, 23, 31), "frame_down4": (48, 64, 12, 16),
                                 "frame_1col": (13, 1, 5, 4), "frame_1row": (1, 17, 3, 9), "frame_same": (6, 8, 6, 8),
                                 "frame_odd": (31, 29, 17, 30)}.items():
        c[name] = ("resize64", u8(h, w).astype(np.float64) / 255.0, nh, nw)
    # masks: single channel float64 (unwrap_utils.py:68-70)
    c["mask_down"] = ("resize64", (u8(40, 56, 1)[:, :, 0].astype(np.float64) / 255.0), 10, 14)
    # flows: float32, two channels (unwrap_utils.py:33-38)
    for name, (h, w, nh, nw) in {"flow_down": (36, 64, 9, 16), "flow_up": (7, 9, 20, 21), "flow_aniso": (30, 50, 17, 13)}.items():
        c[name] = ("resize_flow", (rng.standard_normal((h, w, 2)) * 4).astype(np.float32), nh, nw)
    # remap: random positions, exact ties of the 1/32 quantisation, positions outside every border
    img = (rng.standard_normal((12, 15, 2)) * 3).astype(np.float32)
    pos = np.stack((rng.uniform(-2.5, 16.5, (12, 15)), rng.uniform(-2.5, 13.5, (12, 15))), -1).astype(np.float32)
    c["remap_random"] = ("remap", img, pos)
    ties = np.zeros((4, 16, 2), np.float32)
    ties[..., 0] = (np.arange(16) * 2 + 1) / 64.0 + np.arange(4)[:, None] * 3          # k/32 + 1/64: cvRound ties, half to even
    ties[..., 1] = (np.arange(16)[::-1] * 2 + 1) / 64.0 + np.arange(4)[:, None] * 2
    c["remap_ties"] = ("remap", img, ties)
    edge = np.zeros((4, 6, 2), np.float32)
    edge[..., 0] = np.array([-1.0, -0.5, -0.015625, 13.984375, 14.5, 15.0])
    edge[..., 1] = np.array([-1.0, -0.25, 10.75, 11.5])[:, None]
    c["remap_edges"] = ("remap", img, edge)
    # consistency: || f12 + remap(f21, f12 + grid) ||, norms on both sides of 1 (unwrap_utils.py:10-23)
    f12 = (rng.standard_normal((14, 18, 2)) * 1.5).astype(np.float32)
    f21 = (-f12 + rng.standard_normal((14, 18, 2)).astype(np.float32) * 0.6).astype(np.float32)
    c["consistency"] = ("consistency", f12, f21)
    return c


def run_cv2(cv2, kind, *a):
    if kind == "resize64":
        img, nh, nw = a
        return cv2.resize(img, (nw, nh))
    if kind == "resize_flow":                           # unwrap_utils.py:33-38 verbatim
        flow, nh, nw = a
        oldh, oldw = flow.shape[0:2]
        flow = cv2.resize(flow, (nw, nh), interpolation=cv2.INTER_LINEAR)
        flow[:, :, 0] *= nh / oldh
        flow[:, :, 1] *= nw / oldw
        return flow
    if kind == "remap":
        img, pos = a
        return cv2.remap(img, pos, None, cv2.INTER_LINEAR)
    if kind == "consistency":                           # unwrap_utils.py:10-23 verbatim
        flow12, flow21 = a
        flow = flow12.copy()
        h, w = flow.shape[:2]
        flow[:, :, 0] += np.arange(w)
        flow[:, :, 1] += np.arange(h)[:, np.newaxis]
        wflow21 = cv2.remap(flow21, flow, None, cv2.INTER_LINEAR)
        diff = flow12 + wflow21
        return (diff[:, :, 0] ** 2 + diff[:, :, 1] ** 2) ** .5
    raise ValueError(kind)


def run_restatement(kind, *a):
    from oracle import cv_oracle as C
    if kind == "resize64":
        return C.cv_resize_linear(a[0], a[2], a[1])
    if kind == "resize_flow":
        return C.cv_resize_flow(a[0], a[1], a[2])
    if kind == "remap":
        return C.cv_remap_linear(a[0], a[1])
    return C.cv_compute_consistency(a[0], a[1])


def main():
    try:
        import cv2
    except ImportError:
        sys.exit("make_golden_cv2.py: `import cv2` failed - run this on a machine with opencv-python 4.x (none of this project's images has it); "
                 "tests/test_cv2_vectors.py skips until tests/golden/cv2_vectors.npz exists")
    cs = cases(np.random.default_rng(20260926))
    out = {"cv2_version": np.array(cv2.__version__), "names": np.array(sorted(cs))}
    for name, (kind, *a) in cs.items():
        got = run_cv2(cv2, kind, *[x.copy() if isinstance(x, np.ndarray) else x for x in a])
        mine = run_restatement(kind, *a)
        d = float(np.abs(np.asarray(got, np.float64) - np.asarray(mine, np.float64)).max())
        print("%-14s %-12s cv2 %s %s   max |cv2 - restatement| = %.3g%s" % (name, kind, got.shape, got.dtype, d, "" if d == 0 else "   <-- NOT bit-equal"))
        out[name + ".kind"] = np.array(kind)
        for i, x in enumerate(a):
            out["%s.in%d" % (name, i)] = np.asarray(x)
        out[name + ".out"] = got
        out[name + ".dist_at_generation"] = np.array(d)
    np.savez_compressed(OUT, **out)
    print("written", OUT, "with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()
