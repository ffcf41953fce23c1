"""Consumes tests/golden/cv2_vectors.npz — outputs of the REAL OpenCV for the calls of the reference's input builder, written by
oracle/make_golden_cv2.py on a machine that has cv2 (no image of this project does: until someone runs it these tests skip and row
(f)2 of the scope table stays "parity unpinned").  With the fixture: the restatement (oracle/cv_oracle.py), the host loader's
vectorised arithmetic (stage1.py) and - on the GPU - the device kernels k_resize_bilinear / k_flow_consistency are held against what
OpenCV itself computed.  Bit-equality is asserted for the float64 frames / masks, the remap and the consistency norm; for float32
resizes (the flows) one ulp is admitted IF AND ONLY IF the fixture recorded such a distance at generation time (OpenCV's AVX2 / FMA3
dispatch fuses one multiply-add of the vertical pass: cv_oracle.py docstring) - the tolerance is what the fixture shows, not assumed."""
import os

import numpy as np
import pytest

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cv2_vectors.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(FIX), reason="tests/golden/cv2_vectors.npz not generated yet (needs a machine with cv2: python oracle/make_golden_cv2.py)")


def _cases():
    g = np.load(FIX)
    for name in [str(n) for n in g["names"]]:
        kind = str(g[name + ".kind"])
        ins = [g["%s.in%d" % (name, i)] for i in range(3) if "%s.in%d" % (name, i) in g]
        yield name, kind, ins, g[name + ".out"], float(g[name + ".dist_at_generation"])


def _tol(kind, out, dist):
    """0 unless the fixture itself saw the FMA ulp on a float32 resize."""
    if kind == "resize_flow" and dist > 0:
        assert dist <= 4e-7 * float(np.abs(out).max()), "the fixture's own distance is more than one ulp: not the FMA effect"
        return dist
    assert dist == 0.0 or kind == "resize_flow", "the restatement was not bit-equal to cv2 when the fixture was written"
    return 0.0


def test_restatement_equals_opencv():
    from oracle import cv_oracle as C
    for name, kind, ins, out, dist in _cases():
        if kind == "resize64":
            got = C.cv_resize_linear(ins[0], int(ins[2]), int(ins[1]))
        elif kind == "resize_flow":
            got = C.cv_resize_flow(ins[0], int(ins[1]), int(ins[2]))
        elif kind == "remap":
            got = C.cv_remap_linear(ins[0], ins[1])
        else:
            got = C.cv_compute_consistency(ins[0], ins[1])
        assert got.shape == out.shape and got.dtype == out.dtype, name
        assert np.abs(got.astype(np.float64) - out.astype(np.float64)).max() <= _tol(kind, out, dist), name


@pytest.mark.gpu
def test_device_input_builder_equals_opencv():
    """k_resize_bilinear / k_flow_consistency through the C ABI against cv2's outputs (frames enter as uint8, as in the loader)."""
    import torch
    import aiod_amd
    A = aiod_amd.atlasfit
    dev = torch.device("cuda", 0)
    for name, kind, ins, out, dist in _cases():
        if kind == "resize64":
            u8 = np.rint(ins[0] * 255.0).astype(np.uint8)
            assert np.array_equal(u8.astype(np.float64) / 255.0, ins[0]), name          # the fixture's float64 frames are exact uint8 / 255
            src = torch.from_numpy(u8.reshape(u8.shape[0], u8.shape[1], -1)).to(dev).contiguous()
            nh, nw, ch = int(ins[1]), int(ins[2]), src.shape[2]
            dst = torch.zeros(nh * nw * ch, device=dev)
            A.resize_bilinear_device(src, dst, nh, nw, ch, 1, 0)
            got = dst.view(nh, nw, ch).cpu().numpy().reshape(out.shape)
            assert np.array_equal(got, out.astype(np.float32)), name                   # the loader stores float32(cv2's float64 result)
        elif kind == "resize_flow":
            src = torch.from_numpy(ins[0]).to(dev).contiguous()
            nh, nw = int(ins[1]), int(ins[2])
            dst = torch.zeros(nh * nw * 2, device=dev)
            A.resize_bilinear_device(src, dst, nh, nw, 2, 1, 0, scale=(nh / ins[0].shape[0], nw / ins[0].shape[1]))
            got = dst.view(nh, nw, 2).cpu().numpy()
            assert np.abs(got.astype(np.float64) - out.astype(np.float64)).max() <= _tol(kind, out, dist), name
        elif kind == "consistency":
            f12, f21 = (torch.from_numpy(x).to(dev).contiguous() for x in ins[:2])
            h, w = ins[0].shape[:2]
            dst = torch.zeros(h * w, device=dev)
            A.flow_consistency_device(f12, f21, dst, 1, 0, thresh=0.0)
            assert np.array_equal(dst.view(h, w).cpu().numpy(), out), name
            A.flow_consistency_device(f12, f21, dst, 1, 0, thresh=1.0)
            assert np.array_equal(dst.view(h, w).cpu().numpy(), (out < 1.0).astype(np.float32)), name
