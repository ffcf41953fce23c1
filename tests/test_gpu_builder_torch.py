"""A third opinion on the input builder that runs HERE: OpenCV is not in this image, so
k_resize_bilinear / k_flow_consistency (the device side of load_input_data, unwrap_utils.py:10-38,127-131) have only ever been
compared with this repository's own restatement of cv2 (oracle/cv_oracle.py).  torch ships independent implementations of the same
geometry:

  * cv2.resize(..., INTER_LINEAR) samples at the half-pixel centres src = (dst + 0.5) * scale - 0.5 with the edge clamped —
    exactly torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=False).  Evaluated in fp64 it is the exact
    bilinear value; OpenCV (and this kernel) form the source coordinate in float32 — `fx = (float)((dx + 0.5) * scale - 0.5); fx -= sx` —
    so a coefficient carries the rounding of a coordinate of magnitude ~W (6e-8 W), times the difference of the two taps: agreement to
    ~1e-7 x W x (value spread), not bit for bit (exact for the integer scale factors of the shipped --down 4, where the coefficients are 0.5).
  * cv2.remap(flow21, flow12 + grid, INTER_LINEAR) with the default constant-0 border is
    grid_sample(mode="bilinear", padding_mode="zeros", align_corners=True) on pixel coordinates — up to cv2's quantisation of the
    sampling position to 1/32 px (INTER_BITS = 5), which moves a sampled flow by at most (1/64 px) x its local gradient.  The consistency
    NORM therefore agrees within that band and the MASK (norm < 1) may flip only where the exact norm lies inside the band around 1.

This pins the geometry, the transposes, the channel order, the borders and the threshold against code the builder did not write;
row (f)2 stays "parity unpinned" until cv2 itself runs (tests/test_cv2_vectors.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src_hw,dst_hw", [((360, 640), (90, 160)), ((1080, 1920), (432, 768)), ((97, 131), (211, 173)), ((64, 48), (64, 48))])
@pytest.mark.parametrize("u8", [True, False])
def test_resize_against_torch_interpolate(src_hw, dst_hw, u8):
    import aiod_amd.atlasfit as A
    g = torch.Generator().manual_seed(src_hw[0] * 7 + dst_hw[1])
    sh, sw = src_hw
    dh, dw = dst_hw
    if u8:
        src = torch.randint(0, 256, (sh, sw, 3), generator=g, dtype=torch.uint8)
        ref_in = src.double() / 255.0
    else:
        src = (torch.rand(sh, sw, 3, generator=g) * 40 - 20).float()
        ref_in = src.double()
    dst = torch.empty(dh, dw, 3, device="cuda")
    A.resize_bilinear_device(src.cuda().contiguous(), dst, dh, dw, 3, 1, 0)
    ref = torch.nn.functional.interpolate(ref_in.permute(2, 0, 1)[None].cuda(), size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0)
    err = (dst.double() - ref).abs().max().item()
    span = float(ref_in.abs().max())
    print("resize %s -> %s %s: max |kernel - torch fp64 interpolate| = %.3g (value range %.3g)" % (src_hw, dst_hw, "uint8" if u8 else "float32", err, span))
    # a float32 coordinate of magnitude <= max(sh, sw) (half an ulp each for the product and the subtraction) times the spread of two taps (<= 2 span),
    # plus one float32 rounding of the result.  A wrong half-pixel convention, a transposed axis or a clamped edge would be O(span).
    tol = 1.5e-7 * max(sh, sw) * 2.0 * max(span, 1.0) + 2e-7 * max(span, 1.0)
    assert err <= tol, (err, tol)


@pytest.mark.parametrize("hw", [(90, 160), (432, 768)])
def test_flow_consistency_against_torch_grid_sample(hw):
    import aiod_amd.atlasfit as A
    h, w = hw
    g = torch.Generator().manual_seed(h)
    # a smooth forward flow with sub-pixel structure and a backward flow that is its approximate inverse plus errors of the order of the threshold
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    f12 = torch.stack([3.3 + 2.0 * torch.sin(yy / 17.0) + 0.7 * torch.cos(xx / 11.0), -2.1 + 1.5 * torch.cos(yy / 13.0 + xx / 29.0)], dim=-1)
    f21 = -f12 + 0.9 * torch.stack([torch.sin(xx / 5.0 + yy / 7.0), torch.cos(xx / 6.0 - yy / 9.0)], dim=-1) + 0.05 * torch.randn(h, w, 2, generator=g, dtype=torch.float64)
    f12f, f21f = f12.float().contiguous(), f21.float().contiguous()
    nrm = torch.empty(h, w, device="cuda"); msk = torch.empty(h, w, device="cuda")
    A.flow_consistency_device(f12f.cuda(), f21f.cuda(), nrm, 1, 0, thresh=0.0)
    A.flow_consistency_device(f12f.cuda(), f21f.cuda(), msk, 1, 0, thresh=1.0)
    # torch: sample flow21 at (x + u, y + v), bilinear, zeros outside, pixel-centre coordinates (align_corners=True maps -1..1 onto 0..w-1)
    a, b = f12f.double().cuda(), f21f.double().cuda()
    px = xx.cuda() + a[..., 0]; py = yy.cuda() + a[..., 1]
    grid = torch.stack([px / (w - 1) * 2 - 1, py / (h - 1) * 2 - 1], dim=-1)[None]
    warped = torch.nn.functional.grid_sample(b.permute(2, 0, 1)[None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0].permute(1, 2, 0)
    ref = (a + warped).pow(2).sum(-1).sqrt()
    # the 1/32-px quantisation moves the sampling position by <= 1/64 px per axis: the warped flow by <= (|d/dx| + |d/dy|) / 64 of flow21
    gx = (b[:, 1:] - b[:, :-1]).abs().amax(); gy = (b[1:] - b[:-1]).abs().amax()
    band = float((gx + gy) / 64.0) * np.sqrt(2.0) + 1e-5
    inside = (px >= 1) & (px <= w - 2) & (py >= 1) & (py <= h - 2)          # off the border the zero padding makes the field discontinuous: compared apart
    err = (nrm.double() - ref).abs()
    print("consistency %s: max |norm - torch grid_sample| inside %.4f px (band %.4f), at the border %.4f; mask flips %d, all within the band: %s"
          % (hw, float(err[inside].max()), band, float(err[~inside].max()) if (~inside).any() else 0.0,
             int(((ref < 1.0) != (msk > 0.5)).sum()), bool((((ref < 1.0) != (msk > 0.5)) & ((ref - 1.0).abs() > band) & inside).sum() == 0)))
    assert float(err[inside].max()) <= band
    flips = (ref < 1.0) != (msk > 0.5)
    assert int((flips & inside & ((ref - 1.0).abs() > band)).sum()) == 0                   # a flip only where the exact norm is within the band of the threshold
    assert float((flips & inside).double().mean()) < 0.02
    # samples that leave the frame: both read zeros there (constant border), so the norm is |flow12| up to the partial taps at the rim
    far = (px < -1) | (px > w) | (py < -1) | (py > h)
    if far.any():
        assert float((nrm.double() - a.pow(2).sum(-1).sqrt()).abs()[far].max()) <= 1e-5
