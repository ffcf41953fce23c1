"""CPU restatement of the split-operand arithmetic of the bf16 kernels (all-in-one-deflicker_amd/csrc/bfsplit.h): every fp32
operand is split into three bf16 values (round-to-nearest-even at each level, residuals by exact fp32 subtraction) and a
product is accumulated as hh + hm + mh + mm + hl + lh in fp32 ("bf16x6": the chains, k_dw_bf<6>).  Checked here: the split
is (nearly) exact, and the contraction dW = dZ^T X carries fp32-level round-off against an fp64 reference — no more than a
plain fp32 GEMM.  The three-product form (hi + mid only: k_dw_bf<3>, an opt-in of the weight-gradient GEMM since round 3, never the default) is visibly
coarser per contraction and still far inside what separates an fp32 gradient from its fp64 twin on real batches
(tests/test_gpu_fullsize.py holds it to that)."""
import numpy as np
import torch


def split3(x):
    h = x.to(torch.bfloat16).float(); r = x - h
    m = r.to(torch.bfloat16).float(); q = r - m
    return h, m, q.to(torch.bfloat16).float()


def contract_x6(a, b):
    ah, am, al = split3(a); bh, bm, bl = split3(b)
    out = ah @ bl + al @ bh          # same term order as dw_segment_bf: the small products first
    out = out + am @ bm
    out = out + ah @ bm
    out = out + am @ bh
    return out + ah @ bh


def test_three_way_split_is_exact_to_fp32_rounding():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(200000, generator=g) * torch.logspace(-6, 3, 200000)
    h, m, l = split3(x)
    err = ((h.double() + m.double() + l.double()) - x.double()).abs() / x.double().abs()
    assert float(err.max()) <= 2.0 ** -24                      # 8 + 8 + 8 mantissa bits
    for part in (h, m, l):                                     # every level is a bf16 value: low 16 bits of the fp32 pattern are zero
        assert int((part.view(torch.int32) & 0xFFFF).abs().max()) == 0


def test_six_term_contraction_has_fp32_level_error():
    g = torch.Generator().manual_seed(1)
    K = 40000                                                  # rows of a batch (the contraction dimension of dW)
    dz = torch.randn(K, 256, generator=g) * torch.rand(K, 1, generator=g) * (torch.rand(K, 256, generator=g) > 0.5) * 1e-3
    x = torch.relu(torch.randn(K, 256, generator=g) * 0.5 + 0.1)
    ref = dz.double().T @ x.double()
    e32 = float((((dz.T @ x).double() - ref).norm() / ref.norm()))
    e6 = float(((contract_x6(dz.T.contiguous(), x).double() - ref).norm() / ref.norm()))
    ah, am, _ = split3(dz.T.contiguous()); bh, bm, _ = split3(x)
    e3 = float((((ah @ bh + ah @ bm + am @ bh).double() - ref).norm() / ref.norm()))
    print("relative error vs fp64: fp32 GEMM %.3g, bf16x6 %.3g, bf16x3 %.3g" % (e32, e6, e3))
    assert e6 < 2.0 * e32 + 1e-7                               # six terms: fp32-class
    assert e3 > 5.0 * e6                                       # three terms (16 mantissa bits) are visibly coarser ...
    assert e3 < 2e-5                                           # ... at the 1e-6 level of the result's norm
