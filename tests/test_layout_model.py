"""A numpy model of the register / LDS / HBM layouts the HIP kernels rely on (af_dev.h), executed with an
emulated v_mfma_f32_32x32x2_f32.  It pins the index algebra on the CPU: C-layout chaining between layers,
the packed weight image, the T-layout tiles with the dW source swizzle, and the PE slot permutations."""
import numpy as np


def mfma_32x32x2(a, b, c):
    """a[64], b[64]: lane l holds A[i=l&31][k=l>>5] / B[k=l>>5][j=l&31]; c[64][16]: lane l, reg r holds
    D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]   (cdna_hip_programming.md §3)."""
    A = np.zeros((32, 2), np.float64); B = np.zeros((2, 32), np.float64)
    for l in range(64):
        A[l & 31, l >> 5] = a[l]; B[l >> 5, l & 31] = b[l]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def img_index(mpad, m, k):          # af_img_index
    return ((((k >> 3) * 2 + ((k >> 2) & 1)) * mpad + m) << 2) + (k & 3)


def pack_image(W, mpad):            # W[m][k] -> packed image (floats)
    M, K = W.shape
    img = np.zeros(((K + 7) // 8) * 2 * mpad * 4)
    for m in range(M):
        for k in range(K):
            img[img_index(mpad, m, k)] = W[m, k]
    return img


def layer_transposed(img, mpad, mtiles, ngroups, breg):
    """acc[T][lane][16] = sum over k-groups of MFMA steps; breg[lane][4*g+p] is the B operand (C-layout)."""
    acc = [np.zeros((64, 16)) for _ in range(mtiles)]
    for g in range(ngroups):
        for p in range(4):
            for T in range(mtiles):
                a = np.array([img[(((g * 2 + (l >> 5)) * mpad + 32 * T + (l & 31)) << 2) + p] for l in range(64)])
                acc[T] = mfma_32x32x2(a, breg[:, 4 * g + p], acc[T])
    return acc


def c_layout_of(X):                 # X[32 rows][256] -> regs[lane][128]
    regs = np.zeros((64, X.shape[1] // 2))
    for l in range(64):
        j, h = l & 31, l >> 5
        for rho in range(X.shape[1] // 2):
            T, q, p = rho >> 4, (rho >> 2) & 3, rho & 3
            regs[l, rho] = X[j, 32 * T + 8 * q + 4 * h + p]
    return regs


def test_chained_layers_stay_in_registers():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((32, 64)); W1 = rng.standard_normal((64, 64)); W2 = rng.standard_normal((32, 64))
    acc1 = layer_transposed(pack_image(W1, 64), 64, 2, 8, c_layout_of(X))
    regs = np.concatenate(acc1, axis=1)                     # output of layer 1 IS the C-layout of its result
    assert np.allclose(regs, c_layout_of(X @ W1.T))
    acc2 = layer_transposed(pack_image(W2, 32), 32, 1, 8, regs)
    Y = X @ W1.T @ W2.T
    for l in range(64):
        for r in range(16):
            assert abs(acc2[0][l, r] - Y[l & 31, (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]) < 1e-9


def test_dw_tiles_with_source_swizzle():
    rng = np.random.default_rng(1)
    dZ = rng.standard_normal((32, 32)); Xa = rng.standard_normal((32, 32))     # one row tile, one feature tile each
    def t_layout(M):                # [f][32 rows]
        return M.T.copy().reshape(-1)
    def stage(tile):                # LDS slot s=(f*8+c) <- global slot f*8 + (c ^ ((f>>1)&7))
        lds = np.zeros_like(tile)
        for s in range(32 * 8):
            f, c = s >> 3, s & 7
            src = (f << 3) + (c ^ ((f >> 1) & 7))
            lds[s * 4:s * 4 + 4] = tile[src * 4:src * 4 + 4]
        return lds
    la, lb = stage(t_layout(dZ)), stage(t_layout(Xa))
    acc = np.zeros((64, 16)); db = np.zeros(64)
    for g in range(4):
        def frag(lds):
            out = np.zeros((64, 4))
            for l in range(64):
                m, h = l & 31, l >> 5
                off = m * 32 + (((2 * g + h) ^ ((m >> 1) & 7)) << 2)
                out[l] = lds[off:off + 4]
            return out
        af, bf = frag(la), frag(lb)
        db += af.sum(1)
        for p in range(4):
            acc = mfma_32x32x2(af[:, p], bf[:, p], acc)
    dW = dZ.T @ Xa
    for l in range(64):
        for r in range(16):
            assert abs(acc[l, r] - dW[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]) < 1e-9
    assert np.allclose(db[:32] + db[32:], dZ.sum(0))


def pe_slot_of_feature(kind, e):    # af_pe_slot_of_feature
    if kind != 2:
        return e
    if e < 24:
        k, c = divmod(e, 6); h = k >> 1; rho = (k & 1) * 6 + c
    else:
        c = e - 24; h = c // 3; rho = 12 + c % 3
    return ((rho >> 2) << 3) + (h << 2) + (rho & 3)


def test_pe_slot_permutations_are_injective():
    assert [pe_slot_of_feature(1, e) for e in range(40)] == list(range(40))
    slots = [pe_slot_of_feature(2, e) for e in range(30)]
    assert len(set(slots)) == 30 and max(slots) < 32


def test_imlp_shapes_follow_the_config_and_match_the_reference_counts():
    """atlasfit.imlp_shapes derives the layer shapes from the AfConfig: the shipped config gives the reference's
    parameter counts, a different architecture gives different shapes (which AtlasFit.__init__ then refuses
    against the library's own count before anything reaches the C side)."""
    import aiod_amd
    A = aiod_amd.atlasfit
    cfg = A.default_config(64, 48, 4, two_layer=True)
    counts = {net: sum(o * k + o for o, k in A.imlp_shapes(net, cfg)) for net in (A.NET_MAPPING1, A.NET_ATLAS, A.NET_MAPPING2, A.NET_ALPHA)}
    assert counts == {A.NET_MAPPING1: 264706, A.NET_ATLAS: 416379, A.NET_MAPPING2: 133122, A.NET_ALPHA: 402945}
    assert A.imlp_shapes(A.NET_ATLAS, cfg) == A.imlp_shapes(A.NET_ATLAS)            # the defaults are the shipped architecture
    assert A.imlp_shapes(A.NET_ATLAS, cfg)[4] == (256, 296) and A.imlp_shapes(A.NET_ATLAS, cfg)[7] == (3, 296)
    other = A.default_config(64, 48, 4, number_of_channels_atlas=128, number_of_layers_mapping1=4, positional_encoding_num_atlas=6)
    assert A.imlp_shapes(A.NET_ATLAS, other)[0] == (128, 24) and len(A.imlp_shapes(A.NET_MAPPING1, other)) == 4
    assert sum(o * k + o for o, k in A.imlp_shapes(A.NET_ATLAS, other)) != 416379


def test_committed_traffic_summary_agrees_with_the_byte_model():
    """bench.py quotes the dominant kernel's measured HBM traffic from the newest profiles/r*_traffic.json (written by
    tools/traffic_from_pmc.py from the rocprofv3 PMC passes) - and only while the measurement agrees with the byte count the
    layouts imply for the rows of the run (bench.hbm_model_bytes).  Guarded here by BYTES, not by names: every hot kernel of the
    default arithmetic is in the committed summary, within 12 % of the model of the profiled command (tools/collect_profiles.sh:
    40 steps centred on the global-rigidity switch, the synthetic video's 0.9826 valid matches)."""
    import bench
    N = 10000
    rows = [bench.model_rows(i, N, 0.9826, False) for i in range(4981, 5021)]
    mean = tuple(sum(r[j] for r in rows) / len(rows) for j in range(4))
    assert abs(mean[0] - (5 + 1 + 2 * 0.9826) * N) < 1 and mean[1] == 3 * N
    for cls, launches in (("fwd_1", 2), ("bwd_1", 2), ("dw", 1)):
        name = bench.KERNEL_OF_CLASS[cls]
        model = bench.hbm_model_bytes(name, mean, launches)
        got, src = bench.committed_traffic(name, model, tol=0.12)
        assert got is not None, (name, src)
        assert abs(got / model - 1.0) <= 0.12, (name, got, model)
    # a stale summary (another row mix: twice the rows) is refused, not quoted
    stale, why = bench.committed_traffic(bench.KERNEL_OF_CLASS["dw"], 2 * bench.hbm_model_bytes("k_dw", mean, 1), tol=0.12)
    assert stale is None and "not quoted" in why
