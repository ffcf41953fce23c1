"""Per-layer error of the chains' arithmetics MEASURED FROM THE KERNELS' OUTPUTS on the real tensors of the configs[1] schedule
(the admission rule of the f16x3 chains, mlphf.hip).

A training step leaves every hidden layer's input X_l and output relu(Z_l) (forward chain) and every dZ_l (backward chain) in HBM for the
weight-gradient GEMMs (include/atlasfit.h af_debug_tiles).  One layer's product is therefore checkable in isolation: recompute
Z_l = W_l X_l + b_l (forward kind, implicit_neural_networks.py:62-80) and dX_l = W_l^T dZ_l (backward kind, the dX half of
loss.backward(), stage1_neural_atlas.py:230) in fp64 FROM THE KERNEL'S OWN INPUT TILES and compare with what the kernel wrote, relative to
sum_k |a_k b_k| (the quantity every fp32 dot-product bound is stated in) — on the entries the ReLU lets through.

States: after the pre-train (iteration 0), after 5 000 and after 10 001 iterations of the shipped schedule on the field-flow video, both
nets, every 256-wide layer.  Arithmetics: mode 0 = v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain (the yardstick), 1 = bf16x6
(six bf16 products), 3 = f16x3 (two-term fp16 split, scale per row, three products).  Asserted for f16x3, the arithmetic admitted under this
rule: rms AND worst-case error no larger than the fp32 chain's for every net, kind and state (measured: 0.5-0.8x / 0.6-0.9x of it).  bf16x6,
admitted in round 2 on a numpy emulation, measures 0.9-1.3x of the fp32 chain's on the same tensors: held to 1.5x here; the figures are
printed for profiles/r6_gemm_error_from_kernels.txt."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MODES = {0: "fp32 MFMA (fmaf chain)", 1: "bf16x6", 3: "f16x3"}


def _rows(tiles):                # (nt, F, 32) T-layout tiles -> (nt * 32, F)
    return torch.from_numpy(np.ascontiguousarray(tiles)).cuda().permute(0, 2, 1).reshape(-1, tiles.shape[1])


def _measure(af, net, sd, rows_net, ntiles, pe_feats):
    """{(kind, layer): (rms, worst, n)} of one net from the tiles of the last step; sd = the parameters the step ran on."""
    nl = len([k for k in sd if k.endswith(".weight")])
    acts = {l: _rows(af.debug_tiles(net, "acts", l, rows_net, 0, ntiles)).double() for l in range(nl - 1)}
    dz = {l: _rows(af.debug_tiles(net, "dz", l, rows_net, 0, ntiles)).double() for l in range(nl - 1)}
    pe = _rows(af.debug_tiles(net, "pe", 0, rows_net, 0, ntiles)).double()[:, :pe_feats] if pe_feats else None
    out = {}
    for l in range(1, nl - 1):
        W = torch.from_numpy(sd["hidden.%d.weight" % l]).cuda().double()
        b = torch.from_numpy(sd["hidden.%d.bias" % l]).cuda().double()
        X = acts[l - 1]
        if W.shape[1] > 256:                                  # a skip layer: cat([x, PE(input)]) (implicit_neural_networks.py:66-69)
            X = torch.cat([X, pe], dim=1)
        Z = X @ W.T + b
        den = X.abs() @ W.abs().T + b.abs()
        got = acts[l]
        keep = (Z > 0) & (got > 0) & (den > 0)
        e = ((got - Z).abs() / den)[keep]
        out[("forward", l)] = (float(e.pow(2).mean().sqrt()), float(e.max()), int(keep.sum()))
        # backward: dZ_{l-1} = (dZ_l W_l[:, :256]) . [X_l > 0]
        Wh = W[:, :256]
        dX = dz[l] @ Wh
        den = dz[l].abs() @ Wh.abs()
        got = dz[l - 1]
        keep = (acts[l - 1] > 0) & (den > 0)
        e = ((got - dX).abs() / den)[keep]
        out[("backward", l)] = (float(e.pow(2).mean().sqrt()), float(e.max()), int(keep.sum()))
    return out


def test_layer_products_of_every_arithmetic_against_fp64_on_the_schedules_tensors():
    import aiod_amd
    import bench
    dev = torch.device("cuda", 0)
    resx, resy, F = 768, 432, 80
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F))
    af.upload_video(*bench.synth_video_device(resx, resy, F, seed=0, device=dev, flow="field"))
    sds = bench.init_state_dicts(0)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(100, seed=1)
    N = af.N
    g = torch.Generator().manual_seed(21)
    U = 2.0 ** -24
    worst_ratio = {}
    done = 0
    for tag, upto in (("after the pre-train", 0), ("after 5000 iterations", 5000), ("after 10001 iterations", 10001)):
        if upto > done:
            af.set_mlp_mode(1)
            af.train_steps(done, upto - done, None, seed=2, return_losses=False)
            done = upto
        state = {net: af.state_dict(net) for net in af.nets}
        adam = {net: af.adam_state(net) for net in af.nets}
        inds = torch.randint(F * resx * resy, (N,), generator=g).numpy()
        it = min(upto, 10000)
        nseg = 9 if it <= 5000 else 7
        res = {}
        for mode in MODES:
            af.set_mlp_mode(mode)
            for net in af.nets:
                af.load_state_dict(net, state[net])
                af.set_adam_state(net, *adam[net])
            af.train_steps(it, 1, inds, return_losses=False)
            res[mode] = {"mapping": _measure(af, aiod_amd.NET_MAPPING1, state[aiod_amd.NET_MAPPING1], nseg * N, 1500, 0),
                         "atlas": _measure(af, aiod_amd.NET_ATLAS, state[aiod_amd.NET_ATLAS], 3 * N, 900, 40)}
        for net in af.nets:                                   # leave the state as the schedule had it
            af.load_state_dict(net, state[net])
            af.set_adam_state(net, *adam[net])
        for name in ("mapping", "atlas"):
            for kind in ("forward", "backward"):
                layers = sorted(l for (k, l) in res[0][name] if k == kind)
                line = {}
                for mode in MODES:
                    rms = max(res[mode][name][(kind, l)][0] for l in layers)
                    wst = max(res[mode][name][(kind, l)][1] for l in layers)
                    line[mode] = (rms, wst)
                n = sum(res[0][name][(kind, l)][2] for l in layers)
                print("%-24s %-7s %-8s (%d layers, %.1e entries; worst layer, units of 2^-24): " % (tag, name, kind, len(layers), n)
                      + "   ".join("%s rms %.2f worst %.1f" % (MODES[m], line[m][0] / U, line[m][1] / U) for m in MODES), flush=True)
                for m in (1, 3):
                    worst_ratio[(tag, name, kind, m)] = (line[m][0] / line[0][0], line[m][1] / line[0][1])
    for key, (r_rms, r_worst) in sorted(worst_ratio.items()):
        lim = 1.0 if key[3] == 3 else 1.5          # bf16x6: measured 0.9-1.3x over the runs of round 6 (the worst case over 1e7 entries is an extreme-value statistic)
        assert r_rms <= lim, ("rms error above the fp32 chain's", key, r_rms)
        assert r_worst <= lim, ("worst-case error above the fp32 chain's", key, r_worst)
    print("largest rms / worst-case ratio to the fp32 chain: bf16x6 %.2f / %.2f, f16x3 %.2f / %.2f"
          % tuple(max(v[i] for k, v in worst_ratio.items() if k[3] == m) for m in (1, 3) for i in (0, 1)))
    af.close()


def _dw_errors(af, net, sd, rows_net, ntiles, pe_feats, grads):
    """{layer: (rms, worst)} of the weight gradient of the hidden layers: sum over the first `ntiles` row tiles recomputed in fp64 from the kernels' own dZ_l
    and X_l tiles — NOT comparable with the kernel's gradient, which sums every live row; so the kernel is run on a batch whose rows beyond those tiles
    do not exist: see the caller (the whole batch's tiles are read)."""
    nl = len([k for k in sd if k.endswith(".weight")])
    out = {}
    off = 0
    shapes = [(sd["hidden.%d.weight" % l].shape, sd["hidden.%d.bias" % l].shape) for l in range(nl)]
    offs = []
    for (ws, bs) in shapes:
        offs.append(off); off += ws[0] * ws[1] + bs[0]
    acts = {}
    for l in range(1, nl - 1):
        ws = shapes[l][0]
        if l - 1 not in acts:
            acts[l - 1] = _rows(af.debug_tiles(net, "acts", l - 1, rows_net, 0, ntiles)).double()
        dz = _rows(af.debug_tiles(net, "dz", l, rows_net, 0, ntiles)).double()
        X = acts[l - 1]
        ref = dz.T @ X                                          # (256 out, 256 in)
        den = dz.abs().T @ X.abs()
        got = torch.from_numpy(grads[offs[l]:offs[l] + ws[0] * ws[1]].reshape(ws)).cuda().double()[:, :256]
        keep = den > 0
        e = ((got - ref).abs() / den)[keep]
        out[l] = (float(e.pow(2).mean().sqrt()), float(e.max()))
        del acts[l - 1]
    return out


def test_weight_gradient_of_every_arithmetic_against_fp64_on_the_schedules_tensors():
    """The same yardstick for k_dw (dW_l = dZ_l^T X_l over the whole batch: split-K partial sums per workgroup, summed in fp32 by k_adam): every hidden
    layer's gradient block recomputed in fp64 from the kernels' own tiles, per ENTRY relative to sum_r |dZ[r][o] X[r][i]|; dw modes 0 (fp32 MFMA) and 1
    (bf16x6, the default) on the f16x3 chains' tensors at iterations 0 / 5 000 / 10 001.  Asserted: bf16x6 within 1.6x of the fp32 MFMA's rms and 3x of its
    worst case (measured 0.8-1.4x / 0.8-2.3x: the six products' dropped terms show on single entries, not in the norm).  (Round 6 also built k_dw on three fp16 products with RECIPROCAL row scales and measured it with this test: fp32-grade in norm, but an entry
    whose significant rows all lie 2^17 below the segment's dominant rows loses bits to fp16's subnormal floor — rms 13, worst 1 700 units of 2^-24 at
    iteration 5 000 where fp32 has 0.65 / 7.6 — and only 2.4 % faster in the step: not shipped; tools/experiments/README.md, profiles/r6_k_dw_hf_experiment_*.)"""
    import aiod_amd
    import bench
    dev = torch.device("cuda", 0)
    resx, resy, F = 768, 432, 80
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F))
    af.upload_video(*bench.synth_video_device(resx, resy, F, seed=0, device=dev, flow="field"))
    sds = bench.init_state_dicts(0)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(100, seed=1)
    N = af.N
    g = torch.Generator().manual_seed(22)
    U = 2.0 ** -24
    names = {0: "fp32 MFMA", 1: "bf16x6"}
    done, ratios = 0, {}
    for tag, upto in (("after the pre-train", 0), ("after 5000 iterations", 5000), ("after 10001 iterations", 10001)):
        if upto > done:
            af.set_dw_mode(1)
            af.train_steps(done, upto - done, None, seed=2, return_losses=False)
            done = upto
        state = {net: af.state_dict(net) for net in af.nets}
        adam = {net: af.adam_state(net) for net in af.nets}
        inds = torch.randint(F * resx * resy, (N,), generator=g).numpy()
        it = min(upto, 10000)
        nseg = 9 if it <= 5000 else 7
        res = {}
        for mode in names:
            af.set_dw_mode(mode)
            for net in af.nets:
                af.load_state_dict(net, state[net])
                af.set_adam_state(net, *adam[net])
            af.set_debug(True)
            losses = af.train_steps(it, 1, inds)[0]
            af.set_debug(False)
            live = int((5 if nseg == 7 else 7) * N + losses[6] + losses[7])                    # mapping rows that exist this iteration (valid flow matches compacted)
            res[mode] = {"mapping": _dw_errors(af, aiod_amd.NET_MAPPING1, state[aiod_amd.NET_MAPPING1], nseg * N, (live + 31) // 32, 0, af.last_grads(aiod_amd.NET_MAPPING1)),
                         "atlas": _dw_errors(af, aiod_amd.NET_ATLAS, state[aiod_amd.NET_ATLAS], 3 * N, (3 * N + 31) // 32, 40, af.last_grads(aiod_amd.NET_ATLAS))}
        for net in af.nets:
            af.load_state_dict(net, state[net])
            af.set_adam_state(net, *adam[net])
        for name in ("mapping", "atlas"):
            line = {m: (max(v[0] for v in res[m][name].values()), max(v[1] for v in res[m][name].values())) for m in names}
            print("%-24s %-7s weight gradient (%d layers; worst layer, units of 2^-24): " % (tag, name, len(res[0][name]))
                  + "   ".join("%s rms %.2f worst %.1f" % (names[m], line[m][0] / U, line[m][1] / U) for m in names), flush=True)
            ratios[(tag, name, 1)] = (line[1][0] / line[0][0], line[1][1] / line[0][1])
    af.set_dw_mode(1)
    for key, (r_rms, r_worst) in sorted(ratios.items()):
        assert r_rms <= 1.6 and r_worst <= 3.0, (key, r_rms, r_worst)          # measured over the runs of round 6: rms 0.8-1.4x, worst case 0.8-2.3x
    print("largest rms / worst-case ratio to the fp32 MFMA k_dw: bf16x6 %.2f / %.2f" % tuple(max(v[i] for v in ratios.values()) for i in (0, 1)))
    af.close()
