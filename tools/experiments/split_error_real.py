#!/usr/bin/env python
"""Error of the candidate arithmetics on the REAL tensors of the loop (the evidence a two-term fp16 split would be judged on;
the kernels for it do not exist — DESIGN.md 7 item 1).

Inputs: the nets' parameters after the pre-train, after 5 000 and after 10 001 iterations of the configs[1] schedule, dumped on the MI355X by
`python tools/experiments/split_error_real.py --dump states.npz` (GPU box), then on any CPU:

    python tools/experiments/split_error_real.py states.npz

For each state the oracle runs one batch (N = 10 000) forward and backward with hooks and collects, per hidden layer of both nets, X_l (the layer's
input, post-ReLU), dZ_l (the gradient at its pre-activation) and W_l; then three kinds of dot products are sampled —
forward  y = sum_i W[o,i] X[r,i] (K = 256), backward dx = sum_o W[o,i] dZ[r,o] (K = 256), weight gradient dw = sum_r dZ[r,o] X[r,i] over one
workgroup's segment of rows (K = 352 = 11 row tiles) — and evaluated in fp64 (the truth), as an fp32 fmaf chain (what "fp32" means on a CPU),
in bf16x6 as the kernels do it (three bf16 terms per operand, the six leading products, fp32 accumulation per 16-wide MFMA), in bf16x3 (the
opt-in k_dw), and in the two-term fp16 split with one power-of-two scale per tensor (3 and 4 products).  Errors are quoted relative to
sum_k |a_k b_k| (the quantity every fp32 error bound of a dot product is stated in): rms and WORST case over the samples."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dump(path):
    import torch, aiod_amd, bench
    dev = torch.device("cuda", 0)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80))
    af.upload_video(*bench.synth_video_device(768, 432, 80, seed=0, device=dev, flow="field"))
    sds = bench.init_state_dicts(0)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(100, seed=1)
    out = {}
    def snap(tag):
        for net in af.nets:
            out["%s_net%d" % (tag, net)] = af.get_params_flat(net)
    snap("it0")
    af.train_steps(0, 5000, None, seed=2, return_losses=False); snap("it5000")
    af.train_steps(5000, 5001, None, seed=2, return_losses=False); snap("it10001")
    np.savez_compressed(path, **out)
    print("written", path, {k: v.shape for k, v in out.items()})


def bf16(x):
    x = np.asarray(x, np.float32); u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)
def split_bf3(x):
    h = bf16(x); r1 = (x - h).astype(np.float32); m = bf16(r1); r2 = (r1 - m).astype(np.float32); return h, m, bf16(r2)
def split_f16(x, s):
    xs = (x * np.float32(s)).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32); l = (xs - h).astype(np.float32).astype(np.float16).astype(np.float32)
    return h, l
def mfma_acc(terms):        # products of a 16-wide group summed exactly, one fp32 rounding into the accumulator per MFMA (terms in issue order)
    M, K = terms[0][0].shape
    acc = np.zeros(M, np.float32)
    for k0 in range(0, K, 16):
        for a, b in terms:
            acc = (acc.astype(np.float64) + (a[:, k0:k0 + 16].astype(np.float64) * b[:, k0:k0 + 16].astype(np.float64)).sum(1)).astype(np.float32)
    return acc
def fmaf_chain(a, b):
    acc = np.zeros(a.shape[0], np.float32)
    for k in range(a.shape[1]):
        acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)
    return acc
def pow2_scale(t):          # per-tensor power of two that puts the largest magnitude just below 2^14 (fp16 max 65504 = 2^16: two binades of headroom)
    m = float(np.abs(t).max())
    return 2.0 ** (14 - np.ceil(np.log2(m))) if m > 0 else 1.0


def study_full_dw(name, dz_cols, x_cols, sz, sx, res, seg=352):
    """The weight gradient as k_dw + k_adam form it: every 352-row segment accumulated on its own (a workgroup's partial block), the partial sums added
    in fp32 in fixed order.  dz_cols, x_cols: (samples, rows) — column o of dZ and column i of X for each sampled entry dW[o, i]."""
    ref = (dz_cols.astype(np.float64) * x_cols.astype(np.float64)).sum(1)
    denom = (np.abs(dz_cols.astype(np.float64)) * np.abs(x_cols.astype(np.float64))).sum(1)
    keep = denom > 0
    R = dz_cols.shape[1] // seg * seg
    def run(terms_of):
        tot = np.zeros(dz_cols.shape[0], np.float32)
        for r0 in range(0, R, seg):
            tot = (tot.astype(np.float64) + terms_of(slice(r0, r0 + seg)).astype(np.float64)).astype(np.float32)
        return tot
    ah, am, al = split_bf3(dz_cols); bh, bm, bl = split_bf3(x_cols)
    fh, fl = split_f16(dz_cols, sz); gh, gl = split_f16(x_cols, sx)
    inv = np.float64(1.0 / (sz * sx))
    e = {"fp32 fmaf chain": run(lambda q: fmaf_chain(dz_cols[:, q], x_cols[:, q])),
         "bf16x6 (shipped)": run(lambda q: mfma_acc([(al[:, q], bh[:, q]), (ah[:, q], bl[:, q]), (am[:, q], bm[:, q]), (am[:, q], bh[:, q]), (ah[:, q], bm[:, q]), (ah[:, q], bh[:, q])])),
         "bf16x3 (opt-in k_dw)": run(lambda q: mfma_acc([(am[:, q], bh[:, q]), (ah[:, q], bm[:, q]), (ah[:, q], bh[:, q])])),
         "fp16x2, 3 products": run(lambda q: mfma_acc([(fl[:, q], gh[:, q]), (fh[:, q], gl[:, q]), (fh[:, q], gh[:, q])])).astype(np.float64) * inv,
         "fp16x2, 4 products": run(lambda q: mfma_acc([(fl[:, q], gl[:, q]), (fl[:, q], gh[:, q]), (fh[:, q], gl[:, q]), (fh[:, q], gh[:, q])])).astype(np.float64) * inv}
    ref_r = (dz_cols[:, :R].astype(np.float64) * x_cols[:, :R].astype(np.float64)).sum(1)
    for k, v in e.items():
        err = np.abs(np.asarray(v, np.float64) - ref_r)[keep] / denom[keep]
        old = res.setdefault(name, {}).get(k, (0.0, 0.0))
        res[name][k] = (max(old[0], float(np.sqrt(np.mean(err ** 2)))), max(old[1], float(err.max())))


def study(name, a, b, sa, sb, res):
    """a, b: (samples, K) fp32 operand rows of the sampled dot products; sa, sb: the per-TENSOR fp16 scales of the tensors they were drawn from."""
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    denom = (np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64))).sum(1)
    keep = denom > 0
    e = {"fp32 fmaf chain": fmaf_chain(a, b)}
    ah, am, al = split_bf3(a); bh, bm, bl = split_bf3(b)
    e["bf16x6 (shipped)"] = mfma_acc([(al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)])
    e["bf16x3 (opt-in k_dw)"] = mfma_acc([(am, bh), (ah, bm), (ah, bh)])
    fh, fl = split_f16(a, sa); gh, gl = split_f16(b, sb)
    inv = np.float64(1.0 / (sa * sb))
    e["fp16x2, 3 products"] = (mfma_acc([(fl, gh), (fh, gl), (fh, gh)]).astype(np.float64) * inv)
    e["fp16x2, 4 products"] = (mfma_acc([(fl, gl), (fl, gh), (fh, gl), (fh, gh)]).astype(np.float64) * inv)
    for k, v in e.items():
        err = np.abs(np.asarray(v, np.float64) - ref)[keep] / denom[keep]
        old = res.setdefault(name, {}).get(k, (0.0, 0.0))                      # over the layers of a net: the worst layer's figures
        res[name][k] = (max(old[0], float(np.sqrt(np.mean(err ** 2)))), max(old[1], float(err.max())))


def main(path):
    import torch
    from oracle import atlas_oracle as O
    import aiod_amd
    st = dict(np.load(path))
    cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
    v = O.synthetic_video(768, 432, 80, seed=0, flow="field")
    rng = np.random.default_rng(0)
    S = 3000
    res = {}
    for tag, it in (("it0", 0), ("it5000", 5000), ("it10001", 10001)):
        m, a = O.build_single_atlas_models(cfg, seed=0)
        for mdl, key in ((m, tag + "_net%d" % aiod_amd.NET_MAPPING1), (a, tag + "_net%d" % aiod_amd.NET_ATLAS)):
            flat, off = st[key], 0
            with torch.no_grad():
                for p in mdl.parameters():
                    p.copy_(torch.from_numpy(flat[off:off + p.numel()].reshape(p.shape))); off += p.numel()
        caps = {}
        hooks = []
        for nm, mdl in (("mapping", m), ("atlas", a)):
            for li, lin in enumerate(mdl.hidden):
                if lin.weight.shape[1] < 256 or lin.weight.shape[0] < 256:
                    continue                                   # the 256-wide contractions are what runs on the split products
                def fwd(mod, inp, out, key=(nm, li)):
                    caps.setdefault(key, {}).setdefault("X", []).append(inp[0].detach()[:, :256].numpy().copy())
                    out.register_hook(lambda g, key=key: caps[key].setdefault("dZ", []).append(g.detach().numpy().copy()))
                hooks.append(lin.register_forward_hook(fwd))
        tr = O.SingleAtlasTrainer(cfg, v, mapping=m, atlas=a)
        inds = torch.from_numpy(rng.integers(0, v.F * v.resx * v.resy, cfg["samples_batch"]))
        tr.loss_and_grads(min(it, 10000), inds)
        for h in hooks:
            h.remove()
        for (nm, li), c in sorted(caps.items()):
            X = np.concatenate(c["X"]); dZ = np.concatenate(c["dZ"][::-1])          # backward hooks fire in reverse call order
            W = dict((("mapping", m), ("atlas", a)))[nm].hidden[li].weight.detach().numpy()[:, :256]
            sx, sz, sw = pow2_scale(X), pow2_scale(dZ), pow2_scale(W)
            r = rng.integers(0, X.shape[0], S); o = rng.integers(0, 256, S); i = rng.integers(0, 256, S)
            study("%s forward   (W x X, K=256)" % nm, W[o], X[r], sw, sx, res.setdefault(tag, {}))
            study("%s backward  (W^T x dZ, K=256)" % nm, W[:, i].T.copy(), dZ[r], sw, sz, res.setdefault(tag, {}))
            r0 = rng.integers(0, X.shape[0] - 352, S // 4)
            aa = np.stack([dZ[r0[k]:r0[k] + 352, o[k]] for k in range(S // 4)]); bb = np.stack([X[r0[k]:r0[k] + 352, i[k]] for k in range(S // 4)])
            study("%s weight gradient (dZ^T x X, K=352 rows)" % nm, aa, bb, sz, sx, res.setdefault(tag, {}))
            so, si = rng.integers(0, 256, 48), rng.integers(0, 256, 48)
            study_full_dw("%s weight gradient, the whole batch (all rows, 352-row partial blocks summed in fp32)" % nm, dZ[:, so].T.copy(), X[:, si].T.copy(), sz, sx, res.setdefault(tag, {}))
            rowmax = np.abs(dZ).max(axis=1)
            print(tag, nm, "layer", li, "row maxima of dZ: median / max = 2^%.1f, 1st percentile / max = 2^%.1f" % (np.log2(np.median(rowmax) / rowmax.max()), np.log2(max(np.percentile(rowmax, 1), 1e-300) / rowmax.max())))
            print(tag, nm, "layer", li, "max|X| %.3g  max|dZ| %.3g  max|W| %.3g   smallest non-zero |dZ| / max|dZ| = 2^%.0f"
                  % (np.abs(X).max(), np.abs(dZ).max(), np.abs(W).max(), np.log2(np.abs(dZ[dZ != 0]).min() / np.abs(dZ).max())), flush=True)
    # pooled per (state, kind): worst layer
    for tag in res:
        print("==", tag)
        for kind, by in res[tag].items():
            print("  ", kind)
            for k, (rms, mx) in by.items():
                print("      %-24s rms %.2e   worst %.2e      (x 2^-24 = %.2f / %.2f)" % (k, rms, mx, rms / 2 ** -24, mx / 2 ** -24))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        dump(sys.argv[2])
    else:
        main(sys.argv[1])
