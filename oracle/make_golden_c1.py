"""BASELINE configs[0] run through the REFERENCE's own modules: the acceptance metric on a BASELINE config.

For each seed: the seeded synthetic 80-frame 160x90 video (the sample mp4 is 640x360, --down 4), the
reference's `IMLP`s initialised under `torch.manual_seed(seed)`, the reference's `pre_train_mapping`
(100 x F steps, unwrap_utils.py:176-198), then `iters_num` = 1001 iterations of the loop body of
src/stage1_neural_atlas.py:153-231 driven with the reference's own loss functions and `torch.optim.Adam`
(same harness as oracle/make_golden.py), and finally the mean PSNR of the reconstructed frames
(evaluate.py:640-661,740-743, restated in oracle/atlas_oracle.py — evaluate.py itself needs cv2/skimage).

Every random draw comes from torch's global CPU generator in the reference's order (model init, pre-train rows
then columns per step, one `torch.randint(P, (N, 1))` per loop iteration), so the GPU test can replay the very
same draws from the seed alone (tests/test_gpu_c1.py) — the fixture only stores the results:

    tests/golden/c1_reference.npz   per seed: PSNR after the pre-train, final PSNR (mean + per frame), the six
                                    loss terms every 100 iterations, wall-clock of the CPU run (5 threads)
    tests/golden/c1_reference_rerun.npz
                                    the same three seeds with 3 threads (`--seeds k --threads 3 --out ...` per seed,
                                    merged): the reference against ITSELF.  Only the summation order inside its GEMMs
                                    changes, and the final PSNR moves by up to 0.51 dB on a seed (25.013 -> 25.523 on
                                    seed 0; means over the seeds 25.525 vs 25.647 dB), single frames by up to 2.5 dB:
                                    the reproducibility floor tests/test_gpu_c1.py builds its tolerances on

    tests/golden/c1_reference_fp64_seed2.npz (and _seed0, _seed1)
                                    `--double`: the same modules, weights and draws in fp64 (seed 2: 25.30 dB, above the
                                    four fp32 runs of that seed, 24.93 .. 25.14): where exact arithmetic lands
    tests/golden/c1_reference_more.npz
                                    round 3: further seeds, one reference run each (`--seeds k --threads 3 --out ...` per seed, then
                                    `--merge`): with the three two-arm seeds the mean over seeds has a standard error <= ~0.1 dB,
                                    which is what "PSNR within 0.1 dB" needs to be testable at all

Build container only (imports /root/reference read-only; ~17 min of CPU per seed):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_c1.py [--seeds 0 1 2] [--threads 6]
"""
import argparse
import contextlib
import io
import os
import sys
import time
import types

import numpy as np
import torch

REF = os.environ.get("AF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
for _name in ("cv2", "imageio"):
    sys.modules.setdefault(_name, types.ModuleType(_name))

from src.models.stage_1.implicit_neural_networks import IMLP                       # noqa: E402
from src.models.stage_1.unwrap_utils import get_tuples, pre_train_mapping           # noqa: E402

from oracle import atlas_oracle as O                                                # noqa: E402
from oracle.make_golden import ref_iteration                                        # noqa: E402

RESX, RESY, NF = 160, 90, 80           # configs[0]: Winter_Scenes_in_Holland 640x360 / 4, 80 frames
ITERS = 1001                           # BASELINE configs[0] iters_num=1000 -> evaluate at iteration 1000
PRETRAIN_ITERS = 100                   # config_flow_100.json:36
LOG_EVERY = 100


def shipped_config():
    import importlib.util
    spec = importlib.util.spec_from_file_location("af_atlasfit_cfg", os.path.join(ROOT, "all-in-one-deflicker_amd", "atlasfit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return dict(mod.REFERENCE_CONFIG)


def _grad64_step(models32, twins64, video64, i, jif, c, opt):
    """--grad64: the gradient of iteration i from an fp64 twin of the CURRENT fp32 weights (forward and backward in fp64 on the same batch),
    cast to fp32 and handed to the fp32 torch.optim.Adam of the fp32 weights.  Everything the run keeps between iterations (weights, Adam
    moments, draws) is what the fp32 reference run keeps; only the round-off of torch's fp32 forward / backward inside ONE iteration is gone.
    If the reference's fp32 gradient noise is what costs it ~0.1 dB against the HIP path, this arm must land where the
    HIP path lands."""
    for m32, m64 in zip(models32, twins64):
        with torch.no_grad():
            for p32, p64 in zip(m32.parameters(), m64.parameters()):
                p64.copy_(p32.double()); p64.grad = None
    torch.set_default_dtype(torch.float64)
    try:
        loss, terms = ref_iteration(i, jif, video64, twins64[0], twins64[1], c)
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    opt.zero_grad()
    for m32, m64 in zip(models32, twins64):
        for p32, p64 in zip(m32.parameters(), m64.parameters()):
            p32.grad = p64.grad.float()
    opt.step()
    return terms


def _save_partial(path, seed, flow, curve, psnr_at, t_pre, t_loop, psnr_pre):
    """configs[1] runs take hours: what the run has produced so far is written after every logged iteration, so that an
    interrupted run still leaves the loss curve and the intermediate PSNRs behind (tests/test_gpu_c2.py accepts a partial file)."""
    tmp = path + ".tmp.npz"
    np.savez_compressed(tmp, seed=seed, flow_kind=flow, resx=RESX, resy=RESY, nframes=NF, iters=ITERS, log_every=LOG_EVERY,
                        curve=np.array(curve, np.float64), psnr_at_iter=np.array([k for k, _ in psnr_at], np.int64),
                        psnr_at=np.array([v for _, v in psnr_at], np.float64), psnr_pre=psnr_pre,
                        cpu_seconds=np.array([t_pre, t_loop]), threads=torch.get_num_threads(), complete=False)
    os.replace(tmp, path)


def run_seed(seed, c, double=False, flow="constant", grad64=False, psnr_iters=(), partial=None):
    video = O.synthetic_video(RESX, RESY, NF, seed=seed, flow=flow)
    torch.manual_seed(seed)
    # stage1_neural_atlas.py:112-128 (mapping first, then atlas)
    rm = IMLP(input_dim=3, output_dim=2, hidden_dim=256, use_positional=False, positional_dim=4, num_layers=6, skip_layers=[], verbose=False)
    ra = IMLP(input_dim=2, output_dim=3, hidden_dim=256, use_positional=True, positional_dim=10, num_layers=8, skip_layers=[4, 7], verbose=False)
    if double:     # --double: the same fp32 initial weights, draws and video, every operation from here on in fp64 (a diagnostic, not a fixture)
        rm.double(); ra.double()
        torch.set_default_dtype(torch.float64)
        for k, v in list(vars(video).items()):
            if torch.is_tensor(v) and v.dtype == torch.float32:
                setattr(video, k, v.double())
    opt = torch.optim.Adam([{"params": list(rm.parameters())}, {"params": list(ra.parameters())}], lr=0.0001)
    if grad64:
        import copy
        twins = [copy.deepcopy(rm).double(), copy.deepcopy(ra).double()]
        video64 = copy.copy(video)
        for k, v in list(vars(video).items()):
            if torch.is_tensor(v) and v.dtype == torch.float32:
                setattr(video64, k, v.double())
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        pre_train_mapping(rm, NF, c["uv_mapping_scale"], resx=RESX, resy=RESY, larger_dim=video.larger_dim, device="cpu",
                          pretrain_iters=PRETRAIN_ITERS)
    t_pre = time.time() - t0
    psnr_pre, _ = O.mean_psnr(rm, ra, video)
    jif_all = get_tuples(NF, video.video_frames)
    N = c["samples_batch"]
    curve, psnr_at, t_eval = [], [], 0.0
    t0 = time.time()
    for i in range(ITERS):
        if i in psnr_iters:     # PSNR of the state BEFORE iteration i (i.e. after i iterations), outside the draw stream: rendering draws nothing
            te = time.time()
            psnr_at.append((i, O.mean_psnr(rm, ra, video)[0]))
            t_eval += time.time() - te
            print("seed %d: PSNR %.4f dB after %d iterations" % (seed, psnr_at[-1][1], i), flush=True)
        inds = torch.randint(jif_all.shape[1], (np.int64(N * 1.0), 1))
        if grad64:
            terms = _grad64_step((rm, ra), twins, video64, i, jif_all[:, inds], c, opt)
        else:
            loss, terms = ref_iteration(i, jif_all[:, inds], video, rm, ra, c)
            opt.zero_grad(); loss.backward(); opt.step()
        if i % LOG_EVERY == 0:
            curve.append(terms)
            print("seed %d iter %4d  total %.4f  rgb %.5f  (%.0f s)" % (seed, i, terms[5], terms[0], time.time() - t0), flush=True)
            if partial:
                _save_partial(partial, seed, flow, curve, psnr_at, t_pre, time.time() - t0 - t_eval, psnr_pre)
    t_loop = time.time() - t0 - t_eval
    psnr, per = O.mean_psnr(rm, ra, video)
    print("seed %d: PSNR %.4f dB after the pre-train -> %.4f dB after %d iterations (pre-train %.0f s, loop %.0f s)" % (seed, psnr_pre, psnr, ITERS, t_pre, t_loop), flush=True)
    return dict(psnr_pre=psnr_pre, psnr=psnr, per_frame=np.array(per), curve=np.array(curve, np.float64), t_pre=t_pre, t_loop=t_loop,
                psnr_at_iter=np.array([k for k, _ in psnr_at], np.int64), psnr_at=np.array([v for _, v in psnr_at], np.float64),
                video_checksum=float(video.video_frames.double().sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c1_reference.npz"))
    ap.add_argument("--double", action="store_true", help="run the schedule in fp64 from the fp32 initial weights (diagnostic: where exact arithmetic lands)")
    ap.add_argument("--merge", nargs="+", default=None, help="stack per-seed files written by earlier invocations into --out (thread counts and flow kinds are kept per seed)")
    ap.add_argument("--grad64", action="store_true", help="fp32 weights, Adam state and draws; the gradient of every loop iteration from an fp64 twin (diagnostic: is torch-fp32's gradient round-off what separates the reference from the HIP path?)")
    ap.add_argument("--flow", default="constant", choices=["constant", "field"], help="round 4: 'field' = oracle.atlas_oracle.synthetic_video(flow='field'), a per-pixel, per-frame flow field with holed masks")
    ap.add_argument("--resx", type=int, default=RESX)
    ap.add_argument("--resy", type=int, default=RESY)
    ap.add_argument("--nframes", type=int, default=NF)
    ap.add_argument("--iters", type=int, default=ITERS, help="round 5: 10001 with --resx 768 --resy 432 = BASELINE configs[1], the shipped iters_num (config_flow_100.json:6)")
    ap.add_argument("--log-every", type=int, default=LOG_EVERY)
    ap.add_argument("--psnr-at", type=int, nargs="*", default=[], help="also render + PSNR after this many loop iterations (e.g. 5000 = where global rigidity stops, config_flow_100.json:44)")
    ap.add_argument("--partial", default=None, help="write the curve / PSNRs so far to this file after every logged iteration")
    args = ap.parse_args()
    globals().update(RESX=args.resx, RESY=args.resy, NF=args.nframes, ITERS=args.iters, LOG_EVERY=args.log_every)
    if args.merge:
        parts = [dict(np.load(f)) for f in args.merge]
        parts.sort(key=lambda d: int(d["seeds"][0]))
        assert len({int(d["iters"]) for d in parts}) == 1
        for d in parts:         # files written before round 4 carry one thread count and no flow kind
            n = len(d["seeds"])
            d["threads_per_seed"] = d.get("threads_per_seed", np.full(n, int(d["threads"])))
            d["flow_kind"] = d.get("flow_kind", np.array(["constant"] * n))
        out = {k: parts[0][k] for k in ("resx", "resy", "nframes", "iters", "pretrain_iters", "log_every")}
        out["threads"] = parts[0]["threads"] if len({int(d["threads"]) for d in parts}) == 1 else np.int64(-1)
        keys = ["seeds", "psnr_pre", "psnr", "psnr_per_frame", "curves", "cpu_seconds", "video_checksum", "threads_per_seed", "flow_kind"]
        if all("psnr_at" in d for d in parts):
            keys += ["psnr_at"]; out["psnr_at_iter"] = parts[0]["psnr_at_iter"]
        for k in keys:
            out[k] = np.concatenate([d[k] for d in parts], axis=0)
        np.savez_compressed(args.out, **out)
        print("merged", [int(x) for x in out["seeds"]], "->", args.out, "PSNR", np.array2string(out["psnr"], precision=3))
        return
    if args.threads > 0:
        torch.set_num_threads(args.threads)
    c = shipped_config()
    res = [run_seed(s, c, args.double, args.flow, args.grad64, tuple(args.psnr_at), args.partial) for s in args.seeds]
    np.savez_compressed(
        args.out, seeds=np.array(args.seeds), resx=RESX, resy=RESY, nframes=NF, iters=ITERS, pretrain_iters=PRETRAIN_ITERS, log_every=LOG_EVERY,
        psnr_pre=np.array([r["psnr_pre"] for r in res]), psnr=np.array([r["psnr"] for r in res]),
        psnr_per_frame=np.stack([r["per_frame"] for r in res]), curves=np.stack([r["curve"] for r in res]),
        cpu_seconds=np.array([[r["t_pre"], r["t_loop"]] for r in res]), threads=torch.get_num_threads(),
        video_checksum=np.array([r["video_checksum"] for r in res]),
        threads_per_seed=np.full(len(res), torch.get_num_threads()), flow_kind=np.array([args.flow] * len(res)),
        psnr_at_iter=res[0]["psnr_at_iter"], psnr_at=np.stack([r["psnr_at"] for r in res]),
    )
    print("written", args.out, "PSNR", [r["psnr"] for r in res])


if __name__ == "__main__":
    main()
