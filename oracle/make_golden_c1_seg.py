"""BASELINE configs[4]'s acceptance metric ("fg+bg dual atlas with alpha MLP, PSNR parity vs reference") on a
complete schedule: the REFERENCE's own seg modules driven end to end, the fg/bg twin of oracle/make_golden_c1.py.

For each seed: the seeded synthetic 80-frame 160x90 video (configs[0]'s size) with the moving soft-edged disc as
foreground mask (oracle/atlas_oracle.py:synthetic_seg_video), the reference's four `IMLP`s initialised under
`torch.manual_seed(seed)` in the reference's construction order (stage1_neural_atlas_seg.py:127-161), the
reference's `pre_train_mapping` on mapping1 and then mapping2 (:173-179, 100 x F steps each), then 1001 iterations
of the loop body of src/stage1_neural_atlas_seg.py:193-311 driven with the reference's own loss functions and
`torch.optim.Adam` (param-group order of :163-167; the harness is oracle/make_golden_seg.py:ref_seg_iteration),
and finally the mean PSNR of the alpha-blended reconstruction (evaluate.py:302-337, restated in
oracle/atlas_oracle.py:render_frame_seg - evaluate.py itself needs cv2/skimage).

The hyper-parameters are the shipped config_flow_100.json (stop_bootstrapping_iteration 10000, stop_global_rigidity
5000: neither switches within 1001 iterations, exactly as `test.py`-style runs of the reference at iters_num 1000).

Every random draw comes from torch's global CPU generator in the reference's order (four model inits, per pre-train
step the row then the column draw for mapping1's 8000 steps then mapping2's, one `torch.randint(P, (N, 1))` per
loop iteration), so tests/test_gpu_c1_seg.py replays the same draws from the seed alone; the fixture stores results:

    tests/golden/c1_seg_reference.npz   per seed and per arm (thread count): PSNR after the pre-trains, final PSNR
                                        (mean + per frame), the 12 loss terms every 100 iterations, CPU wall-clock

One invocation = one seed at one thread count, written to --out; `--merge a.npz b.npz ... --out fixture.npz` stacks them.
Build container only (imports /root/reference read-only; ~40-60 min of CPU per run):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_c1_seg.py --seed 0 --threads 3 --out gpurun_out/c1seg/s0_t3.npz
"""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

RESX, RESY, NF = 160, 90, 80           # configs[0]'s geometry (640x360 / 4, 80 frames)
ITERS = 1001
PRETRAIN_ITERS = 100                   # config_flow_100.json "pretrain_iter_number"
LOG_EVERY = 100


def run(seed, threads, double, flow="constant"):
    from oracle.make_golden_c1 import shipped_config                                # also puts /root/reference on sys.path
    from oracle.make_golden_seg import ref_models, ref_seg_iteration
    from oracle import atlas_oracle as O
    from src.models.stage_1.unwrap_utils import get_tuples, pre_train_mapping
    if threads > 0:
        torch.set_num_threads(threads)
    c = shipped_config()
    video = O.synthetic_seg_video(RESX, RESY, NF, seed=seed, flow=flow)
    m1, m2, at, al = ref_models(seed)           # torch.manual_seed(seed); mapping1, mapping2, atlas, alpha
    if double:
        for m in (m1, m2, at, al):
            m.double()
        torch.set_default_dtype(torch.float64)
        for k, v in list(vars(video).items()):
            if torch.is_tensor(v) and v.dtype == torch.float32:
                setattr(video, k, v.double())
    opt = torch.optim.Adam([{"params": list(m1.parameters())}, {"params": list(m2.parameters())},
                            {"params": list(al.parameters())}, {"params": list(at.parameters())}], lr=0.0001)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        for m in (m1, m2):
            pre_train_mapping(m, NF, c["uv_mapping_scale"], resx=RESX, resy=RESY, larger_dim=video.larger_dim, device="cpu",
                              pretrain_iters=PRETRAIN_ITERS)
    t_pre = time.time() - t0
    psnr_pre, _ = O.mean_psnr_seg(m1, m2, at, al, video)
    print("seed %d: pre-trains done in %.0f s, PSNR %.4f dB" % (seed, t_pre, psnr_pre), flush=True)
    jif_all = get_tuples(NF, video.video_frames)
    N = c["samples_batch"]
    curve = []
    t0 = time.time()
    for i in range(ITERS):
        inds = torch.randint(jif_all.shape[1], (np.int64(N * 1.0), 1))
        loss, terms = ref_seg_iteration(i, jif_all[:, inds], video, m1, m2, at, al, c)
        opt.zero_grad(); loss.backward(); opt.step()
        if i % LOG_EVERY == 0:
            curve.append(terms)
            print("seed %d iter %4d  total %.4f  rgb %.5f  (%.0f s)" % (seed, i, terms[11], terms[0], time.time() - t0), flush=True)
    t_loop = time.time() - t0
    psnr, per = O.mean_psnr_seg(m1, m2, at, al, video)
    print("seed %d: PSNR %.4f dB after the pre-trains -> %.4f dB after %d iterations (pre-train %.0f s, loop %.0f s)"
          % (seed, psnr_pre, psnr, ITERS, t_pre, t_loop), flush=True)
    return dict(seed=seed, threads=torch.get_num_threads(), double=int(double), psnr_pre=psnr_pre, psnr=psnr, per_frame=np.array(per),
                curve=np.array(curve, np.float64), cpu_seconds=np.array([t_pre, t_loop]),
                video_checksum=float(video.video_frames.double().sum()), mask_checksum=float(video.mask_frames.double().sum()), flow_kind=flow)


def merge(paths, out):
    runs = [dict(np.load(p)) for p in paths]
    runs.sort(key=lambda r: (int(r["double"]), int(r["seed"]), int(r["threads"])))
    np.savez_compressed(
        out, resx=RESX, resy=RESY, nframes=NF, iters=ITERS, pretrain_iters=PRETRAIN_ITERS, log_every=LOG_EVERY,
        seeds=np.array([int(r["seed"]) for r in runs]), threads=np.array([int(r["threads"]) for r in runs]),
        double=np.array([int(r["double"]) for r in runs]),
        psnr_pre=np.array([float(r["psnr_pre"]) for r in runs]), psnr=np.array([float(r["psnr"]) for r in runs]),
        psnr_per_frame=np.stack([r["per_frame"] for r in runs]), curves=np.stack([r["curve"] for r in runs]),
        cpu_seconds=np.stack([r["cpu_seconds"] for r in runs]),
        video_checksum=np.array([float(r["video_checksum"]) for r in runs]), mask_checksum=np.array([float(r["mask_checksum"]) for r in runs]),
        flow_kind=np.array([str(r.get("flow_kind", "constant")) for r in runs]))
    print("merged %d runs into %s" % (len(runs), out))
    for r in runs:
        print("  seed %d threads %d %s: PSNR %.4f dB" % (int(r["seed"]), int(r["threads"]), "fp64" if int(r["double"]) else "fp32", float(r["psnr"])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--double", action="store_true", help="the same modules, initial weights and draws in fp64 (diagnostic arm)")
    ap.add_argument("--merge", nargs="+", default=None)
    ap.add_argument("--flow", default="constant", choices=["constant", "field"], help="round 4: 'field' = a per-pixel, per-frame flow field with holed masks")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    if args.merge:
        return merge(args.merge, args.out)
    r = run(args.seed, args.threads, args.double, args.flow)
    np.savez_compressed(args.out, **r)
    print("written", args.out)


if __name__ == "__main__":
    main()
