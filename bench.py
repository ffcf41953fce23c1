#!/usr/bin/env python
"""bench.py — atlas-fit sampled points/s (stage 1 main loop) on N GPUs of one node.

A "step" is one iteration of the optimisation loop (src/stage1_neural_atlas.py:151-231 of the reference)
over one batch of samples_batch = 10 000 sampled (x,y,t) points of a synthetic 80-frame 768x432 video
(BASELINE.json configs[1]).  Inputs are resident in HBM before the timed region.  The K timed steps are
centred on iteration 5000 so that half of them carry the global-rigidity rows, like the 10 001-iteration
schedule the metric is quoted on.  N > 1: one independent video per GPU (weak scaling), RCCL barrier only.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0         # same guide: dense bf16 MFMA peak (2.5 PF; 2495 TF measured)
BF16X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0   # an fp32-faithful product = six bf16 partial products (bfsplit.h): 416.7 TF of fp32-equivalent work
HBM_PEAK_GBS = 8000.0                  # same guide: HBM3E 8 TB/s spec (6.3 TB/s measured for a streaming copy)
# algorithmic HBM bytes of k_dw per 32-row tile of each net (every operand tile [features][32 rows] read once):
# mapping1 4x64 + 36 + 36 KB, atlas 6x64 + 40 + 40 + 40 + 36 + 12, mapping2 2x64 + 36 + 36, alpha 6x64 + 40 + 36
DW_TILE_BYTES = {"map1": 328 * 1024, "atlas": 512 * 1024, "map2": 200 * 1024, "alpha": 460 * 1024}
METRIC = "atlas-fit sampled points/sec (stage1, 10k iters) @1/2/4/8 GPU; PSNR vs ref"     # BASELINE.json "metric"
# launch classes of af_get_timing -> kernel names as rocprofv3 prints them
F16X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0    # fp16 MFMA runs at the bf16 rate (same guide); an fp32-faithful product = three fp16 partial products (mlphf.hip): 833.3 TF
# The arithmetic modes are the LIBRARY's (af_get_modes; include/atlasfit.h af_set_mlp_mode / af_set_dw_mode): 3 = f16x3 chains (two-term fp16 split with a scale
# per row, three products: the default since round 6), 1 = bf16x6, 2 = bf16x6 forward + three-product bf16 backward (experiment), 0 = fp32 MFMA;
# k_dw: 1 = bf16x6 (default), 2 = bf16x3 (opt-in, narrower than fp32), 0 = fp32 MFMA.  AF_EXPERIMENT=1 AF_MLP_MODE=<m> / AF_DW_MODE=<m> select another (atlasfit.py).
MLP_MODE, DW_MODE = 3, 1          # set from the handle in main() (configure_modes)
MLP_BF, DW_BF, DW_PRODUCTS, KERNEL_OF_CLASS = True, True, 6, {}


def configure_modes(mlp_mode, dw_mode):
    """Kernel names as rocprofv3 prints them and the peak of each arithmetic, for the modes the handle runs in."""
    global MLP_MODE, DW_MODE, MLP_BF, DW_BF, DW_PRODUCTS, KERNEL_OF_CLASS
    MLP_MODE, DW_MODE = int(mlp_mode), int(dw_mode)
    MLP_BF, DW_BF = MLP_MODE != 0, DW_MODE != 0
    DW_PRODUCTS = {0: 1, 1: 6, 2: 3}[DW_MODE]
    fwd = {0: "k_mlp_fwd_multi<true>", 3: "k_mlp_fwd_multi_hf<true>"}.get(MLP_MODE, "k_mlp_fwd_multi_bf<true>")
    bwd = {0: "k_mlp_bwd_multi", 2: "k_mlp_bwd_multi_bf3", 3: "k_mlp_bwd_multi_hf"}.get(MLP_MODE, "k_mlp_bwd_multi_bf")
    KERNEL_OF_CLASS = {"fwd_1": fwd, "fwd_2": fwd, "bwd_1": bwd, "bwd_2": bwd, "dw": ("k_dw_bf<%d>" % DW_PRODUCTS) if DW_BF else "k_dw",
                       "prep": "k_prep", "loss": "k_loss", "adam": "k_adam<true>"}


def chain_peak_tflops():
    """Dense matrix-pipe peak of fp32-faithful work for the chains' arithmetic: never the raw 16-bit peak."""
    return {0: FP32_MFMA_PEAK_TFLOPS, 3: F16X3_PEAK_TFLOPS}.get(MLP_MODE, BF16X6_PEAK_TFLOPS)


configure_modes(3, 1)


def _remap_zero(img, mapx, mapy):
    """Bilinear sample of img (H, W, C) at (mapx, mapy) with a constant-0 border — what cv2.remap does in unwrap_utils.py:22."""
    H, W = img.shape[:2]
    x0, y0 = mapx.floor(), mapy.floor()
    fx, fy = (mapx - x0)[..., None], (mapy - y0)[..., None]

    def tap(yy, xx):
        ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H))[..., None]
        return img[yy.clamp(0, H - 1).long(), xx.clamp(0, W - 1).long()] * ok

    return tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) + tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy


def _consistent(f12, f21, xx, yy):
    """unwrap_utils.py:10-23,151-159: a flow vector is valid where ||f12 + warp(f21, f12)|| < 1."""
    d = f12 + _remap_zero(f21, f12[..., 0] + xx, f12[..., 1] + yy)
    return ((d[..., 0] ** 2 + d[..., 1] ** 2).sqrt() < 1.0).float()


def synth_video_device(resx, resy, nframes, seed, device, flow="constant", flicker=True):
    """Seeded synthetic flickering video generated directly in HBM (same construction as SURVEY.md §8d:
    translating smooth texture, per-frame gain/gamma flicker, exact flows, reference consistency rule).

    flow="field": the texture moves by a different similarity transform every frame (rotation, zoom, translation about the frame
    centre), so the flow differs at every pixel of every frame and nothing is dyadic; each field carries a sub-pixel ripple and a
    Gaussian error bump a few px high, so the reference's consistency rule leaves holes in the masks, not only border strips
    (the constant (1.5, 0.5) field cannot see a wrong per-pixel flow gather or a mis-rounded advected coordinate)."""
    import torch
    if flow == "field":
        return _synth_video_field_device(resx, resy, nframes, seed, device, flicker)
    assert flow == "constant" and flicker, flow
    g = torch.Generator(device="cpu").manual_seed(seed)
    nw = 12
    scale = 2 * 3.141592653589793 * 6 / max(resx, resy)
    kx = ((torch.rand(nw, 3, generator=g) * 2 - 1) * scale).to(device)
    ky = ((torch.rand(nw, 3, generator=g) * 2 - 1) * scale).to(device)
    ph = (torch.rand(nw, 3, generator=g) * 2 * 3.141592653589793).to(device)
    amp = (torch.rand(nw, 3, generator=g) * 0.7 + 0.3).to(device)
    gain = torch.rand(nframes, generator=g) * 0.4 + 0.8
    gamma = torch.rand(nframes, generator=g) * 0.2 + 0.9
    vx, vy = 1.5, 0.5
    yy, xx = torch.meshgrid(torch.arange(resy, device=device, dtype=torch.float32),
                            torch.arange(resx, device=device, dtype=torch.float32), indexing="ij")
    frames = torch.empty(resy, resx, 3, nframes, device=device)
    for f in range(nframes):
        xs, ys = (xx - f * vx)[..., None, None], (yy - f * vy)[..., None, None]
        tex = (amp.T[None, None] * torch.sin(kx.T[None, None] * xs + ky.T[None, None] * ys + ph.T[None, None])).sum(-1)
        tex = 0.5 + 0.5 * tex / amp.sum(0)
        frames[:, :, :, f] = (float(gain[f]) * tex.clamp(1e-3, 1.0) ** float(gamma[f])).clamp(0.0, 1.0)
    flows = torch.zeros(resy, resx, 2, nframes, device=device)
    flows_rev = torch.zeros_like(flows)
    flows[:, :, 0, :-1] = vx; flows[:, :, 1, :-1] = vy
    flows_rev[:, :, 0, 1:] = -vx; flows_rev[:, :, 1, 1:] = -vy
    # consistency rule ||f12 + warp(f21)|| < 1 with zero border: invalid where the flow leaves the frame
    inb_f = ((xx + vx >= 0) & (xx + vx <= resx - 1) & (yy + vy >= 0) & (yy + vy <= resy - 1)).float()
    inb_b = ((xx - vx >= 0) & (xx - vx <= resx - 1) & (yy - vy >= 0) & (yy - vy <= resy - 1)).float()
    mask = torch.zeros(resy, resx, nframes, device=device)
    mask_rev = torch.zeros_like(mask)
    mask[:, :, :-1] = inb_f[..., None]
    mask_rev[:, :, 1:] = inb_b[..., None]
    return frames, flows, flows_rev, mask, mask_rev


def _synth_video_field_device(resx, resy, nframes, seed, device, flicker=True):
    import math
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    nw = 12
    scale = 2 * math.pi * 6 / max(resx, resy)
    kx = ((torch.rand(nw, 3, generator=g) * 2 - 1) * scale).to(device)
    ky = ((torch.rand(nw, 3, generator=g) * 2 - 1) * scale).to(device)
    ph = (torch.rand(nw, 3, generator=g) * 2 * math.pi).to(device)
    amp = (torch.rand(nw, 3, generator=g) * 0.7 + 0.3).to(device)
    gain = torch.rand(nframes, generator=g) * 0.4 + 0.8
    gamma = torch.rand(nframes, generator=g) * 0.2 + 0.9
    if not flicker:                                            # tests of the generator itself: frame f+1 warped by the flow == frame f
        gain, gamma = torch.ones(nframes), torch.ones(nframes)
    n, m = nframes - 1, float(min(resx, resy))

    def uni(lo, hi, *shape):
        return (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).tolist()

    theta, zoom, tx, ty = uni(-0.008, 0.008, n), uni(0.994, 1.006, n), uni(-2.0, 2.0, n), uni(-1.2, 1.2, n)
    bcx, bcy = uni(0.2 * resx, 0.8 * resx, n, 2), uni(0.2 * resy, 0.8 * resy, n, 2)
    bax, bay, br = uni(-3.0, 3.0, n, 2), uni(1.5, 3.0, n, 2), uni(0.08 * m, 0.16 * m, n, 2)
    rk, rp, ra = uni(0.05, 0.4, n, 2), uni(0.0, 6.28, n, 2), uni(0.02, 0.08, n, 2)
    yy, xx = torch.meshgrid(torch.arange(resy, device=device, dtype=torch.float32),
                            torch.arange(resx, device=device, dtype=torch.float32), indexing="ij")
    frames = torch.empty(resy, resx, 3, nframes, device=device)
    flows = torch.zeros(resy, resx, 2, nframes, device=device)
    flows_rev = torch.zeros_like(flows)
    mask = torch.zeros(resy, resx, nframes, device=device)
    mask_rev = torch.zeros_like(mask)
    cx, cy = 0.5 * (resx - 1), 0.5 * (resy - 1)
    inv = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)                        # frame pixel -> texture coordinate (2x3, row-major); frame 0 = identity
    for f in range(nframes):
        xs = (inv[0] * xx + inv[1] * yy + inv[2])[..., None, None]
        ys = (inv[3] * xx + inv[4] * yy + inv[5])[..., None, None]
        tex = (amp.T[None, None] * torch.sin(kx.T[None, None] * xs + ky.T[None, None] * ys + ph.T[None, None])).sum(-1)
        tex = 0.5 + 0.5 * tex / amp.sum(0)
        frames[:, :, :, f] = (float(gain[f]) * tex.clamp(1e-3, 1.0) ** float(gamma[f])).clamp(0.0, 1.0)
        if f == n:
            break
        co, si = zoom[f] * math.cos(theta[f]), zoom[f] * math.sin(theta[f])
        a = (co, -si, cx - co * cx + si * cy + tx[f], si, co, cy - si * cx - co * cy + ty[f])            # A_f: frame f -> frame f+1
        det = a[0] * a[4] - a[1] * a[3]
        l = (a[4] / det, -a[1] / det, -a[3] / det, a[0] / det)
        ai = (l[0], l[1], -(l[0] * a[2] + l[1] * a[5]), l[2], l[3], -(l[2] * a[2] + l[3] * a[5]))       # A_f^-1
        pair = []
        for j, mm in enumerate((a, ai)):
            gb = torch.exp(-((xx - bcx[f][j]) ** 2 + (yy - bcy[f][j]) ** 2) / br[f][j] ** 2)
            rip = ra[f][j] * torch.sin(rk[f][j] * (xx + 0.7 * yy) + rp[f][j])
            pair.append(torch.stack(((mm[0] - 1.0) * xx + mm[1] * yy + mm[2] + bax[f][j] * gb + rip,
                                     mm[3] * xx + (mm[4] - 1.0) * yy + mm[5] + bay[f][j] * gb - rip), dim=-1))
        flows[:, :, :, f], flows_rev[:, :, :, f + 1] = pair[0], pair[1]
        mask[:, :, f] = _consistent(pair[0], pair[1], xx, yy)
        mask_rev[:, :, f + 1] = _consistent(pair[1], pair[0], xx, yy)
        inv = (inv[0] * ai[0] + inv[1] * ai[3], inv[0] * ai[1] + inv[1] * ai[4], inv[0] * ai[2] + inv[1] * ai[5] + inv[2],
               inv[3] * ai[0] + inv[4] * ai[3], inv[3] * ai[1] + inv[4] * ai[4], inv[3] * ai[2] + inv[4] * ai[5] + inv[5])
    return frames, flows, flows_rev, mask, mask_rev


def synth_fg_mask_device(resx, resy, nframes, seed, device):
    """Foreground mask of the two-layer workload: a soft-edged disc crossing the frame (fractional values, like the
    bilinearly resized masks the reference feeds, unwrap_utils.py:68-70)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed + 1000)
    cx0 = float(torch.rand(1, generator=g)) * 0.1 * resx + 0.3 * resx
    cy0 = float(torch.rand(1, generator=g)) * 0.2 * resy + 0.4 * resy
    r = 0.22 * min(resx, resy)
    yy, xx = torch.meshgrid(torch.arange(resy, device=device, dtype=torch.float32),
                            torch.arange(resx, device=device, dtype=torch.float32), indexing="ij")
    m = torch.empty(resy, resx, nframes, device=device)
    for f in range(nframes):
        d = ((xx - (cx0 + 0.8 * f)) ** 2 + (yy - (cy0 - 0.3 * f)) ** 2).sqrt()
        m[:, :, f] = ((r - d) / 2.0 + 0.5).clamp(0.0, 1.0)
    return m


def init_state_dicts(seed, two_layer=False):
    """torch default nn.Linear init in the reference's construction order (stage1_neural_atlas.py:112-128;
    stage1_neural_atlas_seg.py:127-161: mapping1, mapping2, atlas, alpha)."""
    import torch
    import aiod_amd
    torch.manual_seed(seed)
    sds = {}
    nets = (aiod_amd.NET_MAPPING1, aiod_amd.NET_MAPPING2, aiod_amd.NET_ATLAS, aiod_amd.NET_ALPHA) if two_layer else (aiod_amd.NET_MAPPING1, aiod_amd.NET_ATLAS)
    for net in nets:
        sd = {}
        for i, (o, k) in enumerate(aiod_amd.atlasfit.imlp_shapes(net)):
            lin = torch.nn.Linear(k, o)
            sd["hidden.%d.weight" % i] = lin.weight.detach()
            sd["hidden.%d.bias" % i] = lin.bias.detach()
        sds[net] = sd
    return sds


def cpu_baseline(resx, resy, nframes, seed, sds, video_dev, budget_s, two_layer=False):
    """The oracle (a PyTorch-CPU restatement of the reference loop) on the host cores, same workload,
    bounded sample: iterations until ~budget_s seconds (>= 4), half with the global-rigidity term."""
    import torch
    import aiod_amd
    from oracle import atlas_oracle as O
    cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
    frames, flows, flows_rev, mask, mask_rev = [t.cpu() for t in video_dev[:5]]
    if two_layer:
        v = O.SegVideo(frames, flows[..., None], flows_rev[..., None], mask[..., None], mask_rev[..., None], video_dev[5].cpu())
        models = O.build_seg_models(cfg, seed=0)
        for net, mdl in zip((aiod_amd.NET_MAPPING1, aiod_amd.NET_MAPPING2, aiod_amd.NET_ATLAS, aiod_amd.NET_ALPHA), models):
            mdl.load_state_dict(sds[net])
        tr = O.SegAtlasTrainer(cfg, v, models=models)
    else:
        v = O.Video(frames, flows[..., None], flows_rev[..., None], mask[..., None], mask_rev[..., None])
        m, a = O.build_single_atlas_models(cfg, seed=0)
        m.load_state_dict(sds[aiod_amd.NET_MAPPING1]); a.load_state_dict(sds[aiod_amd.NET_ATLAS])
        tr = O.SingleAtlasTrainer(cfg, v, mapping=m, atlas=a)
    N = cfg["samples_batch"]
    g = torch.Generator().manual_seed(seed)
    P = tr.jif_all.shape[1]
    tr.step(0, torch.randint(P, (N,), generator=g))          # warm-up (not timed)
    # pick the thread count that serves this box best (all cores is often NOT the fastest for these small GEMMs)
    best, best_dt = torch.get_num_threads(), None
    for nt in sorted({8, 16, 32, 64, torch.get_num_threads()}):
        if nt > torch.get_num_threads() and nt != best:
            continue
        torch.set_num_threads(nt)
        t1 = time.perf_counter(); tr.step(4000, torch.randint(P, (N,), generator=g)); d1 = time.perf_counter() - t1
        if best_dt is None or d1 < best_dt:
            best, best_dt = nt, d1
    torch.set_num_threads(best)
    t0 = time.perf_counter(); n = 0
    while n < 4 or (time.perf_counter() - t0 < budget_s and n < 200):
        it = 4000 if n % 2 == 0 else 6000                   # alternate: with / without global rigidity
        tr.step(it, torch.randint(P, (N,), generator=g)); n += 1
    dt = time.perf_counter() - t0
    out = {"value": N * n / dt, "unit": "sampled points/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d loop iterations (N=10000, alternating with/without the global-rigidity term) of the oracle restatement, %.1f s" % (n, dt)}
    try:      # the reference's OWN modules cannot run here (/root/reference does not travel): their figure from the build container, for the label only
        ref = json.load(open(os.path.join(ROOT, "profiles", "r2_cpu_reference.json")))["cpu_baseline"]
        out["reference_modules_build_container"] = {"value": ref["value"], "cores": ref["cores"], "source": "profiles/r2_cpu_reference.json (not this host)"}
    except Exception:
        pass
    return out


def timed_region(run, sync, dist=None, device=None):
    """Barrier + sync, run(), sync + barrier; returns the MAX over ranks of the elapsed seconds
    (the contract's timing rule).  `dist` is torch.distributed when world_size > 1."""
    import torch
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    timed_region.local_seconds = dt                      # this rank's own window (rank_stats gathers them)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def rank_stats(local_seconds, steps, device_name, dist=None, device=None):
    """Per-rank view of the timed window for the JSON line: every rank's ms/step and device name (gathered with the process group's
    own collectives, outside the timed region)."""
    import torch
    if dist is None:
        return {"ms_per_step_min": local_seconds / steps * 1e3, "ms_per_step_max": local_seconds / steps * 1e3, "ms_per_step_by_rank": [local_seconds / steps * 1e3],
                "devices": [device_name]}
    world = dist.get_world_size()
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    names = [None] * world
    try:
        dist.all_gather_object(names, device_name)
    except Exception as e:                               # the names are informational; the timing is not
        names = [device_name] + ["(all_gather_object failed: %r)" % (e,)] * (world - 1)
    ms = [float(x.item()) / steps * 1e3 for x in ts]
    return {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "ms_per_step_by_rank": ms, "devices": names}


def check_gpu_count(n_gpus):
    """SURVEY.md 7, last bullet: enumerate the visible devices and say what is there BEFORE any rank dies in set_device."""
    import torch
    seen = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_gpus > seen:
        names = [torch.cuda.get_device_name(i) for i in range(seen)]
        sys.exit("bench.py: --gpus %d but this process sees %d GPU(s) %s (HIP_VISIBLE_DEVICES=%s, ROCR_VISIBLE_DEVICES=%s): one rank per GPU, nothing launched"
                 % (n_gpus, seen, names, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
    return seen


def shard_for_rank(rank, world, n_videos=None):
    """Independent videos shard one per GPU: rank r owns videos r, r+world, ... (seeds double as video ids)."""
    n_videos = world if n_videos is None else n_videos
    return list(range(rank, n_videos, world))


def self_spawn(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher (no RANK in the environment): re-exec under torch.distributed.run,
    one rank per GPU of this node, rendezvous on 127.0.0.1 at a free port.  The ranks then take the normal path."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    # N ranks build their synthetic videos and states at once: cap the host threads of each so they do not oversubscribe the node
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_gpus) // n_gpus)))
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def dry_run(backend, args):
    """AF_BENCH_DRY_RUN=<backend> (tests, CPU): everything of the multi-rank launch path that does not need a GPU - rendezvous,
    process group, video shard of the rank, the barrier-bracketed timed region with its MAX all-reduce - then one JSON line."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group(backend)
    assert args.gpus == world, (args.gpus, world)
    dt = timed_region(lambda: time.sleep(0.01 * (rank + 1)), lambda: None, dist if world > 1 else None, torch.device("cpu"))
    st = rank_stats(timed_region.local_seconds, 1, "cpu:%d" % rank, dist if world > 1 else None, torch.device("cpu"))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "backend": backend, "n_gpus": world, "video_of_rank0": shard_for_rank(0, world), "max_region_s": dt,
                          "ranks": st, "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "videos": [shard_for_rank(r, world) for r in range(world)]}))


def model_rows(it, N, p_valid, two_layer, stop_global=5000):
    """MLP rows per net (map1, atlas, map2, alpha) of iteration `it` when a share p_valid of the 2N flow matches is valid
    (k_prep compacts them): the row model behind the HBM byte model below.  No GPU needed."""
    live = 2.0 * N * p_valid
    m = 5 * N + (2 * N if it <= stop_global else 0) + live
    return (m, (6 if two_layer else 3) * N, m if two_layer else 0, (3 * N + live) if two_layer else 0)


# bytes a chain moves through HBM per row: per hidden layer one 1 KB T-layout tile row of X_l (forward) or dZ_l
# (backward) + 32 B of sign bits written (forward) / read (backward); the PE tile of the atlas / alpha nets; coordinates in, outputs out
CHAIN_HIDDEN = {"map1": 5, "atlas": 7, "map2": 3, "alpha": 7}
CHAIN_PE_BYTES = {"map1": 0, "atlas": 160, "map2": 0, "alpha": 128}
# bf16 h/m/l weight stream of one orientation: 256x256 hidden layers x 3 levels x 2 B; every XCD's L2 (8 of them) fetches it once per launch
CHAIN_STREAM_BYTES = {"map1": 4 * 393216, "atlas": 6 * 393216, "map2": 2 * 393216, "alpha": 6 * 393216}
# f16x3 chains: two fp16 levels per 256x256 hidden layer
CHAIN_STREAM_BYTES_HF = {"map1": 4 * 262144, "atlas": 6 * 262144, "map2": 2 * 262144, "alpha": 6 * 262144}


def hbm_model_bytes(kernel, rows4, launches_per_step):
    """Expected HBM bytes per launch of a hot kernel from the layouts alone (what DESIGN.md 2.1 / 2.2 say must move), for
    the rows of one step.  bench.py quotes a committed PMC measurement as `traffic` only while it agrees with this model."""
    r = dict(zip(("map1", "atlas", "map2", "alpha"), rows4))
    if kernel.startswith("k_dw"):
        return sum(DW_TILE_BYTES[n] * r[n] / 32.0 for n in r) / launches_per_step
    per_row = {n: CHAIN_HIDDEN[n] * (1024 + 32) + CHAIN_PE_BYTES[n] + 32 for n in r}
    if kernel.startswith("k_mlp_bwd"):
        per_row["map1"] += 128; per_row["map2"] += 128     # dz of layer 0's three inputs is not stored; the x0 tile is read by k_dw, not here
    sb = CHAIN_STREAM_BYTES_HF if kernel.endswith(("_hf", "_hf<true>")) else CHAIN_STREAM_BYTES
    streams = 8 * sum(sb[n] * (2 if n == "map1" else 1) for n in r if r[n] > 0)     # mapping1 has tiles in both launches of a direction
    return (sum(per_row[n] * r[n] for n in r) + streams) / launches_per_step


EVENT_EVERY = 4     # the timed region's HIP events sit in every 4th step (af_set_timing's sample period)


def committed_traffic(kernel, model_bytes, tol=0.12):
    """The newest profiles/r*_traffic.json that holds `kernel`: (bytes per launch, source) when the measurement agrees
    with the layout model within `tol`, else (None, why)."""
    import glob
    import re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), key=lambda q: [int(x) for x in re.findall(r"r(\d+)", os.path.basename(q))[:1]] + [q], reverse=True)
    for tpath in paths:
        try:
            tj = json.load(open(tpath))
            b = tj["kernels"][kernel]["hbm_bytes"]
        except Exception:
            continue
        rel = os.path.relpath(tpath, ROOT)
        if model_bytes > 0 and abs(b / model_bytes - 1.0) <= tol:
            return b, "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; %s); layout model %.4g B, measured/model %.3f" % (rel, tj["correction"], model_bytes, b / model_bytes)
        return None, "%s holds %.4g B for this kernel but the layout model of this run says %.4g B (ratio %.3f): stale or another workload, not quoted" % (rel, b, model_bytes, b / model_bytes if model_bytes else float("nan"))
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--resx", type=int, default=768)
    ap.add_argument("--resy", type=int, default=432)
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pretrain-iters", type=int, default=1)
    ap.add_argument("--settle-steps", type=int, default=200, help="untimed loop iterations in front of the --warmup steps (0.2 s: lets the power manager reach the steady state "
                    "of this load before a short timed window; 0 = the cold start a fresh process gives)")
    ap.add_argument("--two-layer", action="store_true", help="BASELINE configs[4]: fg/bg dual-atlas path (stage1_neural_atlas_seg.py) instead of configs[1]")
    ap.add_argument("--videos-per-gpu", type=int, default=1, help="(a different workload from the headline one) V independent videos per GPU, "
                    "optimised concurrently from V host threads on V streams: the kernels of one fill the idle tail rounds of the others")
    ap.add_argument("--first-iter", type=int, default=-1, help="first timed iteration (default: K steps centred on the global-rigidity switch at 5000/5001)")
    ap.add_argument("--valid-fraction", type=float, default=1.0, help="keep this fraction of the (synthetic, ~0.98 valid) flow-consistency masks, chosen at random "
                    "per pixel: real RAFT masks are far from all-valid, and only valid rows are credited as algorithmic work (loss_utils.py:326-356)")
    args = ap.parse_args()

    # AF_BENCH_SHARE_DEVICE=1 (a rehearsal, never a result): all ranks on GPU 0, gloo for the barrier / MAX since RCCL refuses two ranks on one
    # device — everything of the N-rank path except N GPUs: N processes initialise, build their videos, pass the barrier-bracketed window, gather
    # `ranks`, print ONE line (labelled).  The one-GPU box is the only hardware this repository has ever been given.
    share = os.environ.get("AF_BENCH_SHARE_DEVICE", "0") not in ("", "0")
    if args.gpus > 1 and "RANK" not in os.environ:         # bare `python bench.py --gpus N`: spawn the N ranks ourselves
        if not os.environ.get("AF_BENCH_DRY_RUN") and not share:
            check_gpu_count(args.gpus)
        return self_spawn(args.gpus, sys.argv[1:])
    if os.environ.get("AF_BENCH_DRY_RUN"):
        return dry_run(os.environ["AF_BENCH_DRY_RUN"], args)
    import torch
    import aiod_amd
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if share:
        local = 0
    n_visible = check_gpu_count(1 if share else max(args.gpus, local + 1))      # a launcher that started more ranks than there are GPUs: say so, once per rank, before set_device
    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if share else dev          # where the barrier's / all-reduce's tensors live
    assert args.gpus == world, "--gpus %d but WORLD_SIZE %d: launch with torch.distributed.run --nproc-per-node == --gpus (or bare, without RANK in the environment)" % (args.gpus, world)

    cfg = aiod_amd.default_config(args.resx, args.resy, args.frames, two_layer=args.two_layer)
    N = cfg.samples_batch
    af = aiod_amd.AtlasFit(cfg, device=local)
    configure_modes(af.arithmetic["mlp_mode"], af.arithmetic["dw_mode"])
    vseed = shard_for_rank(rank, world)[0]
    video = synth_video_device(args.resx, args.resy, args.frames, seed=vseed, device=dev)
    if args.valid_fraction < 1.0:
        gm = torch.Generator(device=dev).manual_seed(97 + vseed)
        video = video[:3] + tuple(m * (torch.rand(m.shape, device=dev, generator=gm) < args.valid_fraction).float() for m in video[3:5])
    if args.two_layer:
        video = video + (synth_fg_mask_device(args.resx, args.resy, args.frames, seed=vseed, device=dev),)
    af.upload_video(*video)
    sds = init_state_dicts(1234 + rank, args.two_layer)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    pre_ms = None
    if args.pretrain_iters > 0:                                  # outside the timed region; puts the mapping(s) in a realistic regime
        af.pre_train_mapping(1, seed=rank + 7)                   # warm the code path, then time pre_train_mapping for the record
        torch.cuda.synchronize(); t0 = time.perf_counter()
        af.pre_train_mapping(args.pretrain_iters, seed=rank)
        torch.cuda.synchronize(); pre_ms = (time.perf_counter() - t0) * 1e3 / (args.pretrain_iters * args.frames)
        if args.two_layer:
            af.pre_train_mapping(args.pretrain_iters, seed=rank + 100, net=aiod_amd.NET_MAPPING2)

    K, W = args.steps, args.warmup
    switch = cfg.stop_global_rigidity + 1                           # first iteration without the global term
    first = max(0, switch - K // 2) if args.first_iter < 0 else args.first_iter
    classes = af.TIMING_NAMES      # prep, fwd_1, fwd_2, loss, bwd_1, bwd_2, dw, adam (include/atlasfit.h: af_get_timing)
    by_name = {}
    for c in classes:
        by_name.setdefault(KERNEL_OF_CLASS[c], []).append(c)

    # ---- settle (untimed, in front of the W warm-up steps): the board's power manager needs a few tens of milliseconds of THIS load to reach its
    # steady state — a 20-step window behind 5 warm-up steps reads up to 5 % low after seconds of interpreter start and light setup kernels
    # (tools/window_probe.py) — and the metric is the throughput of a 10 000-iteration job, not of its first 30 ms.  The line says so.
    wfirst = max(0, first - W)
    if args.settle_steps > 0:
        af.train_steps(max(0, wfirst - args.settle_steps), args.settle_steps, None, seed=rank + 900, return_losses=False)
    # ---- warm-up (W untimed steps, all kernel classes timed to find the dominant one)
    af.set_timing(0xFFFF)
    if W > 0:
        af.train_steps(wfirst, W, None, seed=rank, return_losses=False)
    tw = af.timing(reset=True)
    # the dominant KERNEL (by name, all of its launches in a step: k_mlp_fwd_multi = fwd_1 + fwd_2, ...), not the dominant launch
    dom = max(by_name, key=lambda k: sum(tw[c][0] for c in by_name[k])) if W > 0 else KERNEL_OF_CLASS["dw"]
    dom_classes = by_name[dom]
    # events only around the dominant kernel's launches, and only in every EVENT_EVERY-th timed step: an event costs ~5 us in-stream
    # (the all-launches pass below runs 0.06 ms per step slower than the timed region for its 12 further events), four of them in
    # every step of a 1.1 ms step would take 1.8 % off `value` for the sake of measuring `roofline`
    af.set_timing(sum(1 << classes.index(c) for c in dom_classes), every=EVENT_EVERY)

    # ---- optional extra videos on this GPU (own handle, stream, weights, table), warmed like the first
    extra = []
    for v in range(1, max(1, args.videos_per_gpu)):
        h2 = aiod_amd.AtlasFit(cfg, device=local)
        vid2 = synth_video_device(args.resx, args.resy, args.frames, seed=vseed + 1000 * v, device=dev)
        if args.two_layer:
            vid2 = vid2 + (synth_fg_mask_device(args.resx, args.resy, args.frames, seed=vseed + 1000 * v, device=dev),)
        h2.upload_video(*vid2)
        sd2 = init_state_dicts(4321 + rank + 17 * v, args.two_layer)
        for net in h2.nets:
            h2.load_state_dict(net, sd2[net])
        if args.pretrain_iters > 0:
            h2.pre_train_mapping(args.pretrain_iters, seed=rank + v)
            if args.two_layer:
                h2.pre_train_mapping(args.pretrain_iters, seed=rank + 100 + v, net=aiod_amd.NET_MAPPING2)
        if W > 0:
            h2.train_steps(wfirst, W, None, seed=rank + v, return_losses=False)
        extra.append(h2)

    # ---- timed region: EXACTLY K steps (of every video on this GPU), the default (fp32-faithful) arithmetic
    got = {}

    def run_all():
        import threading
        ths = [threading.Thread(target=lambda h=h2, v=i: h.train_steps(first, K, None, seed=rank + 1 + v, return_losses=False)) for i, h2 in enumerate(extra)]
        for t in ths:
            t.start()
        got.update(losses=af.train_steps(first, K, None, seed=rank, return_losses=True))
        for t in ths:
            t.join()
    dt = timed_region(run_all, torch.cuda.synchronize, dist if world > 1 else None, cdev)
    ranks = rank_stats(timed_region.local_seconds, K, torch.cuda.get_device_name(local), dist if world > 1 else None, cdev)
    tk = af.timing(reset=True)

    # ---- per-kernel pass (NOT the timed region): the same K iterations once more, HIP events around every launch, clocks as
    # hot as the timed region left them.  Its per-step sum exceeds ms_per_step by the event overhead; `by_kernel` says so.
    af.set_timing(0xFFFF)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    af.train_steps(first, K, None, seed=rank + 50, return_losses=False)
    torch.cuda.synchronize(); dt_all_events = time.perf_counter() - t0
    ta = af.timing(reset=True)
    af.set_timing(0)

    # ---- second value, never the headline: the same K steps with the weight-gradient GEMM on three bf16 products (narrower
    # than the reference's fp32; af_set_dw_mode(h, 2)), timed by the same rule
    dt3 = None
    if DW_MODE == 1 and not extra:
        af.set_dw_mode(2)
        af.train_steps(max(0, first - 5), 5, None, seed=rank + 60, return_losses=False)
        dt3 = timed_region(lambda: af.train_steps(first, K, None, seed=rank + 61, return_losses=False), torch.cuda.synchronize, dist if world > 1 else None, cdev)
        af.set_dw_mode(1)
    # ---- third value, never the headline: the same K steps with the chains on bf16x6 (the headline arithmetic of rounds 2-5), timed by the same rule
    dt6 = None
    if MLP_MODE == 3 and not extra:
        af.set_mlp_mode(1)
        af.train_steps(max(0, first - 5), 5, None, seed=rank + 70, return_losses=False)
        dt6 = timed_region(lambda: af.train_steps(first, K, None, seed=rank + 71, return_losses=False), torch.cuda.synchronize, dist if world > 1 else None, cdev)
        af.set_mlp_mode(3)

    # Only valid flow matches are evaluated (k_prep compacts them like the reference's torch.where); the library
    # accumulates FLOPs per launch from the planned rows, so the work of the rows behind the live count is taken out here
    # from the measured valid counts.
    L = got["losses"]
    nv = L[:, -4:-2] if args.two_layer else L[:, 6:8]              # (#valid fwd, #valid bwd) per iteration
    FWD = {"map1": 526848.0, "map2": 264704.0, "alpha": 802304.0}; DX = {"map1": 525312.0, "map2": 263168.0, "alpha": 786944.0}
    inv = float((2 * N - nv.sum(axis=1)).sum())                     # invalid flow-match rows over the K steps, per net
    nets_inv = ("map1", "map2", "alpha") if args.two_layer else ("map1",)
    masked_step_flops = inv * sum(2 * FWD[n] + DX[n] for n in nets_inv)          # fwd + dW + dX of rows the reference skips
    masked_dw_flops = inv * sum(FWD[n] for n in nets_inv)
    inv_per_step = inv / K
    rows_k = [af.step_work(first + k)[0] for k in range(K)]             # rows per net (NET_* order: map1, atlas, map2, alpha)

    def live_rows(rows4):     # rows of one step with the dead (invalid-match) rows removed
        return (rows4[0] - inv_per_step, rows4[1], (rows4[2] - inv_per_step) if args.two_layer else 0, (rows4[3] - inv_per_step) if args.two_layer else 0)
    rows_mean = tuple(sum(live_rows(r)[j] for r in rows_k) / K for j in range(4))
    dw_alg_bytes = hbm_model_bytes("k_dw", rows_mean, 1)           # algorithmic = every operand tile of every layer read once

    def masked_of(kname, cls, inv_rows=None):
        inv_rows = inv if inv_rows is None else inv_rows
        if "dw" in cls:
            return inv_rows * sum(FWD[n] for n in nets_inv)
        return inv_rows * (sum(FWD[n] for n in nets_inv) if kname.startswith("k_mlp_fwd") else sum(DX[n] for n in nets_inv)) if kname.startswith("k_mlp") else 0.0

    # ---- roofline of the dominant kernel: algorithmic FLOPs / bytes per launch over the mean HIP-event duration of the TIMED region
    sampled = list(range(0, K, EVENT_EVERY))                         # the timed steps whose launches carried events (af_set_timing's period)
    inv_sampled = float((2 * N - nv[sampled].sum(axis=1)).sum())
    n_launch = max(sum(tk[c][1] for c in dom_classes), 1)
    # the sample period restarts at k = 0 of every af_train_steps call and the timed region is ONE call: the launches counted by the library
    # must be a whole number per sampled step, or `sampled` (and the masked FLOPs charged to it) no longer describes what was timed
    assert n_launch % len(sampled) == 0 and n_launch // len(sampled) == len(dom_classes), (n_launch, len(sampled), dom_classes)
    launches_per_step = n_launch / len(sampled)
    dom_ms = sum(tk[c][0] for c in dom_classes) / n_launch
    is_dw = "dw" in dom_classes
    flops_launch = (sum(tk[c][2] for c in dom_classes) - masked_of(dom, dom_classes, inv_sampled)) / n_launch
    achieved = flops_launch / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    total_flops = sum(af.step_work(first + k)[1] for k in range(K)) - masked_step_flops
    mfma_peak = (BF16X6_PEAK_TFLOPS * 6.0 / DW_PRODUCTS if DW_BF else FP32_MFMA_PEAK_TFLOPS) if is_dw else chain_peak_tflops()
    if is_dw and DW_BF:      # on the bf16 matrix pipe this kernel's 64 FLOP/B sit under the HBM roof: the roofline is bytes, not flops
        roof = {"bound": "hbm", "achieved": dw_alg_bytes / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": dw_alg_bytes}
    else:
        roof = {"bound": "mfma", "achieved": achieved, "peak": mfma_peak, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["peak_definition"] = ("HBM3E 8 TB/s" if roof["bound"] == "hbm" else
                               ("FP32 MFMA peak" if mfma_peak == FP32_MFMA_PEAK_TFLOPS else
                                "dense 16-bit MFMA peak 2500 TFLOP/s / %d partial products per fp32-faithful product" % round(BF16_MFMA_PEAK_TFLOPS / mfma_peak)))
    # every hot kernel by name, from the per-kernel pass above
    by_kernel = {"_source": "separate pass of the same %d iterations after the timed region, HIP events around every launch (%.4f ms/step wall with the events, "
                            "the timed region has them around the dominant kernel only)" % (K, dt_all_events / K * 1e3)}
    for kname, cls in by_name.items():
        ms = sum(ta[c][0] for c in cls); nl = sum(ta[c][1] for c in cls)
        fl = sum(ta[c][2] for c in cls) - masked_of(kname, cls)
        if ms > 0:
            by_kernel[kname] = {"ms_per_step": ms / K, "launches_per_step": nl / K, "tflops": (fl / ms / 1e9) if fl > 0 else None,
                                "frac_of_fp32_mfma_peak": (fl / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS) if fl > 0 else None,
                                "frac_of_bf16x6_peak": (fl / ms / 1e9 / BF16X6_PEAK_TFLOPS) if fl > 0 else None}
            if kname.startswith("k_mlp") and MLP_MODE == 3 and fl > 0:      # this kernel's own arithmetic: three fp16 products per product
                by_kernel[kname]["frac_of_f16x3_peak"] = fl / ms / 1e9 / F16X3_PEAK_TFLOPS
                by_kernel[kname].pop("frac_of_bf16x6_peak", None)
            if "dw" in cls:
                if DW_BF and fl > 0 and DW_PRODUCTS != 6:      # this kernel's own arithmetic: DW_PRODUCTS bf16 products per product
                    by_kernel[kname]["frac_of_bf16x%d_peak" % DW_PRODUCTS] = fl / ms / 1e9 / (BF16X6_PEAK_TFLOPS * 6.0 / DW_PRODUCTS)
                    by_kernel[kname].pop("frac_of_bf16x6_peak", None)
                by_kernel[kname]["algorithmic_hbm_gbs"] = dw_alg_bytes / (ms / nl * 1e-3) / 1e9
                by_kernel[kname]["frac_of_hbm_peak"] = by_kernel[kname]["algorithmic_hbm_gbs"] / HBM_PEAK_GBS
            if kname.startswith(("k_mlp", "k_dw")):
                by_kernel[kname]["hbm_model_bytes_per_launch"] = hbm_model_bytes(kname, rows_mean, nl / K)

    # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process; the committed summary of the same
    # command (tools/collect_profiles.sh -> profiles/r*_traffic.json, FETCH_SIZE x2 + WRITE_SIZE per launch) is quoted only when it
    # covers this kernel and agrees with the byte model of THIS run's rows (hbm_model_bytes) - a stale file yields null.
    traffic, traffic_src = None, None
    if (args.resx, args.resy, args.frames) == (768, 432, 80) and not extra:
        traffic, traffic_src = committed_traffic(dom, hbm_model_bytes(dom, rows_mean, launches_per_step))

    out = None
    if rank == 0:
        V = 1 + len(extra)
        value = world * V * N * K / dt
        out = {
            "metric": METRIC, "value": value, "unit": "sampled points/s",
            "n_gpus": world, "n_gpus_visible": n_visible, **({"shared_device_rehearsal": "AF_BENCH_SHARE_DEVICE=1: all %d ranks on GPU 0 over gloo - exercises the multi-rank path, NOT a scaling result" % world} if share else {}), "ranks": ranks, "steps": K, "warmup": W, "settle_steps": args.settle_steps, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (" + "; ".join(x for x in (
                ("MLP chains f16x3: operands split into 2 fp16 of their value scaled per row, 3 partial products" if MLP_MODE == 3 else
                 "MLP chains bf16x6: operands split into 3 bf16, 6 partial products" + (" (EXPERIMENT: backward chain on 3 products)" if MLP_MODE == 2 else "")) if MLP_BF else "",
                {1: "weight-gradient GEMM bf16x6", 2: "NON-DEFAULT weight-gradient GEMM bf16x3: 2 bf16 per operand, 3 partial products, narrower than fp32"}.get(DW_MODE, "")) if x)
                + "; fp32 accumulate)") if (MLP_BF or DW_BF) else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]%s: single video %d frames %dx%d, samples_batch %d, shipped config_flow_100.json; "
                                   "timed iterations %d..%d (global-rigidity rows while i <= 5000); %s"
                                   % (4 if args.two_layer else 1, " (fg/bg dual atlas + alpha MLP)" if args.two_layer else "",
                                      args.frames, args.resx, args.resy, N, first, first + K - 1,
                                      "one video per GPU" if V == 1 else "%d independent videos per GPU, concurrently (NOT the headline workload)" % V),
                       "videos_per_gpu": V,
                       "samples_batch": N, "frames": args.frames, "resx": args.resx, "resy": args.resy},
            "pretrain_ms_per_step": pre_ms,
            "roofline": {**roof, "kernel": dom, "kernel_launch_classes": dom_classes, "achieved_tflops": achieved,
                         "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                         "kernel_ms": dom_ms, "kernel_events": "HIP events on the handle's stream around this kernel's launches in every %d-th of the %d timed steps (%d launches)" % (EVENT_EVERY, K, n_launch), "flops_per_launch": flops_launch, "launches_per_step": launches_per_step,
                         "by_kernel": by_kernel,
                         "valid_flow_fraction": float(nv.sum() / (2.0 * N * K)),
                         "whole_step_tflops": V * total_flops / dt / 1e12, "whole_step_frac": V * total_flops / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS},
        }
        if dt3 is not None:     # extra keys, never `value`: the opt-in three-product weight-gradient GEMM on the same steps
            out["value_bf16x3_dw"] = world * N * K / dt3
            out["ms_per_step_bf16x3_dw"] = dt3 / K * 1e3
            out["value_bf16x3_dw_note"] = "same K steps with af_set_dw_mode(h, 2): k_dw_bf<3>, 16-bit-mantissa operands, narrower than the reference's fp32 - not the headline"
        if dt6 is not None:     # extra keys, never `value`: the chains on six bf16 products (rounds 2-5's headline arithmetic) on the same steps
            out["value_bf16x6"] = world * N * K / dt6
            out["ms_per_step_bf16x6"] = dt6 / K * 1e3
            out["value_bf16x6_note"] = "same K steps with af_set_mlp_mode(h, 1): k_mlp_*_multi_bf, the fp32-faithful arithmetic of rounds 2-5, kept as the cross-check of f16x3"
        out["arithmetic"] = af.arithmetic
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.resx, args.resy, args.frames, 0, sds, video, args.cpu_seconds, args.two_layer)
            except Exception as e:   # the baseline is a reported number, never the product
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
    af.close()
    for h2 in extra:
        h2.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
