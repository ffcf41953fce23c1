"""BASELINE configs[4]'s acceptance metric — "fg+bg dual atlas with alpha MLP, PSNR parity vs reference" — on a COMPLETE
schedule: the fg/bg twin of tests/test_gpu_c1.py.  80 frames 160x90 with a moving soft-edged disc as foreground mask, both
`pre_train_mapping`s (100 x F steps each, stage1_neural_atlas_seg.py:173-179), 1001 iterations of the four-net loop
(:193-311), alpha-blended reconstruction PSNR (evaluate.py:302-337), through the HIP path — against what the REFERENCE's own
seg modules reached on the same videos, seeds and random draws (tests/golden/c1_seg_reference.npz, written by
oracle/make_golden_c1_seg.py in the build container, ~1 h of CPU per run).

Every random draw of the reference run came from torch's global CPU generator in the reference's order and is replayed here
from the seed alone: nn.Linear init of mapping1, mapping2, atlas, alpha; per pre-train step the row then the column draw,
mapping1's 8000 steps then mapping2's; one torch.randint(P, (N, 1)) per loop iteration.

Tolerances are built like test_gpu_c1.py's: the fixture holds seeds 0..2 at TWO thread counts — the reference against itself, only the
summation order inside its GEMMs differs — and BASELINE.md's 0.1 dB is asserted on top of two standard errors of the measured
run-to-run noise.  Round 4 measured the SAME sensitivity on this side: another split-K partition of the weight-gradient GEMM (another
summation order, nothing else: `af_debug_set_dw_cost`; with round 3's partition the round-4 kernels reproduce round 3's 24.6563 dB on seed 1 to the
last digit) moves this path's final PSNR on seed 1 over 23.37 .. 24.66 dB (six partitions, sd 0.5 dB) — the four-net schedule is
chaotic at the 0.5 dB level per seed on BOTH sides.  Every seed is therefore run on several partitions here, the HIP side's own sigma is
estimated from their spread, seeds are compared by their means, and both sigmas enter the tolerances."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_seg_reference.npz")


# split-K partitions of k_dw the seeds are run on (host.hip build_sched: per-shape tile costs; None = the shipped row): same arithmetic, another summation order
PARTITIONS = (None, "306,150,126,129,87", "306,170,145,148,100")


def _run(seed, g, injected=True, partition=None):
    return _run_inner(seed, g, injected, partition)


def _run_inner(seed, g, injected=True, partition=None):
    import aiod_amd
    import bench
    from oracle import atlas_oracle as O
    resx, resy, F = int(g["resx"]), int(g["resy"]), int(g["nframes"])
    iters, pre_iters = int(g["iters"]), int(g["pretrain_iters"])
    k = [i for i, s in enumerate(g["seeds"]) if int(s) == seed][0]
    flow = str(g["flow_kind"][k]) if "flow_kind" in g else "constant"      # round 4: odd further seeds run on the per-pixel flow field
    v = O.synthetic_seg_video(resx, resy, F, seed=seed, flow=flow)
    assert abs(float(v.video_frames.double().sum()) - float(g["video_checksum"][k])) < 1e-6
    assert abs(float(v.mask_frames.double().sum()) - float(g["mask_checksum"][k])) < 1e-6
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F, two_layer=True))
    if partition is not None:
        af.set_dw_cost(partition)      # af_debug_set_dw_cost: another split-K partition, validated by the library
    af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask, v.mask_frames)
    sds = bench.init_state_dicts(seed, two_layer=True)          # torch.manual_seed(seed) + nn.Linear init: mapping1, mapping2, atlas, alpha
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    N, P = af.N, F * resx * resy
    steps = pre_iters * F
    if injected:
        for net in (aiod_amd.NET_MAPPING1, aiod_amd.NET_MAPPING2):      # the global generator continues where the init left it
            ys = torch.empty((steps, 10000), dtype=torch.int64); xs = torch.empty((steps, 10000), dtype=torch.int64)
            for s in range(steps):
                ys[s] = torch.randint(resy, (10000, 1)).view(-1)
                xs[s] = torch.randint(resx, (10000, 1)).view(-1)
            af.pre_train_mapping(pre_iters, ys.numpy(), xs.numpy(), net=net)
            del ys, xs
        p_pre, _ = af.psnr()
        inds = torch.stack([torch.randint(P, (N, 1)).view(-1) for _ in range(iters)])
        losses = af.train_steps(0, iters, inds.numpy())
    else:                                                           # the product's own Philox sampler (what stage1_seg.main uses)
        af.pre_train_mapping(pre_iters, seed=1000 + seed)
        af.pre_train_mapping(pre_iters, seed=3000 + seed, net=aiod_amd.NET_MAPPING2)
        p_pre, _ = af.psnr()
        losses = af.train_steps(0, iters, None, seed=2000 + seed)
    p_end, per = af.psnr()
    af.close()
    return p_pre, p_end, per, losses


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/c1_seg_reference.npz not generated yet")
def test_configs4_full_schedule_psnr_within_0p1_db_of_reference():
    g = dict(np.load(GOLDEN))
    fp32 = g["double"] == 0
    seeds = sorted({int(s) for s in g["seeds"][fp32]})
    every = int(g["log_every"])
    arms = {s: [i for i in np.nonzero(fp32)[0] if int(g["seeds"][i]) == s] for s in seeds}      # the reference runs of each seed
    pairs = [arms[s] for s in seeds if len(arms[s]) >= 2]
    assert pairs, "the fixture must hold at least one seed at two thread counts (the reference's own reproducibility)"
    d = np.array([g["psnr"][p[0]] - g["psnr"][p[1]] for p in pairs])
    sigma_run = float(np.sqrt(np.mean(d ** 2) / 2.0))             # one run's standard deviation from the pairs of runs
    spread_curve = float(max(np.max(np.abs(g["curves"][p[0]][:, 11] / g["curves"][p[1]][:, 11] - 1.0)) for p in pairs))
    n_runs = sum(len(arms[s]) for s in seeds)
    print("reference against itself (%s threads): final PSNR per run %s dB, pairs differ by %s dB -> sigma of one run %.3f dB ; total-loss curves up to %.1f %% apart"
          % (sorted({int(t) for t in g["threads"][fp32]}), np.array2string(g["psnr"][fp32], precision=3), np.array2string(d, precision=3), sigma_run, 100 * spread_curve))
    hip, ref_means, dev_in = [], [], []
    for seed in seeds:
        refs = [float(g["psnr"][i]) for i in arms[seed]]
        ref_pre = [float(g["psnr_pre"][i]) for i in arms[seed]]
        parts = PARTITIONS if len(refs) >= 2 else PARTITIONS[:1]          # seeds with one reference arm: the shipped partition only (suite time)
        runs = [_run(seed, g, partition=p) for p in parts]
        ends = np.array([r[1] for r in runs]); pres = np.array([r[0] for r in runs])
        dev_in.append(ends - ends.mean())
        curve = runs[0][3][::every, :12]
        rel_total = np.min([np.abs(curve[:, 11] / g["curves"][i][:, 11] - 1.0) for i in arms[seed]], axis=0)       # to the nearer arm
        print("seed %d (%s flow): PSNR after the pre-trains hip %s / reference %s ; after %d iterations hip %s (mean %.4f) / reference %s"
              % (seed, str(g["flow_kind"][arms[seed][0]]) if "flow_kind" in g else "constant", np.array2string(pres, precision=4), np.array2string(np.array(ref_pre), precision=4),
                 int(g["iters"]), np.array2string(ends, precision=4), ends.mean(), np.array2string(np.array(refs), precision=4)))
        print("   total loss every %d iterations, hip:       %s" % (every, np.array2string(curve[:, 11], precision=2)))
        for i in arms[seed]:
            print("   total loss every %d iterations, ref/%d thr: %s" % (every, int(g["threads"][i]), np.array2string(g["curves"][i][:, 11], precision=2)))
        # 16 000 pre-train steps on the same draws: the reconstruction PSNR with an untrained atlas moves by ~0.1 dB with the summation order
        # on this side as well (17.23 .. 17.36 dB over six partitions on seed 1)
        assert abs(float(pres.mean()) - float(np.mean(ref_pre))) < 0.25, (seed, pres, ref_pre)
        assert rel_total[0] < 0.10, (curve[0], [g["curves"][i][0] for i in arms[seed]])
        assert rel_total.max() < 0.05 + 1.5 * spread_curve                  # the curves stay as close as the reference's own two
        hip.append(ends); ref_means.append(float(np.mean(refs)))
    dev = np.concatenate(dev_in)
    dof = max(1, sum(len(h) - 1 for h in hip))
    sigma_hip = float(np.sqrt((dev ** 2).sum() / dof))            # this path's own sigma of one run, pooled over the seeds' partitions
    print("HIP path against itself (partitions %s): pooled sigma of one run %.3f dB (reference: %.3f dB)" % (list(PARTITIONS), sigma_hip, sigma_run))
    d = np.array([h.mean() - r for h, r in zip(hip, ref_means)])
    for seed, h, r, dd in zip(seeds, hip, ref_means, d):
        n_ref = len(arms[seed])
        tol_seed = 0.1 + 2.0 * np.sqrt(sigma_hip ** 2 / len(h) + sigma_run ** 2 / n_ref)
        print("seed %d: hip %.4f - reference %.4f = %+.4f dB (tolerance %.3f dB)" % (seed, h.mean(), r, dd, tol_seed))
        assert abs(dd) <= tol_seed, (seed, h, r)
    se = float(d.std(ddof=1) / np.sqrt(len(d)))
    se_model = float(np.sqrt(np.mean([sigma_hip ** 2 / len(h) + sigma_run ** 2 / len(arms[s_]) for h, s_ in zip(hip, seeds)]) / len(d)))
    print("mean PSNR over seeds %s: hip - reference = %+.4f dB, standard error %.4f dB from the paired differences (%.4f from the two sigmas), n = %d, t = %+.2f"
          % (seeds, d.mean(), se, se_model, len(d), d.mean() / se))
    assert abs(float(d.mean())) <= 0.1 + 2.0 * max(se, se_model), (d, se, se_model)
    # the product's own device sampler on the first seed (different draws): same quality of fit
    p_dev = _run(seeds[0], g, injected=False)[1]
    print("seed %d with the device sampler: %.4f dB" % (seeds[0], p_dev))
    assert abs(p_dev - ref_means[0]) <= 0.1 + 2.0 * np.sqrt(sigma_hip ** 2 + sigma_run ** 2) + 0.2
