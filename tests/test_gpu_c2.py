"""BASELINE configs[1] — the headline config — end to end against the REFERENCE's own modules: 80 frames
768x432, pre_train_mapping 100 x F = 8000 steps, iters_num 10001 of the shipped config (config_flow_100.json:6,44), i.e. 5001 iterations with
the global-rigidity rows and 5000 without them (stage1_neural_atlas.py:151-231,246-251).

tests/golden/c2_reference.npz is written by `oracle/make_golden_c1.py --resx 768 --resy 432 --iters 10001 --log-every 250 --psnr-at 5000` in the
build container (the reference's IMLP / loss functions / pre_train_mapping / torch.optim.Adam; 3-8 CPU-hours per run): per seed the PSNR after
the pre-train, after 5000 iterations and at the end, and the six loss terms every 250 iterations.  Seeds 0, 2, 5, 7 run on the translating video,
seeds 1, 4, 6, 8 on the video whose flow differs at every pixel of every frame (holed masks); the record names each seed's video.  tests/golden/c2_reference_rerun.npz holds SECOND arms of the
same seeds at another thread count (another summation order inside the reference's GEMMs and nothing else): the reference against itself, which
gives its run-to-run sigma at this size from its own pairs.

Every random draw of a reference run came from torch's global CPU generator in the reference's order and is replayed here from the seed alone
(as in test_gpu_c1.py).  The two fp32 trajectories decorrelate over 18 000 Adam steps; what must agree is where they pass and where they end:
the paired difference hip - reference (against the mean of a seed's arms) over the seeds, with its standard error from the seeds' own spread.
BASELINE.md's 0.1 dB is asserted on the mean on top of two standard errors, AND THE STANDARD ERROR ITSELF IS BOUNDED (a comparison that cannot
resolve 0.1 dB must not pass as one that did), at the end and at the global-rigidity switch.

GPU time (the driver's suite has 1200 s): one run per seed on the shipped split-K partition here; AF_C2_ALL_PARTITIONS=1 adds two more partitions
per seed (another summation order on THIS side: its own run-to-run sigma, printed and used for the per-seed tolerance) — the builder's own gpurun
executes that variant, its log is profiles/r6_pytest_c2_all_partitions.log and its pooled sigma is SIGMA_HIP_RECORDED below."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COMPLETE = os.path.join(GOLD, "c2_reference.npz")
ALL_PARTITIONS = (None, "306,150,126,129,87", "306,170,145,148,100")
PARTITIONS = ALL_PARTITIONS if os.environ.get("AF_C2_ALL_PARTITIONS") else ALL_PARTITIONS[:1]
# dB: run-to-run sigma of ONE run of this path by kind of video, pooled over the three partitions and both evaluations of the eight seeds
# (profiles/r6_pytest_c2_all_partitions.log); used for the per-seed net when the default variant runs one partition only
SIGMA_HIP_RECORDED = {"constant": 0.061, "field": 0.157}
# dB: the largest standard error of the mean paired difference under which "within 0.1 dB" counts as a statement the comparison resolved.  What n = 8
# CAN resolve follows from the sigmas above and the reference's own (0.03 / 0.16 dB): ~0.04 dB at the end with three partitions per seed, ~0.065 dB with
# one; at the switch the field-flow videos spread more on both sides (the reference's two arms of seed 4: 0.34 dB apart).  VERDICT r5's 0.035 dB was NOT
# reached — MEASUREMENTS.md A.4 has the numbers and what limits them.
_ALL = bool(os.environ.get("AF_C2_ALL_PARTITIONS"))
SE_MAX = {"after 5000 iterations": 0.08 if _ALL else 0.11, "at the end": 0.05 if _ALL else 0.075}
ITER0_TOL = 0.25
TERMS = ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total")


def _records():
    """{seed: dict(flow, psnr_pre, psnr_at {iteration: dB}, psnr_end or None, curve (n, 6), every, iters, resx, resy, nframes, checksum or None)}"""
    recs = {}
    if os.path.exists(COMPLETE):
        g = dict(np.load(COMPLETE))
        for k, s in enumerate(g["seeds"]):
            at = {int(i): float(p) for i, p in zip(g["psnr_at_iter"], g["psnr_at"][k])} if "psnr_at" in g else {}
            recs[int(s)] = dict(flow=str(g["flow_kind"][k]), psnr_pre=float(g["psnr_pre"][k]), psnr_at=at, psnr_end=float(g["psnr"][k]),
                                curve=g["curves"][k], every=int(g["log_every"]), iters=int(g["iters"]), resx=int(g["resx"]), resy=int(g["resy"]),
                                nframes=int(g["nframes"]), checksum=float(g["video_checksum"][k]), cpu_seconds=g["cpu_seconds"][k],
                                threads=int(g["threads_per_seed"][k]) if "threads_per_seed" in g else -1)
    return recs


def _video(seed, rec):
    from oracle import atlas_oracle as O
    v = O.synthetic_video(rec["resx"], rec["resy"], rec["nframes"], seed=seed, flow=rec["flow"])
    if rec["checksum"] is not None:
        assert abs(float(v.video_frames.double().sum()) - rec["checksum"]) < 1e-6 * rec["checksum"]
    return v


def _run(seed, rec, partition, upto, v):
    """The HIP path on the reference run's video, initial weights and draws: PSNR after the pre-train, at the reference's intermediate
    evaluations, at the end (None when `upto` stops earlier), and the loss terms of every iteration up to `upto`."""
    import aiod_amd
    import bench
    resx, resy, F = rec["resx"], rec["resy"], rec["nframes"]
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F))
    if partition is not None:
        af.set_dw_cost(partition)
    af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask)
    sds = bench.init_state_dicts(seed)                      # torch.manual_seed(seed) + nn.Linear init, mapping then atlas
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    N, P = af.N, F * resx * resy
    steps = 100 * F                                         # pretrain_iter_number 100 (config_flow_100.json:36)
    ys = torch.empty((steps, 10000), dtype=torch.int64); xs = torch.empty((steps, 10000), dtype=torch.int64)
    for s in range(steps):                                  # the global generator continues where the init left it
        ys[s] = torch.randint(resy, (10000, 1)).view(-1)
        xs[s] = torch.randint(resx, (10000, 1)).view(-1)
    af.pre_train_mapping(100, ys.numpy(), xs.numpy())
    del ys, xs
    p_pre, _ = af.psnr()
    stops = sorted({i for i in rec["psnr_at"] if i <= upto} | {upto})
    losses, p_at, done = [], {}, 0
    for stop in stops:
        if stop > done:
            # the reference draws torch.randint(P, (N, 1)) once per iteration from the global CPU generator; ONE call for all the iterations of the
            # segment consumes the same stream element by element (the CPU kernel is serial) and leaves the generator in the same state —
            # pinned on CPU by tests/test_oracle.py::test_one_randint_call_replays_the_per_iteration_draws — at a tenth of the time
            inds = torch.randint(P, ((stop - done) * N, 1)).view(stop - done, N)
            losses.append(af.train_steps(done, stop - done, inds.numpy()))
            done = stop
        if stop in rec["psnr_at"]:
            p_at[stop] = af.psnr()[0]
    p_end = af.psnr()[0] if upto == rec["iters"] else None
    af.close()
    return p_pre, p_at, p_end, np.concatenate(losses)


def _second_arms(recs):
    """{seed: (psnr_pre, {iteration: psnr}, psnr_end, threads)} of tests/golden/c2_reference_rerun.npz: the same seed through the reference's modules at
    another thread count (another summation order inside its GEMMs and nothing else) — the reference against itself at this size."""
    rr, out = os.path.join(GOLD, "c2_reference_rerun.npz"), {}
    if os.path.exists(rr):
        g2 = dict(np.load(rr))
        for k, s2 in enumerate(g2["seeds"]):
            if int(s2) in recs and recs[int(s2)]["psnr_end"] is not None:
                out[int(s2)] = (float(g2["psnr_pre"][k]), {int(i): float(p) for i, p in zip(g2["psnr_at_iter"], g2["psnr_at"][k])}, float(g2["psnr"][k]), int(g2["threads_per_seed"][k]))
    return out


@pytest.mark.skipif(not _records(), reason="no configs[1] reference record (tests/golden/c2_reference.npz)")
def test_configs1_full_schedule_against_the_reference_modules():
    recs = _records()
    seeds = sorted(recs)
    arms2 = _second_arms(recs)
    npart = len(PARTITIONS)
    d_pre, d_mid, d_end, narm, sig = [], [], [], [], []
    for seed in seeds:
        rec = recs[seed]
        every, n_logged = rec["every"], len(rec["curve"])
        video = _video(seed, rec)
        runs = [_run(seed, rec, part, rec["iters"], video) for part in PARTITIONS]
        del video
        p_pre = np.array([r[0] for r in runs])
        pre_refs = [rec["psnr_pre"]] + ([arms2[seed][0]] if seed in arms2 else [])
        print("seed %d (%s flow, reference CPU time %.0f s pre-train + %.0f s loop, %d thread(s)%s): PSNR after the pre-train hip %s / reference %s dB"
              % (seed, rec["flow"], rec["cpu_seconds"][0], rec["cpu_seconds"][1], rec["threads"], "; second arm: %d" % arms2[seed][3] if seed in arms2 else "",
                 np.array2string(p_pre, precision=4), np.round(pre_refs, 4)))
        # 8000 pre-train steps on the same draws: the pre-train loss has ONE minimum (uv = 0.8 xy), both sides orbit it — and where on the orbit
        # step 8000 falls moves the PSNR of the (still random) atlas by up to 0.09 dB between this path's OWN partitions (seed 1: 16.856 .. 16.946)
        assert abs(float(p_pre.mean()) - float(np.mean(pre_refs))) <= 0.1 + 2.0 * 0.05 * np.sqrt(1.0 / len(pre_refs) + 1.0 / npart), (p_pre, pre_refs)
        d_pre.append(float(p_pre.mean() - np.mean(pre_refs)))
        curve = np.stack([r[3][::every][:n_logged, :6] for r in runs])          # (partition, logged iteration, term)
        ref = rec["curve"][:curve.shape[1]]
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(ref != 0, np.abs(curve / ref - 1.0), 0.0)
        print("   total loss every %d iterations, reference:        %s" % (every, np.array2string(ref[:, 5], precision=1, max_line_width=600)))
        for part, c in zip(PARTITIONS, curve):
            print("   total loss every %d iterations, hip %-19s %s" % (every, (part or "shipped partition") + ":", np.array2string(c[:, 5], precision=1, max_line_width=600)))
        print("   the six terms (rgb, gradient, rigidity, global rigidity, flow, total) at the last logged iteration, reference: %s" % np.array2string(ref[-1], precision=5, max_line_width=300))
        print("   ... hip, mean over the partitions:                                                                        %s" % np.array2string(curve[:, -1].mean(axis=0), precision=5, max_line_width=300))
        print("   distance from the reference, worst term per logged iteration (best partition): %s" % np.array2string(rel.max(axis=2).min(axis=0), precision=3, max_line_width=600))
        # iteration 0: the same batch on a state 8000 chaotic steps old.  The total is dominated by the global-rigidity term of a few rows; where on
        # its orbit around the pre-train minimum step 8000 falls moves it by up to 15 % between this path's OWN partitions (seed 0: 1041 / 1183 / 1218
        # against the reference's 1229, profiles/r5_pytest_c2_complete.log; 1390 in the f16x3 arithmetic) — one run is held to 25 %, ITER0_TOL
        assert rel[:, 0, 5].min() < ITER0_TOL, (curve[:, 0], ref[0])
        assert np.all((ref[:, 3] > 0) == (np.arange(len(ref)) * every <= 5000)) and np.all((curve[:, :, 3] > 0) == (ref[None, :, 3] > 0))   # the switch at 5000, both sides
        # along the curve: the total and the rgb term (what the PSNR is made of) stay within 15 % of the reference's at every logged iteration, the 5000
        # without global rigidity included (two runs of either side differ by 1-8 % there: profiles/r5_pytest_c2_complete.log)
        for t in (0, 5):                                                          # (iteration 0 has its own bound above)
            assert np.all(rel[:, 1:, t].min(axis=0) <= 0.15), (TERMS[t], rel[:, :, t].min(axis=0))
        for i in sorted(rec["psnr_at"]):
            h = np.array([r[1][i] for r in runs])
            refs = [rec["psnr_at"][i]] + ([arms2[seed][1][i]] if seed in arms2 and i in arms2[seed][1] else [])
            print("   PSNR after %d iterations: hip %s (mean %.4f) / reference %s dB" % (i, np.array2string(h, precision=4), h.mean(), np.round(refs, 4)))
            d_mid.append(float(h.mean() - np.mean(refs))); sig.append(h)
        h = np.array([r[2] for r in runs])
        refs = [rec["psnr_end"]] + ([arms2[seed][2]] if seed in arms2 else [])
        print("   PSNR after %d iterations: hip %s (mean %.4f) / reference %s dB" % (rec["iters"], np.array2string(h, precision=4), h.mean(), np.round(refs, 4)))
        d_end.append(float(h.mean() - np.mean(refs))); sig.append(h); narm.append(len(refs))
    # Run-to-run sigma by KIND OF VIDEO: the translating video is well conditioned, the field-flow one is not — two runs of the REFERENCE differ by
    # 0.01-0.06 dB on the first and 0.09-0.34 dB on the second, and so do two partitions of this path.  One pooled sigma would be too wide a net
    # for one kind and too narrow for the other.
    kind = np.array([recs[s]["flow"] for s in seeds])
    kinds = sorted(set(kind))
    pairs = {k: [] for k in kinds}
    for s2, (pre2, at2, end2, thr2) in sorted(arms2.items()):
        rec = recs[s2]
        pairs[rec["flow"]] += [end2 - rec["psnr_end"]] + [at2[i] - rec["psnr_at"][i] for i in at2 if i in rec["psnr_at"]]
        print("reference against itself, seed %d (%s flow, %d vs %d threads): PSNR after the pre-train %.4f / %.4f, after 5000 iterations %s / %s, at the end %.4f / %.4f dB"
              % (s2, rec["flow"], thr2, rec.get("threads", -1), pre2, rec["psnr_pre"], [round(v, 4) for v in at2.values()], [round(rec["psnr_at"][i], 4) for i in at2 if i in rec["psnr_at"]], end2, rec["psnr_end"]))
    assert len(arms2) >= 4 and all(len(pairs[k]) >= 4 for k in kinds), "the reference's sigma at this size needs its own pairs: >= 4 seeds with a second arm, two of each kind of video"
    sigma_ref = {k: float(np.sqrt(np.mean(np.square(pairs[k])) / 2.0)) for k in kinds}
    # this side: measured over the partitions when they were run (sig holds, per seed, the evaluation at the switch then the one at the end), else the recorded figures
    sig_kind = np.repeat(kind, len(sig) // len(seeds))
    sigma = {k: float(np.sqrt(np.mean([np.var(h, ddof=1) for h, kk in zip(sig, sig_kind) if kk == k]))) for k in kinds} if npart > 1 else dict(SIGMA_HIP_RECORDED)
    for k in kinds:
        print("run-to-run sigma on the %s-flow video: reference %.3f dB (from %d paired evaluations of its own arms) ; this side %.3f dB (%s)"
              % (k, sigma_ref[k], len(pairs[k]), sigma[k], "pooled over %d partitions" % npart if npart > 1 else "recorded, profiles/r6_pytest_c2_all_partitions.log"))
    narm = np.array(narm, np.float64)
    for name, dd in (("after the pre-train", d_pre), ("after 5000 iterations", d_mid), ("at the end", d_end)):
        d = np.array(dd)
        se = float(d.std(ddof=1) / np.sqrt(len(d)))
        print("hip - reference %s (a seed with two reference arms: against their mean): per seed %s dB ; mean %+.4f dB, standard error over seeds %.4f dB (n = %d)"
              % (name, np.array2string(d, precision=4), d.mean(), se, len(d)))
        if name == "after the pre-train":
            continue
        for k in kinds:
            dk = d[kind == k]
            sek = float(dk.std(ddof=1) / np.sqrt(len(dk)))
            print("   %s-flow videos alone: mean %+.4f dB, standard error %.4f dB (n = %d)" % (k, dk.mean(), sek, len(dk)))
            assert abs(float(dk.mean())) <= 0.1 + 2.0 * sek, (name, k, float(dk.mean()), sek)
        # per seed, a net for gross failures: BASELINE.md's 0.1 dB on top of FOUR standard deviations of the difference between one HIP run (the mean of
        # npart) and the mean of that seed's reference arms, sigmas of the seed's kind of video.  Two until round 6 — 16 such checks at two sigmas fail one
        # run in two by chance alone, and a pooled sigma is itself an average over videos: seed 6 (field flow) after 5000 iterations has a run-to-run sigma
        # of 0.25 dB on this side over nine summation orders (25.55 .. 26.36 dB, profiles/r6_c2_seed6_spread.txt; the reference's two arms: 26.33, 26.24),
        # against the pooled 0.16, and its shipped partition is the lowest of the nine.  The statements about parity are the means below, not this net.
        tol_seed = 0.1 + 4.0 * np.sqrt(np.array([sigma_ref[k] for k in kind]) ** 2 / narm + np.array([sigma[k] for k in kind]) ** 2 / npart)
        se_max = SE_MAX[name]
        print("   tolerances: per seed %s dB ; on the mean 0.1 + 2 SE = %.3f dB with SE <= %.3f required" % (np.round(tol_seed, 3), 0.1 + 2.0 * se, se_max))
        assert np.all(np.abs(d) <= tol_seed), (name, d, tol_seed)
        assert len(d) >= 8 and se <= se_max, (name, len(d), se)                        # the comparison resolves what it claims to
        assert abs(float(d.mean())) <= 0.1 + 2.0 * se, (name, float(d.mean()), se)
