"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures
generated from the reference's own modules (oracle/make_golden.py).  Tolerances follow BASELINE.json:
loss terms within 1e-3 relative, PSNR within 0.1 dB; forward outputs are checked much tighter."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(golden, **over):
    import aiod_amd
    c = dict(golden["config"])
    c.update(over)
    return aiod_amd.default_config(int(golden["resx"]), int(golden["resy"]), int(golden["nframes"]), c)


def _upload(af, v):
    af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask)


def _oracle_models(golden, start=True):
    from oracle import atlas_oracle as O
    m, a = O.build_single_atlas_models(golden["config"], seed=int(golden["weight_seed"]))
    assert abs(float(np.abs(O.flat_params(m)).sum()) - float(golden["init_checksum"][0])) < 1e-3
    if start:
        _load_flat(m, golden["start_map"]); _load_flat(a, golden["start_atlas"])
    return m, a


def _load_flat(model, flat):
    off = 0
    with torch.no_grad():
        for p in model.parameters():
            n = p.numel()
            p.copy_(torch.from_numpy(flat[off:off + n].reshape(p.shape)))
            off += n
    assert off == flat.size


@pytest.fixture(scope="module")
def af(golden, small_video):
    import aiod_amd
    h = aiod_amd.AtlasFit(_cfg(golden, pretrain_batch=int(golden["pre_batch"])))
    _upload(h, small_video)
    yield h
    h.close()


@pytest.fixture(scope="module", params=["constant", "field"])
def case(request, af):
    """(handle, fixture, video) on the translating video and on the video whose flow differs at every pixel of every frame and whose
    masks have holes (round 4: `oracle/make_golden.py field` -> single_field.npz, the reference's modules on that video)."""
    if request.param == "constant":
        yield af, request.getfixturevalue("golden"), request.getfixturevalue("small_video")
        return
    import aiod_amd
    g, v = request.getfixturevalue("golden_field"), request.getfixturevalue("small_video_field")
    h = aiod_amd.AtlasFit(_cfg(g, pretrain_batch=int(g["pre_batch"])))
    _upload(h, v)
    yield h, g, v
    h.close()


def test_forward_matches_reference_imlp(af, golden):
    import aiod_amd
    m, a = _oracle_models(golden, start=False)
    af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict())
    af.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
    rows = np.zeros((96, 4), np.float32); rows[:, :3] = golden["rows_xyt"]
    out = af.debug_forward(aiod_amd.NET_MAPPING1, rows)
    err = np.abs(out[:, :2] - golden["fwd_map"]).max()
    assert err < 2e-6, err
    rows = np.zeros((96, 4), np.float32); rows[:, :2] = golden["rows_uv"]
    out = af.debug_forward(aiod_amd.NET_ATLAS, rows)
    err = np.abs(out[:, :3] - golden["fwd_atlas"]).max()
    assert err < 5e-6, err
    # round trip of the parameter views
    sd = af.state_dict(aiod_amd.NET_ATLAS)
    for k, v in a.state_dict().items():
        assert np.array_equal(sd[k], v.numpy()), k


def test_forward_ragged_rows(af, golden):
    """row counts that are not multiples of the 32-row tile / 128-row workgroup"""
    import aiod_amd
    m, a = _oracle_models(golden, start=True)
    af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict())
    g = torch.Generator().manual_seed(11)
    for n in (1, 31, 33, 129, 1000):
        x = torch.rand(n, 3, generator=g) * 2 - 1
        rows = np.zeros((n, 4), np.float32); rows[:, :3] = x.numpy()
        out = af.debug_forward(aiod_amd.NET_MAPPING1, rows)
        with torch.no_grad():
            ref = m(x).numpy()
        assert np.abs(out[:, :2] - ref).max() < 2e-6, n


def test_single_step_losses_and_gradients(case):
    af, golden, small_video = case
    import aiod_amd
    from oracle import atlas_oracle as O
    m, a = _oracle_models(golden)
    tr = O.SingleAtlasTrainer(golden["config"], small_video, mapping=m, atlas=a)
    inds = torch.from_numpy(golden["inds"][0].astype(np.int64))
    terms = tr.loss_and_grads(0, inds)
    gm, ga = O.flat_grads(m), O.flat_grads(a)
    af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict())
    af.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
    af.set_adam_state(aiod_amd.NET_MAPPING1, np.zeros_like(gm), np.zeros_like(gm), 0)
    af.set_adam_state(aiod_amd.NET_ATLAS, np.zeros_like(ga), np.zeros_like(ga), 0)
    af.set_debug(True)
    losses = af.train_steps(0, 1, inds.numpy())[0]
    ref = np.array([terms[k] for k in ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total")])
    print("hip losses", losses[:6], "oracle", ref, "golden", golden["losses"][0])
    assert np.allclose(losses[:6], ref, rtol=1e-4), (losses, ref)
    assert np.allclose(losses[:6], golden["losses"][0], rtol=1e-3)
    hm, ha = af.last_grads(aiod_amd.NET_MAPPING1), af.last_grads(aiod_amd.NET_ATLAS)
    for name, hg, og in (("mapping", hm, gm), ("atlas", ha, ga)):
        rel = np.linalg.norm(hg - og) / np.linalg.norm(og)
        print(name, "grad rel err", rel, "norm", np.linalg.norm(og))
        assert rel < 3e-4, (name, rel)          # against torch-fp32, whose own atlas gradient is ~1e-3 from an fp64 twin on this state (per-layer check below)
    assert abs(np.linalg.norm(hm) / float(golden["grads0_map_norm"]) - 1) < 1e-3
    assert abs(np.linalg.norm(ha) / float(golden["grads0_atlas_norm"]) - 1) < 1e-3
    # per-layer check so a single broken layer is named.  Yardstick: an fp64 twin of the oracle — on this
    # un-pre-trained, badly conditioned state (rigidity ~1.2e3, PE frequencies up to 2^9 pi in atlas layer 0) the
    # reference's own fp32 gradient carries per-layer errors of ~1e-3; HIP may be no further from fp64 than 3x that.
    import copy
    m64, a64 = copy.deepcopy(m).double(), copy.deepcopy(a).double()
    a64.b = a64.b.double()
    v = small_video
    v64 = O.Video(v.video_frames.double(), v.optical_flows.double(), v.optical_flows_reverse.double(), v.optical_flows_mask, v.optical_flows_reverse_mask)
    tr64 = O.SingleAtlasTrainer(golden["config"], v64, mapping=m64, atlas=a64)
    torch.set_default_dtype(torch.float64)
    try:
        tr64.loss_and_grads(0, inds)
    finally:
        torch.set_default_dtype(torch.float32)
    ga64 = O.flat_grads(a64)
    off = 0
    for i, (o, k) in enumerate(aiod_amd.atlasfit.imlp_shapes(aiod_amd.NET_ATLAS)):
        for nm, cnt in (("weight", o * k), ("bias", o)):
            n64 = np.linalg.norm(ga64[off:off + cnt]) + 1e-30
            e_hip = np.linalg.norm(ha[off:off + cnt] - ga64[off:off + cnt]) / n64
            e_ref = np.linalg.norm(ga[off:off + cnt] - ga64[off:off + cnt]) / n64
            assert e_hip < max(3 * e_ref, 1e-3), ("atlas", i, nm, e_hip, e_ref)
            off += cnt
    af.set_debug(False)


def test_trajectory_matches_reference(case):
    """K iterations from the reference's post-pre-train state with the reference's index stream:
    loss terms within 1e-3 relative (BASELINE.json), end weights close, PSNR within 0.1 dB."""
    af, golden, small_video = case
    import aiod_amd
    m, a = _oracle_models(golden)
    af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict())
    af.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
    z = np.zeros(af.param_count(aiod_amd.NET_MAPPING1), np.float32)
    af.set_adam_state(aiod_amd.NET_MAPPING1, z, z, 0)
    z = np.zeros(af.param_count(aiod_amd.NET_ATLAS), np.float32)
    af.set_adam_state(aiod_amd.NET_ATLAS, z, z, 0)
    inds = golden["inds"].astype(np.int64)
    K = inds.shape[0]
    losses = af.train_steps(0, K, inds)
    ref = golden["losses"]
    for i in range(K):
        rel = np.abs(losses[i, :6] - ref[i]) / np.maximum(np.abs(ref[i]), 1e-12)
        rel[3] = 0 if ref[i][3] == 0 and losses[i, 3] == 0 else rel[3]
        print(i, "rel", rel)
        assert rel.max() < 1e-3, (i, losses[i], ref[i])
    end_m = af.get_params_flat(aiod_amd.NET_MAPPING1)[::97]
    end_a = af.get_params_flat(aiod_amd.NET_ATLAS)[::97]
    # Adam divides by sqrt(v): for near-zero gradients fp32 rounding differences move a weight by up to lr per
    # step, so after K steps weights may differ by a fraction of K*lr = 1e-3 while the losses still agree
    dm, da = np.abs(end_m - golden["end_map_sample"]), np.abs(end_a - golden["end_atlas_sample"])
    print("end-weight diff: mapping max %.3g mean %.3g, atlas max %.3g mean %.3g" % (dm.max(), dm.mean(), da.max(), da.mean()))
    assert dm.max() < 1e-3 and da.max() < 1e-3
    assert dm.mean() < 3e-5 and da.mean() < 3e-5
    mean, per = af.psnr()
    print("psnr", mean, float(golden["psnr"]))
    assert abs(mean - float(golden["psnr"])) < 0.1
    # render vs the oracle render with the HIP-trained weights
    from oracle import atlas_oracle as O
    _load_flat(m, af.get_params_flat(aiod_amd.NET_MAPPING1)); _load_flat(a, af.get_params_flat(aiod_amd.NET_ATLAS))
    rec, _ = af.render_frame(2)
    oref = O.render_frame(m, a, small_video.resx, small_video.resy, small_video.F, 2).numpy()
    assert np.abs(rec - oref).max() < 1e-4


def test_render_matches_oracle_per_pixel(case):
    """af_render_frame against the oracle's restatement of evaluate.py:640-661, pixel by pixel (the render was
    pinned only transitively — forward vs IMLP, PSNR after the pre-train vs the reference): every pixel of the first, a middle and the last
    frame no further from an fp64 twin of the oracle than twice what the oracle's own fp32 render is (floor 2e-6), the
    per-frame SSE the library accumulates in fp64 against the fp64 SSE of the oracle's image, and af_psnr against the oracle's mean PSNR;
    on the translating and on the field-flow video."""
    from oracle import atlas_oracle as O
    import aiod_amd
    h, g, v = case
    m, a = _oracle_models(g)
    h.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict()); h.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
    import copy
    m64, a64 = copy.deepcopy(m).double(), copy.deepcopy(a).double()
    a64.b = a64.b.double()
    worst = 0.0
    for f in (0, v.F // 2, v.F - 1):
        want = O.render_frame(m, a, v.resx, v.resy, v.F, f).numpy()
        torch.set_default_dtype(torch.float64)          # the oracle's coordinate grid follows the default dtype
        try:
            want64 = O.render_frame(m64, a64, v.resx, v.resy, v.F, f).numpy()
        finally:
            torch.set_default_dtype(torch.float32)
        got, sse = h.render_frame(f)
        assert got.shape == want.shape == (v.resy, v.resx, 3)
        d, e_ref, e_hip = float(np.abs(got - want).max()), float(np.abs(want - want64).max()), float(np.abs(got - want64).max())
        worst = max(worst, d)
        # the mapping net's last-ulp round-off reaches the atlas through Fourier features up to 2^9 pi (x800 in phase): two fp32
        # implementations differ by what either differs from exact arithmetic, so THAT is the yardstick, measured per frame
        assert e_hip <= max(2e-6, 2.0 * e_ref) and d <= 2e-6 + 2.0 * e_ref, (f, d, e_hip, e_ref)
        gt = v.video_frames[:, :, :, f].numpy().astype(np.float64)
        sse_want = float(((want.astype(np.float64) - gt) ** 2).sum())
        assert abs(sse - sse_want) <= 1e-5 * sse_want, (f, sse, sse_want)
    mean_want, per_want = O.mean_psnr(m, a, v)
    mean_got, per_got = h.psnr()
    print("render vs oracle: worst pixel %.3g; PSNR %.6f vs %.6f dB" % (worst, mean_got, mean_want))
    assert np.abs(per_got - np.array(per_want)).max() <= 1e-4 and abs(mean_got - mean_want) <= 1e-4


def test_pretrain_matches_reference(af, golden):
    import aiod_amd
    m, a = _oracle_models(golden, start=False)
    af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict())
    ys, xs = golden["pre_ys"].astype(np.int64), golden["pre_xs"].astype(np.int64)
    losses = af.pre_train_mapping(2, ys, xs, return_losses=True)
    ref = golden["pre_losses"]
    print("pretrain", losses[:4], ref[:4])
    assert np.allclose(losses, ref, rtol=1e-4)
    p = af.get_params_flat(aiod_amd.NET_MAPPING1)[::97]
    assert np.abs(p - golden["pre_params_sample"]).max() < 1e-5


def test_device_sampler_runs_and_is_deterministic(af, golden):
    import aiod_amd
    m, a = _oracle_models(golden)
    outs = []
    for _ in range(2):
        af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict())
        af.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
        z = np.zeros(af.param_count(aiod_amd.NET_MAPPING1), np.float32); af.set_adam_state(aiod_amd.NET_MAPPING1, z, z, 0)
        z = np.zeros(af.param_count(aiod_amd.NET_ATLAS), np.float32); af.set_adam_state(aiod_amd.NET_ATLAS, z, z, 0)
        outs.append(af.train_steps(0, 4, None, seed=99))
    assert np.array_equal(outs[0], outs[1])          # fixed-order reductions: bit-reproducible
    assert np.isfinite(outs[0]).all()
    # the sampled batch behaves like the oracle's on the same kind of draw (loose statistical check)
    assert abs(outs[0][0, 0] / float(golden["losses"][0][0]) - 1) < 0.3


def test_error_paths(golden):
    import aiod_amd
    with pytest.raises(aiod_amd.AtlasFitError):
        aiod_amd.AtlasFit(_cfg(golden, number_of_channels_atlas=512))      # widths above 256 are not built (1..256 run zero-padded since round 5)
    h = aiod_amd.AtlasFit(_cfg(golden))
    with pytest.raises(aiod_amd.AtlasFitError):
        h.train_steps(0, 1)                            # no video uploaded
    with pytest.raises(aiod_amd.AtlasFitError):
        h.lib.af_set_params(h.h, 0, None, 5) and None
        h._chk(h.lib.af_set_params(h.h, 0, None, 5))
    h.close()


def test_end_to_end_schedule_psnr_parity(small_video, golden):
    """The acceptance criterion of BASELINE.json on a whole (short) schedule: the reference's construction-order init,
    pre_train_mapping, then 60 loop iterations crossing the global-rigidity switch — the SAME draws fed to the CPU
    oracle and to the HIP path — must end within 0.1 dB of each other in reconstruction PSNR.  The two fp32 trajectories decorrelate
    the way any two fp32 implementations of this loop do (measured here: total loss 5e-5 apart over the first ten
    iterations, 4e-2 by iteration 100, where the PSNRs differ by 0.13 dB), hence a schedule of 60 iterations."""
    import aiod_amd
    from oracle import atlas_oracle as O
    cfg = dict(golden["config"]); cfg.update(samples_batch=512, stop_global_rigidity=30)
    v = small_video
    m, a = O.build_single_atlas_models(cfg, seed=21)
    h = aiod_amd.AtlasFit(aiod_amd.default_config(v.resx, v.resy, v.F, cfg, pretrain_batch=512))
    _upload(h, v)
    h.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict()); h.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
    g = torch.Generator().manual_seed(8)
    iters_pre, K = 10, 60
    ys = torch.randint(v.resy, (iters_pre * v.F, 512), generator=g); xs = torch.randint(v.resx, (iters_pre * v.F, 512), generator=g)
    inds = torch.randint(v.F * v.resx * v.resy, (K, 512), generator=g)
    pl_o = O.pre_train_mapping(m, v.F, cfg["uv_mapping_scale"], v.resx, v.resy, v.larger_dim, iters_pre, ys, xs, batch=512)
    pl_h = h.pre_train_mapping(iters_pre, ys.numpy(), xs.numpy(), return_losses=True)
    pl_o = np.array(pl_o)
    prel = np.abs(pl_h - pl_o) / pl_o
    print("pre-train loss rel dev: first20 %.2e  max %.2e ; last-20 means %.5f / %.5f" % (prel[:20].max(), prel.max(), pl_h[-20:].mean(), pl_o[-20:].mean()))
    assert prel[:20].max() < 1e-3 and abs(pl_h[-20:].mean() / pl_o[-20:].mean() - 1) < 0.1
    tr = O.SingleAtlasTrainer(cfg, v, mapping=m, atlas=a)
    ref = np.array([[t[k] for k in ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total")]
                    for t in (tr.step(i, inds[i]) for i in range(K))])
    got = h.train_steps(0, K, inds.numpy())
    p_ref, _ = O.mean_psnr(m, a, v)
    p_hip, _ = h.psnr()
    rel = np.abs(got[:, 5] - ref[:, 5]) / ref[:, 5]
    print("psnr oracle %.4f hip %.4f | total-loss rel dev: first10 %.2e  last10 %.2e  max %.2e" % (p_ref, p_hip, rel[:10].max(), rel[-10:].max(), rel.max()))
    assert ref[0, 2] < 100 and abs(got[0, 2] / ref[0, 2] - 1) < 1e-2         # pre-trained (un-pre-trained: ~1300), same on both sides
    assert rel[:10].max() < 1e-2 and rel.max() < 0.1
    assert abs(p_hip - p_ref) < 0.1, (p_hip, p_ref)
    assert p_ref > 14.0 and ref[-1, 5] < 0.8 * ref[0, 5]                     # the schedule did fit something
    h.close()


def test_more_error_paths(golden, small_video):
    """Every entry point answers misuse with a negative status and a message, never a crash (include/atlasfit.h)."""
    import ctypes
    import aiod_amd
    h = aiod_amd.AtlasFit(_cfg(golden))
    lib = h.lib
    bad = np.zeros(5, np.float32)
    assert lib.af_set_params(h.h, aiod_amd.NET_ATLAS, bad.ctypes.data_as(ctypes.c_void_p), 5) == -1 and b"count" in lib.af_last_error(h.h)
    assert lib.af_set_params(h.h, aiod_amd.NET_ALPHA, bad.ctypes.data_as(ctypes.c_void_p), 5) == -1          # net not part of a single-atlas handle
    assert lib.af_param_count(h.h, aiod_amd.NET_MAPPING2) == 0 and lib.af_loss_width(h.h) == 8
    assert lib.af_pretrain(h.h, aiod_amd.NET_ATLAS, 1, None, None, 0, None) == -1                               # only mapping nets pre-train
    assert lib.af_pretrain(h.h, aiod_amd.NET_MAPPING2, 1, None, None, 0, None) == -1
    assert lib.af_render_frame(h.h, 0, None, None) == -5                                                        # AF_ESTATE: no video yet
    _upload(h, small_video)
    assert lib.af_render_frame(h.h, int(golden["nframes"]), None, None) == -1
    assert lib.af_train_steps(h.h, -1, 1, None, 0, None) == -1 and lib.af_train_steps(h.h, 0, 0, None, 0, None) == 0
    idx = np.array([int(golden["nframes"]) * int(golden["resx"]) * int(golden["resy"])], np.int64)             # one past the last record
    out = np.zeros((1, 16), np.float32)
    assert lib.af_debug_records(h.h, idx.ctypes.data_as(ctypes.c_void_p), 1, out.ctypes.data_as(ctypes.c_void_p)) == -1
    assert lib.af_resize_bilinear(0, None, 0, 4, 4, 3, None, 2, 2, 3, 1, 0, 1.0, 1.0, 0) == -1
    assert lib.af_flow_consistency(0, None, None, 4, 4, None, 1, 0, 1.0, 0) == -1
    assert lib.af_set_dw_mode(h.h, 3) == -1 and lib.af_set_mlp_mode(h.h, -1) == -1                              # the arithmetic switches know 0, 1 (and 2 for k_dw)
    assert lib.af_set_dw_mode(h.h, 0) == 0 and lib.af_set_dw_mode(h.h, 1) == 0 and lib.af_set_dw_mode(h.h, 2) == 0   # re-cuts the dW schedules every way
    assert lib.af_debug_dw_schedule(h.h, 7, None, 0) < 0                                                         # no such schedule
    l = h.train_steps(0, 1, None, seed=1)                                                                        # and the handle still trains
    assert np.isfinite(l).all()
    h.close()


def test_step_clocks_stamp_every_hot_launch(af, golden):
    """af_debug_step_clocks (include/atlasfit.h): with the stamps on, the five hot launches of a step record s_memrealtime / s_memtime per
    workgroup at its start and end; ticks over the 100 MHz span is a shader clock (tools/step_clock.py reads it inside the bench loop)."""
    import aiod_amd
    m, a = _oracle_models(golden)
    af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict()); af.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
    first = af.step_clocks(True)
    assert set(first) == set(af.STEP_CLOCK_LAUNCHES) and all(len(v) == 0 for v in first.values())      # nothing stamped yet: zeros, not garbage
    af.train_steps(0, 3, None, seed=5, return_losses=False)
    st = af.step_clocks(False)                                                                          # read and switch off
    for name in ("fwd_2", "bwd_1", "dw"):                                                               # (a small batch has no whole round of mapping tiles: fwd_1 / bwd_2 may be empty)
        c = st[name].astype(np.float64)
        assert len(c) >= 1, name
        assert (c[:, 2] >= c[:, 0]).all() and (c[:, 3] > c[:, 1]).all(), name
        span_us = (c[:, 2] - c[:, 0]) / 100.0
        busy = span_us > 2.0
        if busy.any():
            mhz = (c[busy, 3] - c[busy, 1]) / span_us[busy]
            assert (mhz > 300).all() and (mhz < 3000).all(), (name, mhz)
    af.train_steps(0, 1, None, seed=5, return_losses=False)
    assert all(len(v) == 0 for v in af.step_clocks(False).values())                                     # off: no stamps


def test_event_timing_sample_period(golden, small_video):
    """af_set_timing's sample period (bits 16..23 of the mask; include/atlasfit.h): with period P only every P-th step of an
    af_train_steps call carries HIP events, counts and FLOPs cover exactly those launches, and the results do not depend on it."""
    import aiod_amd
    m, a = _oracle_models(golden)

    def run(every):
        h = aiod_amd.AtlasFit(_cfg(golden, pretrain_batch=int(golden["pre_batch"])))          # a fresh handle: fresh Adam state
        _upload(h, small_video)
        h.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict()); h.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
        h.set_timing(0xFFFF, every=every)
        losses = h.train_steps(0, 10, None, seed=5, return_losses=True)
        t = h.timing(reset=True)
        h.set_timing(0)
        h.train_steps(10, 2, None, seed=5, return_losses=False)
        off = h.timing(reset=True)["dw"][1]
        h.close()
        return losses, t, off
    l1, t1, off1 = run(1)
    l4, t4, off4 = run(4)
    assert t1["dw"][1] == 10 and t1["adam"][1] == 10 and t1["fwd_2"][1] == 10
    assert t4["dw"][1] == 3 and t4["adam"][1] == 3 and t4["fwd_2"][1] == 3            # steps 0, 4, 8 of the call
    assert t4["dw"][0] > 0 and 0.2 * t1["dw"][2] < t4["dw"][2] < 0.4 * t1["dw"][2]    # FLOPs of the three timed launches only (the fixture's schedule switches regime inside the ten steps)
    assert np.array_equal(l1, l4)                                                    # timing never changes results
    assert off1 == 0 and off4 == 0                                                   # mask 0: no events
