"""Host-side mirror of the reference's stage-1 script: input builder on CPU, CLI + results tree + checkpoint on GPU."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_video(tmp, v, name="vid"):
    from PIL import Image
    d = tmp / name; d.mkdir(parents=True); fd = tmp / (name + "_flow"); fd.mkdir()
    F = v.F
    names = []
    for f in range(F):
        fn = "%05d.png" % f; names.append(fn)
        Image.fromarray(np.round(v.video_frames[:, :, :, f].numpy() * 255).astype(np.uint8)).save(str(d / fn))
    for f in range(F - 1):
        np.save(fd / ("%s_%s.npy" % (names[f], names[f + 1])), v.optical_flows[:, :, :, f, 0].numpy())
        np.save(fd / ("%s_%s.npy" % (names[f + 1], names[f])), v.optical_flows_reverse[:, :, :, f + 1, 0].numpy())
    return d


def test_resize_bilinear_matches_cv2_convention():
    import aiod_amd.stage1 as S
    img = np.arange(16, dtype=np.float64).reshape(4, 4)
    out = S.resize_bilinear(img, 2, 2)           # centres at 0.5, 2.5 -> averages of 2x2 blocks
    assert np.allclose(out, [[2.5, 4.5], [10.5, 12.5]])
    up = S.resize_bilinear(np.array([[0.0, 1.0]]), 4, 1)   # cv2: [0, 0.25, 0.75, 1]
    assert np.allclose(up, [[0.0, 0.25, 0.75, 1.0]])
    assert np.array_equal(S.resize_bilinear(img, 4, 4), img)


def test_input_builder_reproduces_reference_tensors(tmp_path, small_video):
    import aiod_amd.stage1 as S
    v = small_video
    d = _write_video(tmp_path, v)
    m, frames, mr, fr, fl = S.load_input_data_single(v.resy, v.resx, 200, d, True, tmp_path, "vid")
    assert frames.shape == (v.resy, v.resx, 3, v.F)
    assert np.abs(frames - np.round(v.video_frames.numpy() * 255) / 255).max() < 1e-7      # PNG quantisation only
    assert np.array_equal(fl, v.optical_flows.numpy()) and np.array_equal(fr, v.optical_flows_reverse.numpy())
    assert np.array_equal(m, v.optical_flows_mask.numpy()) and np.array_equal(mr, v.optical_flows_reverse_mask.numpy())
    # flows stored at a different resolution are rescaled the reference's way (u by newh/oldh, v by neww/oldw)
    big = np.ones((2 * v.resy, 2 * v.resx, 2), np.float32)
    r = S.resize_flow(big, v.resy, v.resx)
    assert r.shape == (v.resy, v.resx, 2) and np.allclose(r, 0.5)


@pytest.mark.gpu
def test_cli_results_tree_checkpoint_and_resume(tmp_path, small_video, golden, monkeypatch):
    import torch
    import aiod_amd
    import aiod_amd.stage1 as S
    from oracle import atlas_oracle as O
    v = small_video
    _write_video(tmp_path / "data", v, "clip")
    cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
    cfg.update(samples_batch=256, iters_num=41, evaluate_every=20, pretrain_iter_number=2, stop_global_rigidity=10)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    monkeypatch.chdir(tmp_path)
    psnr = S._cli(["--config", str(tmp_path / "cfg.json"), "--vid_name", "clip", "--root", str(tmp_path / "data"), "--down", "1", "--seed", "5"])
    res = tmp_path / "results" / "clip" / "stage_1"
    assert (res / "config.json").exists() and (res / "checkpoint").exists()
    outs = sorted((res / "output").glob("*.png"))
    assert [p.name for p in outs] == ["%05d.png" % f for f in range(v.F)]
    marks = list((res / "000040").glob("PSNR_*")) + list((res / "000020").glob("PSNR_*"))
    assert len(marks) == 2 and abs(float(marks[0].name[5:]) - psnr) < 1e-3
    # the checkpoint is a genuine reference-format file: keys, IMLP-compatible state dicts, Adam state
    ck = torch.load(res / "checkpoint", map_location="cpu", weights_only=False)
    assert set(ck) == {"F_atlas_state_dict", "iteration", "model_F_mapping1_state_dict", "optimizer_all_state_dict"} and ck["iteration"] == 40
    m, a = O.build_single_atlas_models(cfg, seed=0)
    m.load_state_dict(ck["model_F_mapping1_state_dict"]); a.load_state_dict(ck["F_atlas_state_dict"])
    opt = torch.optim.Adam([{"params": list(m.parameters())}, {"params": list(a.parameters())}], lr=1e-4)
    opt.load_state_dict(ck["optimizer_all_state_dict"])
    assert int(opt.state_dict()["state"][0]["step"]) == 41
    # the PNGs are the truncated render of the checkpointed weights (evaluate.py:733)
    from PIL import Image
    rec = O.render_frame(m, a, v.resx, v.resy, v.F, 1).numpy()
    png = np.array(Image.open(outs[1])).astype(np.int32)
    assert np.abs(png - (rec.astype(np.float64) * 255).astype(np.uint8)).max() <= 1
    # resume: load the checkpoint into a fresh handle and continue
    af = aiod_amd.AtlasFit(aiod_amd.default_config(v.resx, v.resy, v.F, cfg))
    it = S.load_checkpoint(af, res / "checkpoint")
    assert it == 40 and af.adam_state(aiod_amd.NET_ATLAS)[2] == 41
    assert np.array_equal(af.state_dict(aiod_amd.NET_ATLAS)["hidden.3.weight"], ck["F_atlas_state_dict"]["hidden.3.weight"].numpy())
    af.close()


@pytest.mark.gpu
def test_cli_with_a_missing_flow_file_fails_cleanly(tmp_path, small_video, monkeypatch):
    """Round 5: the CLI starts pre_train_mapping on its own thread before the clip is read.  An input error in the main thread (here: one flow
    file removed) must come out as the loader's FileNotFoundError with the pre-train thread joined and the handle closed — not as a crash of a
    thread that lost its handle — and the process must be able to go on using the GPU."""
    import aiod_amd
    import aiod_amd.stage1 as S
    v = small_video
    _write_video(tmp_path / "data", v, "clip")
    victim = sorted((tmp_path / "data" / "clip_flow").glob("*.npy"))[3]
    victim.unlink()
    cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
    cfg.update(samples_batch=256, iters_num=5, evaluate_every=4, pretrain_iter_number=20)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match="optical flow"):
        S._cli(["--config", str(tmp_path / "cfg.json"), "--vid_name", "clip", "--root", str(tmp_path / "data"), "--down", "1", "--seed", "5"])
    af = aiod_amd.AtlasFit(aiod_amd.default_config(v.resx, v.resy, v.F, cfg))      # the device is still usable
    af.pre_train_mapping(1, seed=3)
    af.close()


def _write_masks(tmp, v, name):
    from PIL import Image
    d = tmp / (name + "_seg"); d.mkdir()
    for f in range(v.F):
        Image.fromarray(np.round(v.mask_frames[:, :, f].numpy() * 255).astype(np.uint8)).save(str(d / ("%05d.png" % f)))


def test_mask_loader_reproduces_reference_tensor(tmp_path, small_seg_video):
    import aiod_amd.stage1 as S
    v = small_seg_video
    _write_video(tmp_path, v)
    _write_masks(tmp_path, v, "vid")
    m = S.load_mask_frames(v.resy, v.resx, v.F, tmp_path, "vid")
    assert m.shape == (v.resy, v.resx, v.F)
    assert np.abs(m - np.round(v.mask_frames.numpy() * 255) / 255).max() < 1e-7
    half = S.load_mask_frames(v.resy // 2, v.resx // 2, v.F, tmp_path, "vid")      # bilinear (fractional), not nearest
    assert half.shape == (v.resy // 2, v.resx // 2, v.F) and ((half > 0.02) & (half < 0.98)).any()
    with pytest.raises(FileNotFoundError):
        S.load_mask_frames(v.resy, v.resx, v.F, tmp_path, "missing")


@pytest.mark.gpu
def test_two_layer_cli_results_tree_checkpoint_and_resume(tmp_path, small_seg_video, monkeypatch):
    """stage1_neural_atlas_seg.py drop-in: four-net checkpoint in the reference's format, frames, PSNR, resume."""
    import torch
    import aiod_amd
    import aiod_amd.stage1 as S
    from oracle import atlas_oracle as O
    v = small_seg_video
    _write_video(tmp_path / "data", v, "clip")
    _write_masks(tmp_path / "data", v, "clip")
    cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
    cfg.update(samples_batch=256, iters_num=21, evaluate_every=20, pretrain_iter_number=2, stop_global_rigidity=10, stop_bootstrapping_iteration=15)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    monkeypatch.chdir(tmp_path)
    psnr = S._cli(["--config", str(tmp_path / "cfg.json"), "--vid_name", "clip", "--root", str(tmp_path / "data"), "--seed", "5"], two_layer=True)
    res = tmp_path / "results" / "clip" / "stage_1"
    assert (res / "checkpoint").exists() and (res / "000020" / "checkpoint").exists()       # evaluate.py:215-232 writes both
    assert len(list((res / "output").glob("*.png"))) == v.F and len(list((res / "000020").glob("PSNR_*"))) == 1
    ck = torch.load(res / "checkpoint", map_location="cpu", weights_only=False)
    assert set(ck) == {"F_atlas_state_dict", "iteration", "model_F_mapping1_state_dict", "model_F_mapping2_state_dict",
                       "model_F_alpha_state_dict", "optimizer_all_state_dict"} and ck["iteration"] == 20
    m1, m2, at, al = O.build_seg_models(cfg, seed=0)
    m1.load_state_dict(ck["model_F_mapping1_state_dict"]); m2.load_state_dict(ck["model_F_mapping2_state_dict"])
    at.load_state_dict(ck["F_atlas_state_dict"]); al.load_state_dict(ck["model_F_alpha_state_dict"])
    opt = torch.optim.Adam([{"params": list(m.parameters())} for m in (m1, m2, al, at)], lr=1e-4)      # reference group order
    opt.load_state_dict(ck["optimizer_all_state_dict"])
    assert int(opt.state_dict()["state"][0]["step"]) == 21
    mean, _ = O.mean_psnr_seg(m1, m2, at, al, v)               # oracle render of the checkpointed weights vs the PNG-quantised input
    assert abs(mean - psnr) < 0.1
    af = aiod_amd.AtlasFit(aiod_amd.default_config(v.resx, v.resy, v.F, cfg, two_layer=True))
    assert S.load_checkpoint(af, res / "checkpoint") == 20 and af.adam_state(aiod_amd.NET_ALPHA)[2] == 21
    assert np.array_equal(af.state_dict(aiod_amd.NET_ALPHA)["hidden.0.weight"], ck["model_F_alpha_state_dict"]["hidden.0.weight"].numpy())
    af.close()


def _write_builder_case(tmp_path, v, seed=2):
    """Frames and masks stored at twice the working resolution (so the bilinear resize runs), flows stored at another
    resolution (resize_flow's rescaling) and perturbed so that the consistency mask is non-trivial."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    d = tmp_path / "clip"; d.mkdir(); fd = tmp_path / "clip_flow"; fd.mkdir(); sd = tmp_path / "clip_seg"; sd.mkdir()
    H2, W2 = 2 * v.resy, 2 * v.resx
    names = ["%05d.png" % f for f in range(v.F)]
    imgs, masks, flows = [], [], []
    for f in range(v.F):
        im = rng.integers(0, 256, (H2, W2, 3), dtype=np.uint8); mk = rng.integers(0, 256, (H2, W2), dtype=np.uint8)
        Image.fromarray(im).save(str(d / names[f])); Image.fromarray(mk).save(str(sd / names[f]))
        imgs.append(im); masks.append(mk)
    fh, fw = v.resy + 8, v.resx + 16                                  # RAFT's padded resolution differs from the frames'
    for f in range(v.F - 1):
        f12 = (rng.standard_normal((fh, fw, 2)) * 1.5).astype(np.float32)
        f21 = (-f12 + rng.standard_normal((fh, fw, 2)) * 0.6).astype(np.float32)
        np.save(fd / ("%s_%s.npy" % (names[f], names[f + 1])), f12); np.save(fd / ("%s_%s.npy" % (names[f + 1], names[f])), f21)
        flows.append((f12, f21))
    return d, imgs, masks, flows


def _oracle_builder(v, imgs, masks, flows):
    """load_input_data (unwrap_utils.py:40-163) with its two OpenCV calls taken from oracle/cv_oracle.py (the restatement of
    cv2.resize / cv2.remap pinned by tests/test_cv_oracle.py)."""
    from oracle import cv_oracle as C
    F = v.F
    frames = np.zeros((v.resy, v.resx, 3, F), np.float32); mk = np.zeros((v.resy, v.resx, F), np.float32)
    fl = np.zeros((v.resy, v.resx, 2, F), np.float32); fr = np.zeros_like(fl)
    m = np.zeros((v.resy, v.resx, F), np.float32); mr = np.zeros_like(m)
    for i in range(F):
        frames[:, :, :, i] = C.cv_resize_linear(imgs[i].astype(np.float64) / 255.0, v.resx, v.resy)
        mk[:, :, i] = C.cv_resize_linear(masks[i].astype(np.float64) / 255.0, v.resx, v.resy)
    for i in range(F - 1):
        f12 = C.cv_resize_flow(flows[i][0], v.resy, v.resx); f21 = C.cv_resize_flow(flows[i][1], v.resy, v.resx)
        fl[:, :, :, i] = f12; fr[:, :, :, i + 1] = f21
        m[:, :, i] = C.cv_compute_consistency(f12, f21) < 1.0
        mr[:, :, i + 1] = C.cv_compute_consistency(f21, f12) < 1.0
    return frames, mk, fl, fr, m, mr


def test_host_input_builder_equals_opencv_restatement(tmp_path, small_seg_video):
    """The numpy loader of the product (stage1.load_input_data_single / load_mask_frames) is bit-identical to the loader
    assembled from the oracle's OpenCV restatement: float32 resize coefficients, 1/32-px remap positions."""
    import aiod_amd.stage1 as S
    v = small_seg_video
    d, imgs, masks, flows = _write_builder_case(tmp_path, v)
    hm, hf, hmr, hfr, hfl = S.load_input_data_single(v.resy, v.resx, 200, d, True, tmp_path, "clip")
    hmask = S.load_mask_frames(v.resy, v.resx, v.F, tmp_path, "clip")
    frames, mk, fl, fr, m, mr = _oracle_builder(v, imgs, masks, flows)
    assert np.array_equal(hf, frames) and np.array_equal(hmask, mk)
    assert np.array_equal(hfl[..., 0], fl) and np.array_equal(hfr[..., 0], fr)
    assert np.array_equal(hm[..., 0], m) and np.array_equal(hmr[..., 0], mr)
    assert 0.02 < m[:, :, :-1].mean() < 0.98                              # a non-trivial mask
    # the 1/32-px quantisation matters: an exact-fraction bilinear warp flips pixels of this mask
    f12, f21 = fl[:, :, :, 0], fr[:, :, :, 1]
    x = f12[:, :, 0] + np.arange(v.resx, dtype=np.float32); y = f12[:, :, 1] + np.arange(v.resy, dtype=np.float32)[:, None]
    x0 = np.floor(x).astype(int); y0 = np.floor(y).astype(int); fx = (x - x0)[..., None]; fy = (y - y0)[..., None]
    def tap(yy, xx):
        ok = (xx >= 0) & (xx < v.resx) & (yy >= 0) & (yy < v.resy)
        return np.where(ok[..., None], f21[np.clip(yy, 0, v.resy - 1), np.clip(xx, 0, v.resx - 1)], 0.0)
    wexact = tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) + tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy
    dd = f12 + wexact
    print("pixels whose consistency bit differs between exact-fraction and 1/32-px remap:", int(((np.sqrt((dd ** 2).sum(-1)) < 1.0) != (m[:, :, 0] > 0)).sum()))


@pytest.mark.gpu
def test_device_input_builder_equals_opencv_restatement(tmp_path, small_seg_video):
    """af_resize_bilinear / af_flow_consistency (k_resize_bilinear / k_flow_consistency, the device side of load_input_data,
    unwrap_utils.py:40-163) against the oracle's restatement of cv2.resize / cv2.remap: bit-identical tensors — frames,
    fractional fg masks, rescaled flows and both consistency masks (no pixel may flip)."""
    import torch
    import aiod_amd.stage1 as S
    from oracle import cv_oracle as C
    v = small_seg_video
    d, imgs, masks, flows = _write_builder_case(tmp_path, v)
    frames, mk, fl, fr, m, mr = _oracle_builder(v, imgs, masks, flows)
    dm, df, dmr, dfr, dfl, dmask = [t.cpu().numpy() for t in S.load_input_data_device(v.resy, v.resx, 200, d, True, tmp_path, "clip", with_masks=True)]
    assert np.array_equal(df, frames) and np.array_equal(dmask, mk)
    assert np.array_equal(dfl, fl) and np.array_equal(dfr, fr)
    assert np.array_equal(dm, m) and np.array_equal(dmr, mr)
    # the norm field itself
    f12 = torch.from_numpy(np.ascontiguousarray(fl[:, :, :, 0])).cuda(); f21 = torch.from_numpy(np.ascontiguousarray(fr[:, :, :, 1])).cuda()
    out = torch.empty(v.resy, v.resx, device="cuda")
    aiod = __import__("aiod_amd")
    aiod.atlasfit.flow_consistency_device(f12, f21, out, 1, 0, thresh=0.0)
    ref = C.cv_compute_consistency(fl[:, :, :, 0], fr[:, :, :, 1])
    assert np.abs(out.cpu().numpy() - ref).max() <= 1.2e-7 * np.abs(ref).max()      # (x ** .5 is powf in numpy, sqrtf on the device: <= 1 ulp)


@pytest.mark.gpu
@pytest.mark.parametrize("concurrent", [1, 2])
def test_multi_video_launcher_single_rank(tmp_path, small_video, monkeypatch, concurrent):
    """launch_videos.py with one rank: two clips through stage1.main on GPU 0, results tree per clip, one JSON summary —
    one after the other, and both at a time (two host threads, two handles and streams on the same GPU)."""
    import aiod_amd
    from aiod_amd import launch_videos as L
    (tmp_path / "data").mkdir()
    for name in ("clipA", "clipB"):
        _write_video_into(tmp_path / "data", small_video, name)
    cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
    cfg.update(samples_batch=256, iters_num=11, evaluate_every=10, pretrain_iter_number=1, stop_global_rigidity=5)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    out = L.run(["--vid_names", "clipA", "clipB", "--config", str(tmp_path / "cfg.json"), "--root", str(tmp_path / "data"), "--down", "1", "--seed", "3",
                 "--concurrent", str(concurrent)])
    assert out["videos"] == 2 and out["n_gpus"] == 1 and set(out["psnr"]) == {"clipA", "clipB"}
    assert abs(out["psnr"]["clipA"] - out["psnr"]["clipB"]) < 1e-9          # same clip, same seed -> same result, also when they share the GPU
    for name in ("clipA", "clipB"):
        assert len(list((tmp_path / "results" / name / "stage_1" / "output").glob("*.png"))) == small_video.F


def _write_video_into(root, v, name):
    from PIL import Image
    d = root / name; d.mkdir(); fd = root / (name + "_flow"); fd.mkdir()
    names = []
    for f in range(v.F):
        fn = "%05d.png" % f; names.append(fn)
        Image.fromarray(np.round(v.video_frames[:, :, :, f].numpy() * 255).astype(np.uint8)).save(str(d / fn))
    for f in range(v.F - 1):
        np.save(fd / ("%s_%s.npy" % (names[f], names[f + 1])), v.optical_flows[:, :, :, f, 0].numpy())
        np.save(fd / ("%s_%s.npy" % (names[f + 1], names[f])), v.optical_flows_reverse[:, :, :, f + 1, 0].numpy())
    return d


def test_pipeline_driver_builds_the_reference_command_sequence():
    """run_pipeline.py mirrors test.py (test.py:17-43): frame extraction, stage 1 (ours, --gpu forwarded), stage 2."""
    import argparse
    import aiod_amd  # noqa: F401
    from aiod_amd import run_pipeline as R
    o = argparse.Namespace(video_name="data/test/Winter_Scenes_in_Holland.mp4", video_frame_folder=None, fps=10, gpu=3, class_name=None)
    cmds = [c for _, c in R.build_commands(o)]
    assert cmds[0] == "./data/test/Winter_Scenes_in_Holland"
    assert cmds[1] == "ffmpeg -i data/test/Winter_Scenes_in_Holland.mp4 -vf fps=10 -start_number 0 ./data/test/Winter_Scenes_in_Holland/%05d.png"
    assert cmds[2].endswith("stage1.py --vid_name Winter_Scenes_in_Holland --gpu 3")
    assert cmds[3] == "python src/neural_filter_and_refinement.py --video_name Winter_Scenes_in_Holland --fps 10"
    o = argparse.Namespace(video_name=None, video_frame_folder="clips/abc", fps=12, gpu=0, class_name="person")
    cmds = [c for _, c in R.build_commands(o)]
    assert cmds[0] == "mv abc ./data/test/abc" and "stage1_seg.py --vid_name abc --class_name person --gpu 0" in cmds[1]


@pytest.mark.parametrize("script", ["all-in-one-deflicker_amd/stage1.py", "all-in-one-deflicker_amd/stage1_seg.py", "all-in-one-deflicker_amd/run_pipeline.py",
                                    "all-in-one-deflicker_amd/launch_videos.py", "bench.py"])
def test_cli_help_renders(script):
    """argparse formats every help string with %: a literal percent sign in one of them breaks `--help` for the whole CLI."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, script), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "usage" in r.stdout.lower(), r.stderr[-500:]
