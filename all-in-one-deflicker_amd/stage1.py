"""Drop-in stage-1 entry points: same flags, config keys, on-disk inputs and results tree as the reference's
`src/stage1_neural_atlas.py` (CLI :257-281, main :27-255) and, with `two_layer` (see stage1_seg.py), its fg/bg
twin `src/stage1_neural_atlas_seg.py` (CLI :333-369, main :25-331; extra input `<vid>_seg/*.png` read by
`load_input_data`, unwrap_utils.py:40-103; checkpoint with the four nets, evaluate.py:215-232), with the
optimisation loop running in libatlasfit.so.  Host-side pieces restated here (numpy / PIL only — cv2, imageio, skimage, tensorboard are not
required): the input builder `load_input_data_single` (src/models/stage_1/unwrap_utils.py:105-163 incl.
`resize_flow` :33-38 and `compute_consistency` :10-23), and the parts of `evaluate_model_single`
(src/models/stage_1/evaluate.py:605-793) that stage 2 and the metric consume: the checkpoint (:616-622), the
reconstructed frames `output/%05d.png` with the reference's truncating uint8 cast (:732-733) and the
`PSNR_<mean>` marker file (:740-743,781-783).  The debug mp4 / matplotlib panels / tensorboard images are not
produced (out of scope).

    python all-in-one-deflicker_amd/stage1.py --vid_name <name> [--config config_flow_100.json] [--root data/test/] [--down 4] [--gpu 0]
"""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------------------------
# input builder (unwrap_utils.py)
def _linear_coeffs(src, dst):
    """OpenCV resize.cpp, INTER_LINEAR: first tap and the two FLOAT coefficients per destination index:
    f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s; taps outside clamp with weight 1."""
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * (float(src) / float(dst)) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo, hi = s < 0, s >= src - 1
    s[lo] = 0; f[lo] = 0.0
    s[hi] = src - 1; f[hi] = 0.0
    return s, np.minimum(s + 1, src - 1), (np.float32(1.0) - f).astype(np.float32), f


def resize_bilinear(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) with the default INTER_LINEAR, in OpenCV's arithmetic: half-pixel centres, edge
    clamping, no anti-aliasing, float32 interpolation coefficients, horizontal pass then vertical pass in the image's
    own precision (double for the float64 frames / masks, float for the float32 flows).  img: (H, W[, C]) float array."""
    img = np.asarray(img)
    if img.dtype not in (np.float32, np.float64):
        img = img.astype(np.float64)
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img.copy()
    wt = img.dtype.type
    sx, sx1, a0, a1 = _linear_coeffs(w, new_w)
    sy, sy1, b0, b1 = _linear_coeffs(h, new_h)
    shp = (1, new_w) + (1,) * (img.ndim - 2)
    a0, a1 = a0.astype(wt).reshape(shp), a1.astype(wt).reshape(shp)
    top = img[sy][:, sx] * a0 + img[sy][:, sx1] * a1
    bot = img[sy1][:, sx] * a0 + img[sy1][:, sx1] * a1
    shp = (new_h, 1) + (1,) * (img.ndim - 2)
    return (top * b0.astype(wt).reshape(shp) + bot * b1.astype(wt).reshape(shp)).astype(img.dtype)


def resize_flow(flow, newh, neww):
    """unwrap_utils.py:33-38 (u is scaled by newh/oldh and v by neww/oldw, as the reference does)."""
    oldh, oldw = flow.shape[0:2]
    flow = resize_bilinear(flow.astype(np.float32), neww, newh)
    flow[:, :, 0] *= np.float32(newh / oldh)
    flow[:, :, 1] *= np.float32(neww / oldw)
    return flow


def _remap_bilinear_zero(img, mapx, mapy):
    """cv2.remap(img, map, None, INTER_LINEAR), constant-0 border (unwrap_utils.py:22), in OpenCV's arithmetic: the
    float map is converted to fixed point with INTER_BITS = 5 (cvRound(x * 32), half to even) — positions are quantised
    to 1/32 px — and the four taps are blended with the table weights in float32, left to right."""
    h, w = img.shape[:2]
    qx = np.rint(mapx.astype(np.float32) * np.float32(32)).astype(np.int64)
    qy = np.rint(mapy.astype(np.float32) * np.float32(32)).astype(np.int64)
    x0, y0 = qx >> 5, qy >> 5
    fx = ((qx & 31).astype(np.float32) / np.float32(32))[..., None]
    fy = ((qy & 31).astype(np.float32) / np.float32(32))[..., None]
    one = np.float32(1.0)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[..., None], v, 0.0).astype(np.float32)

    return ((tap(y0, x0) * ((one - fy) * (one - fx)) + tap(y0, x0 + 1) * ((one - fy) * fx))
            + tap(y0 + 1, x0) * (fy * (one - fx))) + tap(y0 + 1, x0 + 1) * (fy * fx)


def compute_consistency(flow12, flow21):
    """unwrap_utils.py:10-23."""
    h, w = flow12.shape[:2]
    mapx = flow12[:, :, 0] + np.arange(w, dtype=np.float32)
    mapy = flow12[:, :, 1] + np.arange(h, dtype=np.float32)[:, None]
    diff = flow12 + _remap_bilinear_zero(flow21, mapx, mapy)
    return (diff[:, :, 0] ** 2 + diff[:, :, 1] ** 2) ** 0.5


def load_mask_frames(resy, resx, number_of_frames, vid_root, vid_name):
    """The mask part of load_input_data (unwrap_utils.py:43,60,66-70): `<vid>_seg/*.{jpg,png}` / 255, resized with
    cv2.resize(mask, (resx, resy), cv2.INTER_NEAREST) — the flag lands in the `dst` slot, so the resize is the
    default BILINEAR one and the masks are fractional."""
    from PIL import Image
    seg_dir = Path(vid_root) / f"{vid_name}_seg"
    files = sorted(list(seg_dir.glob("*.jpg")) + list(seg_dir.glob("*.png")))
    if len(files) < number_of_frames:
        raise FileNotFoundError("%d mask frames under %s, need %d: run the reference's src/preprocess_mask_*.py first "
                                "(external segmentation models, out of scope of this library)" % (len(files), seg_dir, number_of_frames))
    out = np.zeros((resy, resx, number_of_frames), np.float32)
    for i in range(number_of_frames):
        m = np.array(Image.open(str(files[i]))).astype(np.float64) / 255.0
        if m.ndim == 3:
            m = m[:, :, 0]
        out[:, :, i] = resize_bilinear(m, resx, resy).astype(np.float32)
    return out


def load_input_data_single(resy, resx, maximum_number_of_frames, data_folder, filter_optical_flow, vid_root, vid_name):
    """unwrap_utils.py:105-163.  Returns numpy fp32 arrays in the reference's layouts:
    video_frames (resy,resx,3,F), optical_flows / _reverse (resy,resx,2,F,1), masks (resy,resx,F,1)."""
    from PIL import Image
    data_folder, vid_root = Path(data_folder), Path(vid_root)
    out_flow_dir = vid_root / f"{vid_name}_flow"
    input_files = sorted(list(data_folder.glob("*.jpg")) + list(data_folder.glob("*.png")))
    if not input_files:
        raise FileNotFoundError("no *.jpg / *.png frames under %s" % data_folder)
    F = int(np.minimum(maximum_number_of_frames, len(input_files)))
    video_frames = np.zeros((resy, resx, 3, F), np.float32)
    optical_flows = np.zeros((resy, resx, 2, F, 1), np.float32)
    optical_flows_mask = np.zeros((resy, resx, F, 1), np.float32)
    optical_flows_reverse = np.zeros((resy, resx, 2, F, 1), np.float32)
    optical_flows_reverse_mask = np.zeros((resy, resx, F, 1), np.float32)
    for i in range(F):
        im = np.array(Image.open(str(input_files[i]))).astype(np.float64) / 255.0
        if im.ndim == 2:
            im = np.tile(im[:, :, None], [1, 1, 3])
        video_frames[:, :, :, i] = resize_bilinear(im[:, :, :3], resx, resy).astype(np.float32)
    for i in range(F - 1):
        fn1, fn2 = input_files[i].name, input_files[i + 1].name
        f12p, f21p = out_flow_dir / f"{fn1}_{fn2}.npy", out_flow_dir / f"{fn2}_{fn1}.npy"
        if not f12p.exists() or not f21p.exists():
            raise FileNotFoundError("optical flow %s missing: run the reference's src/preprocess_optical_flow.py (RAFT on "
                                    "PyTorch-ROCm, out of scope of this library) first" % f12p)
        flow12, flow21 = np.load(f12p).astype(np.float32), np.load(f21p).astype(np.float32)
        if flow12.shape[0] != resy or flow12.shape[1] != resx:
            flow12 = resize_flow(flow12, newh=resy, neww=resx)
            flow21 = resize_flow(flow21, newh=resy, neww=resx)
        optical_flows[:, :, :, i, 0] = flow12
        optical_flows_reverse[:, :, :, i + 1, 0] = flow21
        if filter_optical_flow:
            optical_flows_mask[:, :, i, 0] = compute_consistency(flow12, flow21) < 1.0
            optical_flows_reverse_mask[:, :, i + 1, 0] = compute_consistency(flow21, flow12) < 1.0
        else:
            optical_flows_mask[:, :, i, 0] = 1.0
            optical_flows_reverse_mask[:, :, i + 1, 0] = 1.0
    return optical_flows_mask, video_frames, optical_flows_reverse_mask, optical_flows_reverse, optical_flows


def _prefetch(fn, items, workers=8, depth=16):
    """fn(item) for every item, in order, computed by a thread pool at most `depth` items ahead of the consumer: PIL's decoders,
    zlib and np.load release the GIL, so file decoding runs on the host's cores while the caller feeds the GPU (a 200-frame 1080p
    clip would not fit in memory decoded all at once, hence the bound)."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        q, it = deque(), iter(items)
        for x in it:
            q.append(ex.submit(fn, x))
            if len(q) >= depth:
                break
        while q:
            r = q.popleft().result()
            nxt = next(it, None)
            if nxt is not None:
                q.append(ex.submit(fn, nxt))
            yield r


def count_input_frames(maximum_number_of_frames, data_folder):
    """F as load_input_data_single computes it (unwrap_utils.py:110,112), without decoding anything."""
    data_folder = Path(data_folder)
    n = len(list(data_folder.glob("*.jpg")) + list(data_folder.glob("*.png")))
    if not n:
        raise FileNotFoundError("no *.jpg / *.png frames under %s" % data_folder)
    return int(np.minimum(maximum_number_of_frames, n))


def load_input_data_device(resy, resx, maximum_number_of_frames, data_folder, filter_optical_flow, vid_root, vid_name,
                           with_masks=False, device=0):
    """load_input_data_single / load_input_data (unwrap_utils.py:40-163) with the per-pixel work on the GPU: files
    are decoded on the host (PIL / np.load) and uploaded as they are; the bilinear resizes, the flow rescaling and the
    forward/backward consistency masks run in libatlasfit.so (af_resize_bilinear, af_flow_consistency) and write
    straight into the reference-layout tensors in HBM, which af_upload_video then packs without a host round trip.
    Same arithmetic as the host functions above (fp64 interpolation, fp32 consistency); ~50 s of numpy work for an
    80-frame 4x-downsampled clip become file decoding only.  Returns torch CUDA tensors:
    (optical_flows_mask, video_frames, optical_flows_reverse_mask, optical_flows_reverse, optical_flows[, mask_frames])."""
    import torch
    from PIL import Image
    from .atlasfit import flow_consistency_device, resize_bilinear_device
    dev = torch.device("cuda", device)
    data_folder, vid_root = Path(data_folder), Path(vid_root)
    out_flow_dir = vid_root / f"{vid_name}_flow"
    input_files = sorted(list(data_folder.glob("*.jpg")) + list(data_folder.glob("*.png")))
    if not input_files:
        raise FileNotFoundError("no *.jpg / *.png frames under %s" % data_folder)
    F = int(np.minimum(maximum_number_of_frames, len(input_files)))
    video_frames = torch.zeros((resy, resx, 3, F), device=dev)
    optical_flows = torch.zeros((resy, resx, 2, F), device=dev)
    optical_flows_reverse = torch.zeros((resy, resx, 2, F), device=dev)
    optical_flows_mask = torch.zeros((resy, resx, F), device=dev)
    optical_flows_reverse_mask = torch.zeros((resy, resx, F), device=dev)
    mask_frames = torch.zeros((resy, resx, F), device=dev) if with_masks else None
    mask_files = []
    if with_masks:
        seg_dir = vid_root / f"{vid_name}_seg"
        mask_files = sorted(list(seg_dir.glob("*.jpg")) + list(seg_dir.glob("*.png")))
        if len(mask_files) < F:
            raise FileNotFoundError("%d mask frames under %s, need %d" % (len(mask_files), seg_dir, F))

    def u8(path, channels):
        im = np.array(Image.open(str(path)))
        if im.dtype != np.uint8:
            raise ValueError("%s: only 8-bit images are handled on the device path" % path)
        if channels == 3:
            im = np.tile(im[:, :, None], [1, 1, 3]) if im.ndim == 2 else im[:, :, :3]
        else:
            im = im[:, :, None] if im.ndim == 2 else im[:, :, :1]
        return np.ascontiguousarray(im)

    # decode on a thread pool (round 5: 160 serial PIL / np.load calls were ~3 s of the CLI's wall clock), feed the GPU in file order
    def dec_frame(i):
        return u8(input_files[i], 3), (u8(mask_files[i], 1) if with_masks else None)

    for i, (im, mk) in enumerate(_prefetch(dec_frame, range(F))):
        resize_bilinear_device(torch.from_numpy(im).to(dev), video_frames, resy, resx, 3 * F, F, i, device=device)
        if with_masks:
            resize_bilinear_device(torch.from_numpy(mk).to(dev), mask_frames, resy, resx, F, 0, i, device=device)

    def flow(arr):
        f = torch.from_numpy(arr).to(dev)
        if f.shape[0] != resy or f.shape[1] != resx:
            oldh, oldw = f.shape[0], f.shape[1]
            r = torch.empty((resy, resx, 2), device=dev)
            resize_bilinear_device(f, r, resy, resx, 2, 1, 0, scale=(resy / oldh, resx / oldw), device=device)   # resize_flow, :33-38
            f = r
        return f

    def dec_flows(i):
        fn1, fn2 = input_files[i].name, input_files[i + 1].name
        f12p, f21p = out_flow_dir / f"{fn1}_{fn2}.npy", out_flow_dir / f"{fn2}_{fn1}.npy"
        if not f12p.exists() or not f21p.exists():
            raise FileNotFoundError("optical flow %s missing: run the reference's src/preprocess_optical_flow.py first" % f12p)
        return tuple(np.ascontiguousarray(np.load(q).astype(np.float32)) for q in (f12p, f21p))

    for i, (a12, a21) in enumerate(_prefetch(dec_flows, range(F - 1), depth=8)):
        f12, f21 = flow(a12), flow(a21)
        optical_flows[:, :, :, i] = f12
        optical_flows_reverse[:, :, :, i + 1] = f21
        if filter_optical_flow:
            flow_consistency_device(f12, f21, optical_flows_mask, F, i, 1.0, device=device)
            flow_consistency_device(f21, f12, optical_flows_reverse_mask, F, i + 1, 1.0, device=device)
        else:
            optical_flows_mask[:, :, i] = 1.0
            optical_flows_reverse_mask[:, :, i + 1] = 1.0
    out = (optical_flows_mask, video_frames, optical_flows_reverse_mask, optical_flows_reverse, optical_flows)
    return out + (mask_frames,) if with_masks else out


# ---------------------------------------------------------------------------------------------
# checkpoint format (evaluate.py:616-622 / :215-232; resume stage1_neural_atlas.py:141-146 / _seg.py:180-187)
def _ckpt_layout(two_layer):
    """(checkpoint key, net) in the optimizer's param-group order (stage1_neural_atlas.py:132-134:
    mapping, atlas; stage1_neural_atlas_seg.py:165-169: mapping1, mapping2, alpha, atlas)."""
    from .atlasfit import NET_ALPHA, NET_ATLAS, NET_MAPPING1, NET_MAPPING2
    if two_layer:
        return [("model_F_mapping1_state_dict", NET_MAPPING1), ("model_F_mapping2_state_dict", NET_MAPPING2),
                ("model_F_alpha_state_dict", NET_ALPHA), ("F_atlas_state_dict", NET_ATLAS)]
    return [("model_F_mapping1_state_dict", NET_MAPPING1), ("F_atlas_state_dict", NET_ATLAS)]


def _torch_module(net, sd):
    import torch
    from .atlasfit import imlp_shapes
    m = torch.nn.Module()
    m.hidden = torch.nn.ModuleList([torch.nn.Linear(int(np.asarray(sd["hidden.%d.weight" % i]).shape[1]), int(np.asarray(sd["hidden.%d.weight" % i]).shape[0]))
                                    for i in range(len(sd) // 2)])
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    return m


def save_checkpoint(af, path, iteration):
    """torch.save of the reference's dict: the nets' state dicts, `iteration`, and `optimizer_all_state_dict`
    (a genuine torch.optim.Adam state dict in the reference's param-group order)."""
    import torch
    layout = _ckpt_layout(af.two_layer)
    mods = [_torch_module(net, af.state_dict(net)) for _, net in layout]
    opt = torch.optim.Adam([{"params": list(m.parameters())} for m in mods], lr=float(af.cfg.lr))
    for (_, net), mod in zip(layout, mods):
        m, v, step = af.adam_state(net)
        off = 0
        for p in mod.parameters():
            n = p.numel()
            opt.state[p] = {"step": torch.tensor(float(step)), "exp_avg": torch.from_numpy(m[off:off + n].reshape(p.shape).copy()),
                            "exp_avg_sq": torch.from_numpy(v[off:off + n].reshape(p.shape).copy())}
            off += n
    ck = {key: mod.state_dict() for (key, _), mod in zip(layout, mods)}
    ck["iteration"] = iteration
    ck["optimizer_all_state_dict"] = opt.state_dict()
    torch.save(ck, str(path))


def load_checkpoint(af, path):
    import torch
    from .atlasfit import imlp_shapes
    ck = torch.load(str(path), map_location="cpu", weights_only=False)
    layout = _ckpt_layout(af.two_layer)
    for key, net in layout:
        af.load_state_dict(net, ck[key])
    st = ck["optimizer_all_state_dict"]["state"]
    idx, step = 0, 0
    for _, net in layout:
        ms, vs = [], []
        for (o, k) in imlp_shapes(net, af.cfg):
            for _ in range(2):               # weight, bias
                s = st[idx]; idx += 1
                ms.append(s["exp_avg"].reshape(-1).numpy()); vs.append(s["exp_avg_sq"].reshape(-1).numpy()); step = int(s["step"])
        af.set_adam_state(net, np.concatenate(ms), np.concatenate(vs), step)
    return int(ck["iteration"])


# ---------------------------------------------------------------------------------------------
def evaluate_model_single(af, video_frames, results_folder, iteration, save_checkpoint_file=True):
    """The stage-2 hand-off + metric of evaluate.py:605-793: checkpoint, output/%05d.png, <iter>/PSNR_<mean>."""
    from PIL import Image
    results_folder = Path(results_folder)
    eval_dir = results_folder / ("%06d" % iteration)
    (results_folder / "output").mkdir(parents=True, exist_ok=True)
    eval_dir.mkdir(parents=True, exist_ok=True)
    if save_checkpoint_file:
        save_checkpoint(af, results_folder / "checkpoint", iteration)
        if af.two_layer:                                   # evaluate.py:224-232 keeps a second copy per evaluation
            save_checkpoint(af, eval_dir / "checkpoint", iteration)
    from concurrent.futures import ThreadPoolExecutor
    F = video_frames.shape[3]
    psnrs = np.zeros(F)

    def write(f, rec):      # the reference's truncating uint8 cast (evaluate.py:732-733); zlib releases the GIL, the encodes run beside the renders
        Image.fromarray((rec.astype(np.float64) * 255).astype(np.uint8)).save(str(results_folder / "output" / ("%05d.png" % f)))

    with ThreadPoolExecutor(max_workers=8) as ex:
        jobs = []
        for f in range(F):
            rec, sse = af.render_frame(f)
            jobs.append(ex.submit(write, f, rec))
            psnrs[f] = 10.0 * np.log10(1.0 / (sse / rec.size))
            if len(jobs) > 16:          # renders outrun the PNG encodes: at most 16 frames (25 MB each at 1080p) wait in the queue
                jobs.pop(0).result()
        for j in jobs:
            j.result()
    print(psnrs.mean())
    open(eval_dir / ("PSNR_%f" % psnrs.mean()), "a").close()
    return float(psnrs.mean())


def main(config, args, two_layer=False):
    """stage1_neural_atlas.py:27-255 (two_layer: stage1_neural_atlas_seg.py:25-331) with the loop in libatlasfit.so."""
    from PIL import Image
    from . import atlasfit as A
    import glob
    frames_list = sorted(glob.glob(os.path.join(args.vid_path, "*g")))
    if not frames_list:
        raise FileNotFoundError("no frames under %s" % args.vid_path)
    w, h = Image.open(frames_list[0]).size
    resx, resy = w, h
    if args.down is not None:
        resx, resy = int(resx / args.down), int(resy / args.down)
    iters_num = config["iters_num"]
    evaluate_every = int(config["evaluate_every"])
    data_folder = Path(args.vid_path)
    vid_name, vid_root = data_folder.name, data_folder.parent
    results_folder = Path("./results/%s/stage_1" % vid_name)
    results_folder.mkdir(parents=True, exist_ok=True)
    with open(results_folder / "config.json", "w") as f:
        json.dump(config, f, indent=4)
    import math
    import threading
    import time
    import torch
    t_wall = [("start", time.perf_counter())]
    mark = lambda name: t_wall.append((name, time.perf_counter()))
    dev_ord = getattr(args, "device_ordinal", 0)
    F = count_input_frames(config["maximum_number_of_frames"], data_folder)
    af = A.AtlasFit(A.default_config(resx, resy, F, config, two_layer=two_layer), device=dev_ord)
    af.range_fallback = True      # a weight beyond the fp16 images' range (AF_ERANGE) must not stop a run: go on from the same state on the bf16x6 chains
    # the arithmetic of this run next to its configuration (include/atlasfit.h af_set_mlp_mode / af_set_dw_mode; AF_EXPERIMENT overrides named)
    with open(results_folder / "config.json", "w") as f:
        json.dump(dict(config, atlasfit_arithmetic=af.arithmetic), f, indent=4)
    if af.arithmetic["overrides"]:
        print("arithmetic overrides in force:", af.arithmetic)
    arithmetic_at_start = dict(af.arithmetic)
    # Random draws: the reference uses torch's process-global RNG.  With --seed (an extension) every draw of this call comes
    # from its own torch.Generator, so concurrent videos in one process (launch_videos.py --concurrent) stay reproducible
    # and independent; without it the global RNG is used like the reference does.
    seed = getattr(args, "seed", None)
    gen = torch.Generator().manual_seed(int(seed)) if seed is not None else None
    draw = lambda: int(torch.randint(2 ** 31, (1,), generator=gen))
    start_iteration = 0
    pre = None
    if not config["load_checkpoint"]:
        # nn.Linear default init in the reference's construction order (:112-128; seg :127-161 mapping1, mapping2, atlas, alpha)
        order = (A.NET_MAPPING1, A.NET_MAPPING2, A.NET_ATLAS, A.NET_ALPHA) if two_layer else (A.NET_MAPPING1, A.NET_ATLAS)
        for net in order:
            sd = {}
            for i, (o, k) in enumerate(A.imlp_shapes(net, af.cfg)):
                w, b = torch.empty(o, k), torch.empty(o)
                torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=gen)          # nn.Linear.reset_parameters
                bound = 1 / math.sqrt(k)
                torch.nn.init.uniform_(b, -bound, bound, generator=gen)
                sd["hidden.%d.weight" % i] = w; sd["hidden.%d.bias" % i] = b
            af.load_state_dict(net, sd)
        # pre_train_mapping reads nothing of the video (unwrap_utils.py:176-198: random pixel coordinates of frame f against uv = 0.8 xy;
        # asserted by the fixture generators under tests/), so it starts NOW, on the handle's stream from its own thread (ctypes drops the GIL), while this
        # thread decodes, resizes and uploads the clip: 1.7 s (single) / 3.3 s (fg/bg) of the schedule leave the critical path (round 5).
        # The draws keep the reference's order: init, pre-train seed(s), sampler seed.
        jobs = []
        if config["pretrain_mapping1"]:
            print("pre-training")
            jobs.append((draw(), A.NET_MAPPING1))
        if two_layer and config["pretrain_mapping2"]:
            jobs.append((draw(), A.NET_MAPPING2))
        err = []

        def pretrain():
            try:
                for sd_, net in jobs:
                    af.pre_train_mapping(config["pretrain_iter_number"], seed=sd_, net=net)
            except BaseException as e:      # surfaces in the main thread at join
                err.append(e)
        pre = threading.Thread(target=pretrain, name="af-pretrain")
        pre.start()
    mark("handle + init")
    try:
        if getattr(args, "host_loader", False):      # the numpy restatement of the reference loader (slow; kept as the cross-check)
            flows_mask, video_frames, flows_rev_mask, flows_rev, flows = load_input_data_single(
                resy, resx, config["maximum_number_of_frames"], data_folder, True, vid_root, vid_name)
            mask_frames = load_mask_frames(resy, resx, video_frames.shape[3], vid_root, vid_name) if two_layer else None
        else:
            t = load_input_data_device(resy, resx, config["maximum_number_of_frames"], data_folder, True, vid_root, vid_name,
                                       with_masks=two_layer, device=dev_ord)
            flows_mask, video_frames, flows_rev_mask, flows_rev, flows = t[:5]
            mask_frames = t[5] if two_layer else None
        if video_frames.shape[3] != F:      # the handle (and the pre-train running on it) was sized from the directory listing
            raise RuntimeError("the loader produced %d frames, the handle was created for %d (files changed under %s?)" % (video_frames.shape[3], F, data_folder))
    except BaseException:       # a missing flow / mask file: the pre-train thread still owns the handle — let it finish before the handle goes away
        if pre is not None:
            pre.join()
        af.close()
        raise
    mark("input builder")
    if pre is not None:
        pre.join()
        if err:
            af.close()
            raise err[0]
    mark("pre-train (overlapped) done")
    af.upload_video(video_frames, flows, flows_rev, flows_mask, flows_rev_mask, mask_frames)
    mark("table packed")
    if config["load_checkpoint"]:
        start_iteration = load_checkpoint(af, config["checkpoint_path"])
    sampler_seed = draw()
    i = start_iteration
    last_psnr = None
    while i < iters_num:
        # run up to (and including) the next evaluation iteration in one call; evaluate when i % evaluate_every == 0 and i > start
        nxt = ((i // evaluate_every) + 1) * evaluate_every
        stop = min(iters_num - 1, nxt)
        af.train_steps(i, stop - i + 1, None, seed=sampler_seed, return_losses=False)
        i = stop + 1
        if stop % evaluate_every == 0 and stop > start_iteration:
            last_psnr = evaluate_model_single(af, video_frames, results_folder, stop)
    if af.arithmetic["mlp_mode"] != arithmetic_at_start["mlp_mode"]:      # the range fallback switched the chains' arithmetic on the way: the record says so
        with open(results_folder / "config.json", "w") as f:
            json.dump(dict(config, atlasfit_arithmetic=af.arithmetic), f, indent=4)
    af.close()
    mark("loop + evaluation")
    if os.environ.get("AF_CLI_TIMING"):      # wall clock per stage of this process, for tools/cli_end_to_end.py
        print("AF_CLI_TIMING " + json.dumps({n: round(t1 - t0, 3) for (_, t0), (n, t1) in zip(t_wall, t_wall[1:])}), file=sys.stderr)
    return last_psnr


def _run_reference_preprocessors(args, two_layer):
    """The reference's stage-1 scripts first shell out to its RAFT flow precompute (stage1_neural_atlas.py:276-278)
    and, for the fg/bg path, to a mask preprocessor (stage1_neural_atlas_seg.py:353-366).  Those stay the reference's own
    PyTorch-ROCm code (out of scope here): when this CLI runs inside a checkout of the reference (the scripts exist
    under ./src) it issues the same commands; elsewhere the inputs must already be on disk."""
    import subprocess
    if getattr(args, "skip_preprocess", False):
        return
    cmds = []
    if os.path.exists("src/preprocess_optical_flow.py"):
        cmds.append("python src/preprocess_optical_flow.py --vid-path %s --gpu %s " % (args.vid_path, args.gpu))
    if two_layer:
        if args.class_name == "portrait" and os.path.exists("src/preprocess_mask_portrait.py"):
            cmds.append("python src/preprocess_mask_portrait.py --vid-path %s --gpu %s " % (args.vid_path, args.gpu))
        elif args.class_name != "portrait" and os.path.exists("src/preprocess_mask_rcnn.py"):
            cmds.append("python src/preprocess_mask_rcnn.py --vid-path %s --class_name %s --gpu %s " % (args.vid_path, args.class_name, args.gpu))
    for cmd in cmds:
        print(cmd)
        subprocess.call(cmd, shell=True)


def _cli(argv=None, two_layer=False):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="config_flow_100.json")
    parser.add_argument("--vid_name", type=str, default="Around_the_world_in_1896_001")
    parser.add_argument("--root", type=str, default="data/test/")
    parser.add_argument("--down", type=int, default=1 if two_layer else 4)      # stage1_neural_atlas_seg.py:339 vs stage1_neural_atlas.py:262
    parser.add_argument("--gpu", type=int, default=0)
    if two_layer:
        parser.add_argument("--class_name", type=str, default="portrait", help="(reference flag; the mask preprocessors are external)")
    parser.add_argument("--seed", type=int, default=None, help="(extension) seed torch's RNG for reproducible runs")
    parser.add_argument("--skip_preprocess", action="store_true", help="(extension) do not call the reference's flow / mask preprocessors even if ./src has them")
    parser.add_argument("--host_loader", action="store_true", help="(extension) build the input tensors with the numpy loader instead of the device one")
    args = parser.parse_args(argv)
    # reference :267-268 sets CUDA_VISIBLE_DEVICES.  On ROCm HIP_VISIBLE_DEVICES takes precedence: when the scheduler / user
    # already restricted the visible GPUs, --gpu indexes into that list; both variables end up naming the one chosen device
    # (also for the preprocessor subprocesses).
    preset = [p for p in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if p.strip() != ""]
    if preset:
        if args.gpu >= len(preset):
            raise SystemExit("--gpu %d but HIP_VISIBLE_DEVICES=%s lists only %d device(s)" % (args.gpu, os.environ["HIP_VISIBLE_DEVICES"], len(preset)))
        chosen = preset[args.gpu].strip()
    else:
        chosen = "%d" % args.gpu
    os.environ["CUDA_VISIBLE_DEVICES"] = chosen
    os.environ["HIP_VISIBLE_DEVICES"] = chosen
    args.device_ordinal = 0
    args.vid_path = os.path.join(args.root, args.vid_name)
    _run_reference_preprocessors(args, two_layer)
    cfg_path = args.config if os.path.exists(args.config) else os.path.join("src/config", args.config)
    if os.path.exists(cfg_path):
        with open(cfg_path) as f:
            config = json.load(f)
    else:
        from .atlasfit import REFERENCE_CONFIG
        print("config %s not found: using the shipped hyper-parameters" % cfg_path)
        config = dict(REFERENCE_CONFIG)
    return main(config, args, two_layer)


if __name__ == "__main__":
    if __package__ in (None, ""):
        sys.path.insert(0, os.path.dirname(_HERE))
        import aiod_amd  # noqa: F401
        from aiod_amd import stage1 as _s
        _s._cli()
        sys.exit(0)
    _cli()
