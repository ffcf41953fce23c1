"""CPU ORACLE (test infrastructure, not product code) for the stage-1 neural-atlas hot path.

A plain PyTorch-CPU restatement of the reference algorithm, written from the reference's behaviour and
citing the file:line each function follows (paths relative to the reference repository root).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product path (`all-in-one-deflicker_amd/`) never does and has no CPU fallback.

Pinning: the reference ships no tests / golden vectors for this path.  The restatement
is pinned against the reference's OWN modules (IMLP, loss_utils, get_tuples, pre_train_mapping) imported
from /root/reference by `oracle/make_golden.py`, which also writes the fixtures under `tests/golden/`;
`tests/test_oracle.py` re-checks the restatement against those fixtures on every run.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# networks — src/models/stage_1/implicit_neural_networks.py:9-80
def positional_encoding(x, b):
    """implicit_neural_networks.py:9-13: feature index = k*(2*in) + [sin x0..x_{in-1}, cos x0..x_{in-1}]."""
    proj = x[:, :, None] * b[None, None, :]                      # (n, in, K)
    feats = torch.cat((torch.sin(proj), torch.cos(proj)), dim=1)  # (n, 2*in, K)
    return feats.transpose(2, 1).contiguous().view(x.shape[0], -1)


class OracleIMLP(nn.Module):
    """Coordinate MLP with the reference's layer sizes, parameter names (`hidden.{i}.weight/bias`) and
    construction order, so `torch.manual_seed(s)` yields the reference's initial weights
    (implicit_neural_networks.py:16-60)."""

    def __init__(self, input_dim, output_dim, hidden_dim=256, use_positional=True, positional_dim=10,
                 skip_layers=(4, 6), num_layers=8):
        super().__init__()
        self.use_positional = use_positional
        self.skip_layers = list(skip_layers)
        if use_positional:
            enc = 2 * input_dim * positional_dim
            self.b = torch.tensor([(2 ** j) * np.pi for j in range(positional_dim)])   # fp32(2^j * pi), :34
        else:
            enc = input_dim
        self.hidden = nn.ModuleList()
        for i in range(num_layers):
            fan_in = enc if i == 0 else (hidden_dim + enc if i in self.skip_layers else hidden_dim)
            self.hidden.append(nn.Linear(fan_in, output_dim if i == num_layers - 1 else hidden_dim))

    def forward(self, x):
        """implicit_neural_networks.py:62-80: ReLU before the (detached) skip concat, tanh on the output."""
        if self.use_positional:
            x = positional_encoding(x, self.b)
        skip_in = x.detach().clone()
        for i, layer in enumerate(self.hidden):
            if i > 0:
                x = F.relu(x)
            if i in self.skip_layers:
                x = torch.cat((x, skip_in), 1)
            x = layer(x)
        return torch.tanh(x)


def build_single_atlas_models(config, seed=None):
    """stage1_neural_atlas.py:112-128 (mapping first, then atlas: the RNG consumption order)."""
    if seed is not None:
        torch.manual_seed(seed)
    mapping = OracleIMLP(3, 2, config["number_of_channels_mapping1"], config["use_positional_encoding_mapping1"],
                         config["number_of_positional_encoding_mapping1"], [], config["number_of_layers_mapping1"])
    atlas = OracleIMLP(2, 3, config["number_of_channels_atlas"], True, config["positional_encoding_num_atlas"],
                       [4, 7], config["number_of_layers_atlas"])
    return mapping, atlas


# ------------------------------------------------------------------------------------------------
# sampling table — src/models/stage_1/unwrap_utils.py:166-173
def get_tuples(number_of_frames, resy, resx):
    """Column k = f*resy*resx + y*resx + x holds (x, y, f); the reference's mask `frames > -1` is all-true."""
    k = torch.arange(number_of_frames * resy * resx, dtype=torch.int64)
    p2 = resy * resx
    return torch.stack((k % resx, (k // resx) % resy, k // p2))


# ------------------------------------------------------------------------------------------------
# losses — src/models/stage_1/loss_utils.py
def gradient_loss_single(dx, dy, jif, mapping, atlas, rgb_out, resx, nframes):
    """loss_utils.py:134-170.  NB: normalises by resx/2 (the caller passes resx, stage1_neural_atlas.py:186-188)."""
    t = jif[2] / (nframes / 2.0) - 1
    xp1 = torch.cat(((jif[0] + 1) / (resx / 2) - 1, jif[1] / (resx / 2) - 1, t), dim=1)
    yp1 = torch.cat((jif[0] / (resx / 2) - 1, (jif[1] + 1) / (resx / 2) - 1, t), dim=1)
    dx_gt = dx[jif[1], jif[0], :, jif[2]].squeeze(1)
    dy_gt = dy[jif[1], jif[0], :, jif[2]].squeeze(1)
    uv_y = mapping(yp1)            # the reference evaluates (x, y+1) first (:154-155)
    uv_x = mapping(xp1)
    rgb_y = (atlas(uv_y * 0.5 + 0.5) + 1.0) * 0.5
    rgb_x = (atlas(uv_x * 0.5 + 0.5) + 1.0) * 0.5
    ddx = rgb_x - rgb_out
    ddy = rgb_y - rgb_out
    return torch.mean((dx_gt - ddx).norm(dim=1) ** 2 + (dy_gt - ddy).norm(dim=1) ** 2)


def rigidity_loss(jif, d, larger_dim, nframes, mapping, uv, uv_mapping_scale, return_all=False):
    """loss_utils.py:227-278: backward finite differences d pixels apart, ||J^T J||_F + ||(J^T J + eps)^-1||_F."""
    ys = torch.cat((jif[1] - d, jif[1])) / (larger_dim / 2) - 1
    xs = torch.cat((jif[0], jif[0] - d)) / (larger_dim / 2) - 1
    ts = torch.cat((jif[2], jif[2])) / (nframes / 2.0) - 1
    uv_p = mapping(torch.cat((xs, ys, ts), dim=1))
    u_p = uv_p[:, 0].view(2, -1)
    v_p = uv_p[:, 1].view(2, -1)
    du = uv[:, 0].unsqueeze(0) - u_p          # [0]: wrt y, [1]: wrt x
    dv = uv[:, 1].unsqueeze(0) - v_p
    du_dx = du[1] * larger_dim / 2
    du_dy = du[0] * larger_dim / 2
    dv_dy = dv[0] * larger_dim / 2
    dv_dx = dv[1] * larger_dim / 2
    J = torch.stack((torch.stack((du_dx, du_dy), dim=1), torch.stack((dv_dx, dv_dy), dim=1)), dim=1)
    J = J / uv_mapping_scale
    J = J / d
    G = torch.matmul(J.transpose(1, 2), J)
    a = G[:, 0, 0] + 0.001
    b = G[:, 0, 1]
    c = G[:, 1, 0]
    dd = G[:, 1, 1] + 0.001
    inv = torch.stack((torch.stack((dd, -b), dim=1), torch.stack((-c, a), dim=1)), dim=1)
    inv = inv / (a * dd - b * c)[:, None, None]
    per = (G ** 2).sum(1).sum(1).sqrt() + (inv ** 2).sum(1).sum(1).sqrt()
    return per if return_all else per.mean()


def flow_matches(jif, mask, flows, larger_dim, nframes, forward, uv):
    """loss_utils.py:326-356: rows with a non-zero consistency mask (ascending), advected by the flow."""
    sel = torch.where(mask[jif[1].squeeze(), jif[0].squeeze(), jif[2].squeeze(), :])
    step = 2 ** sel[1]
    rows = sel[0]
    j = jif[:, rows, 0]
    fl = flows[j[1], j[0], :, j[2], sel[1]]
    tgt = torch.stack((j[0] + fl[:, 0], j[1] + fl[:, 1], j[2] + step if forward else j[2] - step))
    xyt = torch.stack((tgt[0] / (larger_dim / 2) - 1, tgt[1] / (larger_dim / 2) - 1, tgt[2] / (nframes / 2) - 1)).T
    return uv[rows], xyt, rows


def optical_flow_loss(jif, uv, flows_rev, mask_rev, larger_dim, nframes, mapping, flows, mask, uv_mapping_scale, alpha):
    """loss_utils.py:299-322 with use_alpha=True (alpha == 1 in the single-atlas path)."""
    uv_f, xyt_f, rows_f = flow_matches(jif, mask, flows, larger_dim, nframes, True, uv)
    l_next = (mapping(xyt_f) - uv_f).norm(dim=1) * larger_dim / (2 * uv_mapping_scale)
    uv_b, xyt_b, rows_b = flow_matches(jif, mask_rev, flows_rev, larger_dim, nframes, False, uv)
    l_prev = (mapping(xyt_b) - uv_b).norm(dim=1) * larger_dim / (2 * uv_mapping_scale)
    return (l_prev * alpha[rows_b].squeeze()).mean() * 0.5 + (l_next * alpha[rows_f].squeeze()).mean() * 0.5


# ------------------------------------------------------------------------------------------------
class Video:
    """The eight dense CPU tensors load_input_data_single returns (unwrap_utils.py:105-163, Appendix B of SURVEY.md)."""

    def __init__(self, frames, flows, flows_rev, mask, mask_rev):
        self.video_frames = frames                      # (resy, resx, 3, F)
        self.resy, self.resx, _, self.F = frames.shape
        self.video_frames_dx = torch.zeros_like(frames)  # unwrap_utils.py:132-133
        self.video_frames_dy = torch.zeros_like(frames)
        self.video_frames_dy[:-1] = frames[1:] - frames[:-1]
        self.video_frames_dx[:, :-1] = frames[:, 1:] - frames[:, :-1]
        self.optical_flows = flows                      # (resy, resx, 2, F, 1)
        self.optical_flows_reverse = flows_rev
        self.optical_flows_mask = mask                  # (resy, resx, F, 1)
        self.optical_flows_reverse_mask = mask_rev
        self.larger_dim = np.maximum(self.resx, self.resy)


def loop_body(i, jif, video, mapping, atlas, config):
    """stage1_neural_atlas.py:153-227: every loss term and the weighted total for one batch `jif` (3, N, 1)."""
    c = config
    nf, L = video.F, video.larger_dim
    rgb_gt = video.video_frames[jif[1], jif[0], :, jif[2]].squeeze(1)
    xyt = torch.cat((jif[0] / (L / 2) - 1, jif[1] / (L / 2) - 1, jif[2] / (nf / 2.0) - 1), dim=1)
    uv = mapping(xyt)
    alpha = torch.ones(jif.shape[1], 1)
    rgb = (atlas(uv * 0.5 + 0.5) + 1.0) * 0.5
    if c.get("use_gradient_loss", True):     # stage1_neural_atlas.py:185-190
        grad_l = gradient_loss_single(video.video_frames_dx, video.video_frames_dy, jif, mapping, atlas, rgb, video.resx, nf)
    else:
        grad_l = torch.zeros(())
    rgb_l = (torch.norm(rgb - rgb_gt, dim=1) ** 2).mean()
    rig_l = rigidity_loss(jif, c["derivative_amount"], L, nf, mapping, uv, c["uv_mapping_scale"])
    glob = c["include_global_rigidity_loss"] and i <= c["stop_global_rigidity"]
    if glob:
        grig_l = rigidity_loss(jif, c["global_rigidity_derivative_amount_fg"], L, nf, mapping, uv, c["uv_mapping_scale"])
    flow_l = optical_flow_loss(jif, uv, video.optical_flows_reverse, video.optical_flows_reverse_mask, L, nf, mapping,
                               video.optical_flows, video.optical_flows_mask, c["uv_mapping_scale"], alpha)
    total = c["rigidity_coeff"] * rig_l + rgb_l * c["rgb_coeff"] + c["optical_flow_coeff"] * flow_l + grad_l * c["gradient_loss_coeff"]
    if glob:
        total = total + c["global_rigidity_coeff_fg"] * grig_l
    terms = {"rgb": rgb_l, "gradient": grad_l, "rigidity": rig_l,
             "global_rigidity": grig_l if glob else torch.zeros(()), "flow": flow_l, "total": total}
    return total, terms


class SingleAtlasTrainer:
    """The optimisation state of stage1_neural_atlas.main(): two nets + Adam(lr 1e-4) (:112-134)."""

    def __init__(self, config, video, seed=None, mapping=None, atlas=None):
        self.config, self.video = config, video
        if mapping is None:
            mapping, atlas = build_single_atlas_models(config, seed)
        self.mapping, self.atlas = mapping, atlas
        self.opt = torch.optim.Adam([{"params": list(mapping.parameters())}, {"params": list(atlas.parameters())}], lr=1e-4)
        self.jif_all = get_tuples(video.F, video.resy, video.resx)

    def step(self, i, inds):
        """One iteration of the loop (:159-231) with injected sample indices `inds` (N,) int64."""
        jif = self.jif_all[:, inds.view(-1, 1)]
        total, terms = loop_body(i, jif, self.video, self.mapping, self.atlas, self.config)
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        return {k: float(v.detach()) for k, v in terms.items()}

    def loss_and_grads(self, i, inds):
        jif = self.jif_all[:, inds.view(-1, 1)]
        total, terms = loop_body(i, jif, self.video, self.mapping, self.atlas, self.config)
        self.opt.zero_grad()
        total.backward()
        return {k: float(v.detach()) for k, v in terms.items()}


def pre_train_mapping(mapping, frames_num, uv_mapping_scale, resx, resy, larger_dim, pretrain_iters, ys=None, xs=None, batch=10000):
    """unwrap_utils.py:176-198.  ys/xs (steps, batch) inject the draws; None reproduces the reference's
    torch.randint order (rows first, then columns)."""
    opt = torch.optim.Adam(mapping.parameters(), lr=1e-4)
    losses, s = [], 0
    for _ in range(pretrain_iters):
        for f in range(frames_num):
            i_s = ys[s].view(-1, 1) if ys is not None else torch.randint(resy, (batch, 1))
            j_s = xs[s].view(-1, 1) if xs is not None else torch.randint(resx, (batch, 1))
            yy = i_s / (larger_dim / 2) - 1
            xx = j_s / (larger_dim / 2) - 1
            xyt = torch.cat((xx, yy, (f / (frames_num / 2.0) - 1) * torch.ones_like(yy)), dim=1)
            uv = mapping(xyt)
            mapping.zero_grad()
            loss = (xyt[:, :2] * uv_mapping_scale - uv).norm(dim=1).mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            s += 1
    return losses


# ------------------------------------------------------------------------------------------------
# render + PSNR — src/models/stage_1/evaluate.py:640-661,740-743
def render_frame(mapping, atlas, resx, resy, nframes, f, chunk=100000):
    larger_dim = np.maximum(np.int64(resx), np.int64(resy))
    ys, xs = torch.where(torch.ones(resy, resx) > 0)
    out = torch.zeros(resy, resx, 3)
    with torch.no_grad():
        n = int(np.ceil(ys.shape[0] / chunk))
        for yc, xc in zip(np.array_split(ys.numpy(), n), np.array_split(xs.numpy(), n)):
            yy = torch.from_numpy(yc).unsqueeze(1) / (larger_dim / 2) - 1
            xx = torch.from_numpy(xc).unsqueeze(1) / (larger_dim / 2) - 1
            uv = mapping(torch.cat((xx, yy, (f / (nframes / 2.0) - 1) * torch.ones_like(yy)), dim=1))
            out[yc, xc] = (atlas(uv * 0.5 + 0.5) + 1) * 0.5
    return out


def psnr(gt_f32, rec_f32):
    """skimage.metrics.peak_signal_noise_ratio(data_range=1): both images promoted to float64 (evaluate.py:740-743)."""
    a = np.asarray(gt_f32, dtype=np.float64)
    b = np.asarray(rec_f32, dtype=np.float64)
    return 10.0 * math.log10(1.0 / np.mean((a - b) ** 2))


def mean_psnr(mapping, atlas, video):
    vals = []
    for f in range(video.F):
        rec = render_frame(mapping, atlas, video.resx, video.resy, video.F, f)
        vals.append(psnr(video.video_frames[:, :, :, f].numpy(), rec.numpy()))
    return float(np.mean(vals)), vals


# ------------------------------------------------------------------------------------------------
# input builder pieces — src/models/stage_1/unwrap_utils.py:10-23
def remap_bilinear_zero(img, mapx, mapy):
    """cv2.remap(img, map, None, INTER_LINEAR) with the default constant-0 border (unwrap_utils.py:22)."""
    h, w = img.shape[:2]
    x0 = np.floor(mapx).astype(np.int64)
    y0 = np.floor(mapy).astype(np.int64)
    fx = (mapx - x0).astype(np.float32)[..., None]
    fy = (mapy - y0).astype(np.float32)[..., None]

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[..., None], v, 0.0).astype(np.float32)

    return (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy)
            + tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)


def compute_consistency(flow12, flow21):
    """unwrap_utils.py:10-23: || flow12 + warp(flow21, flow12) ||."""
    h, w = flow12.shape[:2]
    mapx = flow12[:, :, 0] + np.arange(w, dtype=np.float32)
    mapy = flow12[:, :, 1] + np.arange(h, dtype=np.float32)[:, None]
    diff = flow12 + remap_bilinear_zero(flow21, mapx, mapy)
    return (diff[:, :, 0] ** 2 + diff[:, :, 1] ** 2) ** 0.5


# ------------------------------------------------------------------------------------------------
def field_motion(rng_uniform, nframes, resx, resy):
    """Per-frame similarity motion + flow-error bumps of the `flow="field"` videos.  `rng_uniform(lo, hi, n)` draws; the order
    of the draws is part of the video's definition.  (bench._synth_video_field_device builds the same KIND of video on the device from its
    own torch-generator draws — bench.py's timed path may not import oracle/ — so the two videos are alike, not identical.)"""
    n = nframes - 1
    m = float(min(resx, resy))
    return dict(theta=rng_uniform(-0.008, 0.008, n), zoom=rng_uniform(0.994, 1.006, n), tx=rng_uniform(-2.0, 2.0, n), ty=rng_uniform(-1.2, 1.2, n),
                # a Gaussian bump of flow error (like an occlusion the flow network got wrong), one in the forward and one in the backward field
                bcx=rng_uniform(0.2 * resx, 0.8 * resx, (n, 2)), bcy=rng_uniform(0.2 * resy, 0.8 * resy, (n, 2)),
                bax=rng_uniform(-3.0, 3.0, (n, 2)), bay=rng_uniform(1.5, 3.0, (n, 2)), br=rng_uniform(0.08 * m, 0.16 * m, (n, 2)),
                # sub-pixel ripple on both fields (nothing about a real flow field is dyadic)
                rk=rng_uniform(0.05, 0.4, (n, 2)), rp=rng_uniform(0.0, 6.28, (n, 2)), ra=rng_uniform(0.02, 0.08, (n, 2)))


def field_affine(mo, i, resx, resy):
    """A_i(p) = c + zoom R(theta) (p - c) + t as the 2x3 matrix acting on (x, y, 1), in float64."""
    cx, cy = 0.5 * (resx - 1), 0.5 * (resy - 1)
    co, si = float(mo["zoom"][i]) * math.cos(float(mo["theta"][i])), float(mo["zoom"][i]) * math.sin(float(mo["theta"][i]))
    return np.array([[co, -si, cx - co * cx + si * cy + float(mo["tx"][i])], [si, co, cy - si * cx - co * cy + float(mo["ty"][i])]], np.float64)


def affine_inverse(a):
    l = np.linalg.inv(a[:, :2])
    return np.concatenate((l, -(l @ a[:, 2:3])), axis=1)


def affine_compose(a, b):
    """a after b."""
    return np.concatenate((a[:, :2] @ b[:, :2], a[:, :2] @ b[:, 2:3] + a[:, 2:3]), axis=1)


def _for_each_frame(fn, nframes):
    """Frames of a synthetic video are independent of each other and write disjoint slices: a few threads (numpy releases the GIL inside
    its loops) build them side by side — the same operations per frame in the same order, hence the same bits as the plain loop."""
    workers = min(8, os.cpu_count() or 1, nframes)
    if workers <= 1:
        for f in range(nframes):
            fn(f)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(fn, range(nframes)))


def synthetic_video(resx, resy, nframes, seed=0, vx=1.5, vy=0.5, flow="constant", flicker=True):
    """Seeded synthetic flickering video with analytic optical flow: a smooth random texture
    translating (vx, vy) px/frame, per-frame gain U(0.8,1.2) and gamma U(0.9,1.1); forward/backward flows are
    exact and the consistency masks follow the reference rule (unwrap_utils.py:151-159).

    flow="field" (round 4): the texture moves by a different similarity transform every frame (rotation, zoom, translation about the
    frame centre), so the flow VARIES PER PIXEL AND PER FRAME and no value is dyadic; each field carries a sub-pixel ripple and a
    Gaussian bump of error a few px high, so forward and backward fields are mutually consistent to < 1 px only in part of the frame
    and the masks — the reference's rule, `compute_consistency` — have holes as well as border strips.  This is the video on which a
    transposed / off-by-one flow gather or a wrongly rounded advected coordinate (loss_utils.py:339-351) shows."""
    if flow == "field":
        return _synthetic_video_field(resx, resy, nframes, seed, flicker)
    assert flow == "constant" and flicker, flow
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:resy, 0:resx].astype(np.float64)
    nwave = 12
    kx = rng.uniform(-1, 1, (nwave, 3)) * 2 * np.pi * 6 / max(resx, resy)
    ky = rng.uniform(-1, 1, (nwave, 3)) * 2 * np.pi * 6 / max(resx, resy)
    ph = rng.uniform(0, 2 * np.pi, (nwave, 3))
    amp = rng.uniform(0.3, 1.0, (nwave, 3))
    gain = rng.uniform(0.8, 1.2, nframes)
    gamma = rng.uniform(0.9, 1.1, nframes)
    frames = np.zeros((resy, resx, 3, nframes), np.float32)

    def one_frame(f):
        xs, ys = xx - f * vx, yy - f * vy
        tex = np.zeros((resy, resx, 3))
        for w in range(nwave):
            tex += amp[w] * np.sin(kx[w] * xs[..., None] + ky[w] * ys[..., None] + ph[w])
        tex = 0.5 + 0.5 * tex / amp.sum(0)
        frames[:, :, :, f] = np.clip(gain[f] * np.clip(tex, 1e-3, 1.0) ** gamma[f], 0.0, 1.0)
    _for_each_frame(one_frame, nframes)
    flows = np.zeros((resy, resx, 2, nframes, 1), np.float32)
    flows_rev = np.zeros_like(flows)
    mask = np.zeros((resy, resx, nframes, 1), np.float32)
    mask_rev = np.zeros_like(mask)
    f12 = np.zeros((resy, resx, 2), np.float32); f12[..., 0] = vx; f12[..., 1] = vy
    f21 = -f12
    m12 = (compute_consistency(f12, f21) < 1.0).astype(np.float32)
    m21 = (compute_consistency(f21, f12) < 1.0).astype(np.float32)
    for i in range(nframes - 1):
        flows[:, :, :, i, 0] = f12
        flows_rev[:, :, :, i + 1, 0] = f21
        mask[:, :, i, 0] = m12
        mask_rev[:, :, i + 1, 0] = m21
    t = torch.from_numpy
    return Video(t(frames), t(flows), t(flows_rev), t(mask), t(mask_rev))


def _synthetic_video_field(resx, resy, nframes, seed, flicker=True):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:resy, 0:resx].astype(np.float64)
    nwave = 12
    kx = rng.uniform(-1, 1, (nwave, 3)) * 2 * np.pi * 6 / max(resx, resy)
    ky = rng.uniform(-1, 1, (nwave, 3)) * 2 * np.pi * 6 / max(resx, resy)
    ph = rng.uniform(0, 2 * np.pi, (nwave, 3))
    amp = rng.uniform(0.3, 1.0, (nwave, 3))
    gain = rng.uniform(0.8, 1.2, nframes)
    gamma = rng.uniform(0.9, 1.1, nframes)
    if not flicker:                                                    # tests of the generator itself: frame f+1 warped by the flow == frame f
        gain[:], gamma[:] = 1.0, 1.0
    mo = field_motion(lambda lo, hi, n: rng.uniform(lo, hi, n), nframes, resx, resy)
    frames = np.zeros((resy, resx, 3, nframes), np.float32)
    flows = np.zeros((resy, resx, 2, nframes, 1), np.float32)
    flows_rev = np.zeros_like(flows)
    mask = np.zeros((resy, resx, nframes, 1), np.float32)
    mask_rev = np.zeros_like(mask)
    invs = [np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])]              # frame pixel -> texture coordinate, frame 0 = identity
    pairs = []
    for f in range(nframes - 1):                                       # the chain of transforms is sequential (and costs nothing) ...
        a = field_affine(mo, f, resx, resy)
        ai = affine_inverse(a)
        pairs.append((a, ai))
        invs.append(affine_compose(invs[-1], ai))

    def one_frame(f):                                                  # ... the per-pixel work of a frame depends on its own transform only
        inv = invs[f]
        xs = inv[0, 0] * xx + inv[0, 1] * yy + inv[0, 2]
        ys = inv[1, 0] * xx + inv[1, 1] * yy + inv[1, 2]
        tex = np.zeros((resy, resx, 3))
        for w in range(nwave):
            tex += amp[w] * np.sin(kx[w] * xs[..., None] + ky[w] * ys[..., None] + ph[w])
        tex = 0.5 + 0.5 * tex / amp.sum(0)
        frames[:, :, :, f] = np.clip(gain[f] * np.clip(tex, 1e-3, 1.0) ** gamma[f], 0.0, 1.0)
        if f == nframes - 1:
            return
        a, ai = pairs[f]
        f12, f21 = field_flow_pair(mo, f, a, ai, xx, yy)
        flows[:, :, :, f, 0] = f12
        flows_rev[:, :, :, f + 1, 0] = f21
        mask[:, :, f, 0] = compute_consistency(f12, f21) < 1.0             # unwrap_utils.py:151-159
        mask_rev[:, :, f + 1, 0] = compute_consistency(f21, f12) < 1.0
    _for_each_frame(one_frame, nframes)
    t = torch.from_numpy
    return Video(t(frames), t(flows), t(flows_rev), t(mask), t(mask_rev))


def field_flow_pair(mo, i, a, ai, xx, yy):
    """Forward flow of frame i (A_i p - p) and backward flow of frame i+1 (A_i^-1 p - p), each + ripple + its error bump; float32 (H, W, 2)."""
    out = []
    for j, m in enumerate((a, ai)):
        u = (m[0, 0] - 1.0) * xx + m[0, 1] * yy + m[0, 2]
        v = m[1, 0] * xx + (m[1, 1] - 1.0) * yy + m[1, 2]
        g = np.exp(-((xx - float(mo["bcx"][i, j])) ** 2 + (yy - float(mo["bcy"][i, j])) ** 2) / float(mo["br"][i, j]) ** 2)
        rip = float(mo["ra"][i, j]) * np.sin(float(mo["rk"][i, j]) * (xx + 0.7 * yy) + float(mo["rp"][i, j]))
        out.append(np.stack((u + float(mo["bax"][i, j]) * g + rip, v + float(mo["bay"][i, j]) * g - rip), axis=-1).astype(np.float32))
    return out


def flat_params(model):
    """state_dict order: hidden.0.weight, hidden.0.bias, ..."""
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy().copy()


def flat_grads(model):
    return torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()]).numpy().copy()


# ================================================================================================
# fg/bg dual-atlas path with the alpha MLP — src/stage1_neural_atlas_seg.py (BASELINE configs[4])
def build_seg_models(config, seed=None):
    """stage1_neural_atlas_seg.py:127-161 — construction (= RNG consumption) order mapping1, mapping2, atlas, alpha."""
    if seed is not None:
        torch.manual_seed(seed)
    m1 = OracleIMLP(3, 2, config["number_of_channels_mapping1"], config["use_positional_encoding_mapping1"],
                    config["number_of_positional_encoding_mapping1"], [], config["number_of_layers_mapping1"])
    m2 = OracleIMLP(3, 2, config["number_of_channels_mapping2"], config["use_positional_encoding_mapping2"],
                    config["number_of_positional_encoding_mapping2"], [], config["number_of_layers_mapping2"])
    atlas = OracleIMLP(2, 3, config["number_of_channels_atlas"], True, config["positional_encoding_num_atlas"],
                       [4, 7], config["number_of_layers_atlas"])
    alpha = OracleIMLP(3, 1, config["number_of_channels_alpha"], True, config["positional_encoding_num_alpha"],
                       [], config["number_of_layers_alpha"])
    return m1, m2, atlas, alpha


def alpha_of(model_alpha, xyt):
    """stage1_neural_atlas_seg.py:224-227: tanh output -> (0.001, 0.991)."""
    a = 0.5 * (model_alpha(xyt) + 1.0)
    a = a * 0.99
    return a + 0.001


def gradient_loss_seg(dx, dy, jif, m1, m2, atlas, rgb_out, resx, nframes, model_alpha):
    """loss_utils.py:173-224 (alpha is re-evaluated at the +1 neighbours; everything normalised by resx/2)."""
    t = jif[2] / (nframes / 2.0) - 1
    xp1 = torch.cat(((jif[0] + 1) / (resx / 2) - 1, jif[1] / (resx / 2) - 1, t), dim=1)
    yp1 = torch.cat((jif[0] / (resx / 2) - 1, (jif[1] + 1) / (resx / 2) - 1, t), dim=1)
    ax = alpha_of(model_alpha, xp1)
    ay = alpha_of(model_alpha, yp1)
    dx_gt = dx[jif[1], jif[0], :, jif[2]].squeeze(1)
    dy_gt = dy[jif[1], jif[0], :, jif[2]].squeeze(1)
    uv2_y = m2(yp1); uv2_x = m2(xp1); uv1_y = m1(yp1); uv1_x = m1(xp1)
    r1y = (atlas(uv1_y * 0.5 + 0.5) + 1.0) * 0.5
    r1x = (atlas(uv1_x * 0.5 + 0.5) + 1.0) * 0.5
    r2y = (atlas(uv2_y * 0.5 - 0.5) + 1.0) * 0.5
    r2x = (atlas(uv2_x * 0.5 - 0.5) + 1.0) * 0.5
    ry = r1y * ay + r2y * (1.0 - ay)
    rx = r1x * ax + r2x * (1.0 - ax)
    return torch.mean((dx_gt - (rx - rgb_out)).norm(dim=1) ** 2 + (dy_gt - (ry - rgb_out)).norm(dim=1) ** 2)


def optical_flow_alpha_loss(model_alpha, jif, alpha, flows_rev, mask_rev, larger_dim, nframes, flows, mask):
    """loss_utils.py:385-408 (Eq. 12): L1 between alpha at a pixel and alpha at its flow match."""
    _, xyt_f, rows_f = flow_matches(jif, mask, flows, larger_dim, nframes, True, alpha)
    a_f = alpha_of(model_alpha, xyt_f)
    l_next = (alpha[rows_f] - a_f).abs().mean()
    _, xyt_b, rows_b = flow_matches(jif, mask_rev, flows_rev, larger_dim, nframes, False, alpha)
    a_b = alpha_of(model_alpha, xyt_b)
    l_prev = (a_b - alpha[rows_b]).abs().mean()
    return (l_next + l_prev) * 0.5


SEG_TERMS = ("rgb", "gradient", "rigidity1", "rigidity2", "global_rigidity1", "global_rigidity2", "flow1", "flow2",
             "flow_alpha", "alpha_bootstrapping", "sparsity", "total")


class SegVideo(Video):
    """load_input_data outputs (unwrap_utils.py:40-103): Video + mask_frames (resy, resx, F), fractional fg mask."""

    def __init__(self, frames, flows, flows_rev, mask, mask_rev, mask_frames):
        super().__init__(frames, flows, flows_rev, mask, mask_rev)
        self.mask_frames = mask_frames


def seg_loop_body(i, jif, video, m1, m2, atlas, model_alpha, config):
    """stage1_neural_atlas_seg.py:193-311: all loss terms and the weighted total for one batch."""
    c = config
    nf, L = video.F, video.larger_dim
    boot = 0 if i > c["stop_bootstrapping_iteration"] else c["alpha_bootstrapping_factor"]
    rgb_gt = video.video_frames[jif[1], jif[0], :, jif[2]].squeeze(1)
    a_gt = video.mask_frames[jif[1], jif[0], jif[2]].squeeze(1).unsqueeze(-1)
    xyt = torch.cat((jif[0] / (L / 2) - 1, jif[1] / (L / 2) - 1, jif[2] / (nf / 2.0) - 1), dim=1)
    uv1 = m1(xyt)
    uv2 = m2(xyt)
    alpha = alpha_of(model_alpha, xyt)
    rgb1 = (atlas(uv1 * 0.5 + 0.5) + 1.0) * 0.5
    rgb2 = (atlas(uv2 * 0.5 - 0.5) + 1.0) * 0.5
    rgb = rgb1 * alpha + rgb2 * (1.0 - alpha)
    if c.get("use_gradient_loss", True):     # stage1_neural_atlas_seg.py:237-242
        grad_l = gradient_loss_seg(video.video_frames_dx, video.video_frames_dy, jif, m1, m2, atlas, rgb, video.resx, nf, model_alpha)
    else:
        grad_l = torch.zeros(())
    rgb_l = (torch.norm(rgb - rgb_gt, dim=1) ** 2).mean()
    sparse_l = (torch.norm(rgb1 * (1.0 - alpha), dim=1) ** 2).mean()
    s = c["uv_mapping_scale"]
    rig1 = rigidity_loss(jif, c["derivative_amount"], L, nf, m1, uv1, s)
    rig2 = rigidity_loss(jif, c["derivative_amount"], L, nf, m2, uv2, s)
    glob = c["include_global_rigidity_loss"] and i <= c["stop_global_rigidity"]
    zero = torch.zeros(())
    grig1 = rigidity_loss(jif, c["global_rigidity_derivative_amount_fg"], L, nf, m1, uv1, s) if glob else zero
    grig2 = rigidity_loss(jif, c["global_rigidity_derivative_amount_bg"], L, nf, m2, uv2, s) if glob else zero
    fl1 = optical_flow_loss(jif, uv1, video.optical_flows_reverse, video.optical_flows_reverse_mask, L, nf, m1,
                            video.optical_flows, video.optical_flows_mask, s, alpha)
    fl2 = optical_flow_loss(jif, uv2, video.optical_flows_reverse, video.optical_flows_reverse_mask, L, nf, m2,
                            video.optical_flows, video.optical_flows_mask, s, 1 - alpha)
    fla = optical_flow_alpha_loss(model_alpha, jif, alpha, video.optical_flows_reverse, video.optical_flows_reverse_mask,
                                  L, nf, video.optical_flows, video.optical_flows_mask)
    bce = torch.mean(-a_gt * torch.log(alpha) - (1 - a_gt) * torch.log(1 - alpha))
    total = (c["rigidity_coeff"] * (rig1 + rig2) + rgb_l * c["rgb_coeff"] + c["optical_flow_coeff"] * (fl1 + fl2)
             + bce * boot + fla * c["alpha_flow_factor"] + sparse_l * c["sparsity_coeff"] + grad_l * c["gradient_loss_coeff"])
    if glob:
        total = total + c["global_rigidity_coeff_fg"] * grig1 + c["global_rigidity_coeff_bg"] * grig2
    vals = (rgb_l, grad_l, rig1, rig2, grig1, grig2, fl1, fl2, fla, bce, sparse_l, total)
    return total, dict(zip(SEG_TERMS, vals))


class SegAtlasTrainer:
    """Optimisation state of stage1_neural_atlas_seg.main(): four nets + Adam(lr 1e-4), param-group order
    mapping1, mapping2, alpha, atlas (:165-169)."""

    def __init__(self, config, video, seed=None, models=None):
        self.config, self.video = config, video
        self.m1, self.m2, self.atlas, self.alpha = models if models is not None else build_seg_models(config, seed)
        self.opt = torch.optim.Adam([{"params": list(self.m1.parameters())}, {"params": list(self.m2.parameters())},
                                     {"params": list(self.alpha.parameters())}, {"params": list(self.atlas.parameters())}], lr=1e-4)
        self.jif_all = get_tuples(video.F, video.resy, video.resx)

    def _loss(self, i, inds):
        jif = self.jif_all[:, inds.view(-1, 1)]
        return seg_loop_body(i, jif, self.video, self.m1, self.m2, self.atlas, self.alpha, self.config)

    def step(self, i, inds):
        total, terms = self._loss(i, inds)
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        return {k: float(v.detach()) for k, v in terms.items()}

    def loss_and_grads(self, i, inds):
        total, terms = self._loss(i, inds)
        self.opt.zero_grad()
        total.backward()
        return {k: float(v.detach()) for k, v in terms.items()}


def render_frame_seg(m1, m2, atlas, model_alpha, resx, resy, nframes, f, chunk=100000):
    """evaluate.py:302-337: rgb = rgb1*alpha + rgb2*(1-alpha) for every pixel of frame f."""
    larger_dim = np.maximum(np.int64(resx), np.int64(resy))
    ys, xs = torch.where(torch.ones(resy, resx) > 0)
    out = torch.zeros(resy, resx, 3)
    with torch.no_grad():
        n = int(np.ceil(ys.shape[0] / chunk))
        for yc, xc in zip(np.array_split(ys.numpy(), n), np.array_split(xs.numpy(), n)):
            yy = torch.from_numpy(yc).unsqueeze(1) / (larger_dim / 2) - 1
            xx = torch.from_numpy(xc).unsqueeze(1) / (larger_dim / 2) - 1
            xyt = torch.cat((xx, yy, (f / (nframes / 2.0) - 1) * torch.ones_like(yy)), dim=1)
            r1 = (atlas(m1(xyt) * 0.5 + 0.5) + 1) * 0.5
            r2 = (atlas(m2(xyt) * 0.5 - 0.5) + 1) * 0.5
            a = alpha_of(model_alpha, xyt)
            out[yc, xc] = r1 * a + r2 * (1.0 - a)
    return out


def mean_psnr_seg(m1, m2, atlas, model_alpha, video):
    vals = []
    for f in range(video.F):
        rec = render_frame_seg(m1, m2, atlas, model_alpha, video.resx, video.resy, video.F, f)
        vals.append(psnr(video.video_frames[:, :, :, f].numpy(), rec.numpy()))
    return float(np.mean(vals)), vals


def synthetic_seg_video(resx, resy, nframes, seed=0, flow="constant"):
    """synthetic_video + a soft-edged disc moving across the frame as foreground mask (fractional values,
    like the bilinearly-resized masks the reference actually feeds, unwrap_utils.py:68-70) with its own colour
    texture composited over the background."""
    v = synthetic_video(resx, resy, nframes, seed=seed, flow=flow)
    rng = np.random.default_rng(seed + 1000)
    yy, xx = np.mgrid[0:resy, 0:resx].astype(np.float64)
    r = 0.22 * min(resx, resy)
    cx0, cy0 = rng.uniform(0.3, 0.4) * resx, rng.uniform(0.4, 0.6) * resy
    col = rng.uniform(0.2, 0.9, 3)
    masks = np.zeros((resy, resx, nframes), np.float32)
    frames = v.video_frames.numpy().copy()
    for f in range(nframes):
        cx, cy = cx0 + 0.8 * f, cy0 - 0.3 * f
        d = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
        m = np.clip((r - d) / 2.0 + 0.5, 0.0, 1.0)
        tex = 0.5 + 0.5 * np.sin(0.9 * (xx - cx))[..., None] * np.cos(0.7 * (yy - cy))[..., None] * col
        frames[:, :, :, f] = (m[..., None] * tex + (1 - m[..., None]) * frames[:, :, :, f]).astype(np.float32)
        masks[:, :, f] = m
    t = torch.from_numpy
    return SegVideo(t(frames), v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask,
                    v.optical_flows_reverse_mask, t(masks))
