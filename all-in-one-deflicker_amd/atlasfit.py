"""ctypes binding of libatlasfit.so (include/atlasfit.h).

`AtlasFit` mirrors the objects the reference's loop drives (src/stage1_neural_atlas.py:112-231):
two IMLP networks addressed through `state_dict()`-compatible dictionaries (keys `hidden.{i}.weight`,
`hidden.{i}.bias`, implicit_neural_networks.py:37-52), `pre_train_mapping` (unwrap_utils.py:176-198) and the
loop body.  torch is imported first so the library binds to torch's HIP runtime (one runtime per process).
"""
import ctypes as C
import os
import sys

import numpy as np

NET_MAPPING1, NET_ATLAS, NET_MAPPING2, NET_ALPHA = 0, 1, 2, 3
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libatlasfit.so")
_lib = None

# symbols declared in include/atlasfit.h (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "af_create", "af_destroy", "af_last_error", "af_upload_video", "af_param_count", "af_set_params",
    "af_get_params", "af_get_adam_state", "af_set_adam_state", "af_pretrain", "af_train_steps",
    "af_render_frame", "af_psnr", "af_sync", "af_debug_forward", "af_set_debug", "af_get_last_grads",
    "af_set_timing", "af_get_timing", "af_step_work", "af_loss_width", "af_config_size", "af_debug_records", "af_debug_plan",
    "af_resize_bilinear", "af_flow_consistency", "af_debug_dw_clocks", "af_debug_step_clocks", "af_set_dw_mode", "af_set_mlp_mode", "af_debug_dw_schedule",
    "af_debug_set_dw_cost", "af_debug_tiles", "af_get_modes",
]


AF_ERANGE = -6      # include/atlasfit.h


class AtlasFitError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("atlasfit error %d: %s" % (code, msg))
        self.code = code


class AfConfig(C.Structure):
    _fields_ = [
        ("resx", C.c_int32), ("resy", C.c_int32), ("number_of_frames", C.c_int32),
        ("samples_batch", C.c_int32),
        ("number_of_channels_mapping1", C.c_int32), ("number_of_layers_mapping1", C.c_int32),
        ("number_of_channels_atlas", C.c_int32), ("number_of_layers_atlas", C.c_int32),
        ("positional_encoding_num_atlas", C.c_int32),
        ("use_positional_encoding_mapping1", C.c_int32),
        ("derivative_amount", C.c_int32),
        ("include_global_rigidity_loss", C.c_int32),
        ("global_rigidity_derivative_amount_fg", C.c_int32),
        ("stop_global_rigidity", C.c_int32),
        ("use_gradient_loss", C.c_int32),
        ("rgb_coeff", C.c_float), ("gradient_loss_coeff", C.c_float), ("rigidity_coeff", C.c_float),
        ("optical_flow_coeff", C.c_float),
        ("global_rigidity_coeff_fg", C.c_float),
        ("uv_mapping_scale", C.c_float),
        ("lr", C.c_float),
        ("pretrain_batch", C.c_int32),
        # fg/bg dual-atlas path (src/stage1_neural_atlas_seg.py); read only when two_layer != 0
        ("two_layer", C.c_int32),
        ("number_of_channels_mapping2", C.c_int32), ("number_of_layers_mapping2", C.c_int32),
        ("number_of_channels_alpha", C.c_int32), ("number_of_layers_alpha", C.c_int32),
        ("positional_encoding_num_alpha", C.c_int32),
        ("use_positional_encoding_mapping2", C.c_int32),
        ("global_rigidity_derivative_amount_bg", C.c_int32),
        ("stop_bootstrapping_iteration", C.c_int32),
        ("global_rigidity_coeff_bg", C.c_float),
        ("alpha_bootstrapping_factor", C.c_float), ("alpha_flow_factor", C.c_float), ("sparsity_coeff", C.c_float),
        ("number_of_positional_encoding_mapping1", C.c_int32), ("number_of_positional_encoding_mapping2", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


# the shipped hyper-parameters (reference src/config/config_flow_100.json)
REFERENCE_CONFIG = {
    "maximum_number_of_frames": 200, "iters_num": 10001, "samples_batch": 10000, "optical_flow_coeff": 500.0,
    "evaluate_every": 10000, "derivative_amount": 1, "rgb_coeff": 5000, "rigidity_coeff": 1.0,
    "uv_mapping_scale": 0.8, "pretrain_mapping1": True, "pretrain_mapping2": True,
    "alpha_bootstrapping_factor": 2000.0, "alpha_flow_factor": 4900.0, "positional_encoding_num_alpha": 5,
    "number_of_channels_atlas": 256, "number_of_layers_atlas": 8, "number_of_channels_alpha": 256,
    "number_of_layers_alpha": 8, "stop_bootstrapping_iteration": 10000, "number_of_channels_mapping1": 256,
    "number_of_layers_mapping1": 6, "number_of_channels_mapping2": 256, "number_of_layers_mapping2": 4,
    "gradient_loss_coeff": 1000, "use_gradient_loss": True, "sparsity_coeff": 1000.0,
    "positional_encoding_num_atlas": 10, "use_positional_encoding_mapping1": False,
    "number_of_positional_encoding_mapping1": 4, "use_positional_encoding_mapping2": False,
    "number_of_positional_encoding_mapping2": 2, "pretrain_iter_number": 100, "load_checkpoint": False,
    "checkpoint_path": "", "include_global_rigidity_loss": True, "global_rigidity_derivative_amount_fg": 100,
    "global_rigidity_derivative_amount_bg": 100, "global_rigidity_coeff_fg": 5.0, "global_rigidity_coeff_bg": 50.0,
    "stop_global_rigidity": 5000,
}


def default_config(resx, resy, number_of_frames, config=None, two_layer=False, **over):
    """af_config from the reference's JSON config dict (keys as read at stage1_neural_atlas.py:28-90, or
    stage1_neural_atlas_seg.py:27-107 when two_layer)."""
    cfg = dict(REFERENCE_CONFIG)
    if config:
        cfg.update(config)
    cfg.update(over)
    c = AfConfig()
    c.two_layer = int(bool(two_layer))
    c.number_of_channels_mapping2 = int(cfg["number_of_channels_mapping2"])
    c.number_of_layers_mapping2 = int(cfg["number_of_layers_mapping2"])
    c.number_of_channels_alpha = int(cfg["number_of_channels_alpha"])
    c.number_of_layers_alpha = int(cfg["number_of_layers_alpha"])
    c.positional_encoding_num_alpha = int(cfg["positional_encoding_num_alpha"])
    c.use_positional_encoding_mapping2 = int(bool(cfg["use_positional_encoding_mapping2"]))
    c.number_of_positional_encoding_mapping1 = int(cfg.get("number_of_positional_encoding_mapping1", 4))
    c.number_of_positional_encoding_mapping2 = int(cfg.get("number_of_positional_encoding_mapping2", 2))
    c.global_rigidity_derivative_amount_bg = int(cfg["global_rigidity_derivative_amount_bg"])
    c.stop_bootstrapping_iteration = int(cfg["stop_bootstrapping_iteration"])
    c.global_rigidity_coeff_bg = float(cfg["global_rigidity_coeff_bg"])
    c.alpha_bootstrapping_factor = float(cfg["alpha_bootstrapping_factor"])
    c.alpha_flow_factor = float(cfg["alpha_flow_factor"])
    c.sparsity_coeff = float(cfg["sparsity_coeff"])
    c.resx, c.resy, c.number_of_frames = int(resx), int(resy), int(number_of_frames)
    c.samples_batch = int(cfg["samples_batch"])
    c.number_of_channels_mapping1 = int(cfg["number_of_channels_mapping1"])
    c.number_of_layers_mapping1 = int(cfg["number_of_layers_mapping1"])
    c.number_of_channels_atlas = int(cfg["number_of_channels_atlas"])
    c.number_of_layers_atlas = int(cfg["number_of_layers_atlas"])
    c.positional_encoding_num_atlas = int(cfg["positional_encoding_num_atlas"])
    c.use_positional_encoding_mapping1 = int(bool(cfg["use_positional_encoding_mapping1"]))
    c.derivative_amount = int(cfg["derivative_amount"])
    c.include_global_rigidity_loss = int(bool(cfg["include_global_rigidity_loss"]))
    c.global_rigidity_derivative_amount_fg = int(cfg["global_rigidity_derivative_amount_fg"])
    c.stop_global_rigidity = int(cfg["stop_global_rigidity"])
    c.use_gradient_loss = int(bool(cfg["use_gradient_loss"]))
    c.rgb_coeff = float(cfg["rgb_coeff"])
    c.gradient_loss_coeff = float(cfg["gradient_loss_coeff"])
    c.rigidity_coeff = float(cfg["rigidity_coeff"])
    c.optical_flow_coeff = float(cfg["optical_flow_coeff"])
    c.global_rigidity_coeff_fg = float(cfg["global_rigidity_coeff_fg"])
    c.uv_mapping_scale = float(cfg["uv_mapping_scale"])
    c.lr = float(cfg.get("lr", 1e-4))
    c.pretrain_batch = int(cfg.get("pretrain_batch", 10000))
    return c


def load_library(path=None):
    """dlopen libatlasfit.so.  Fails loudly if it has not been built (python build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must come first: share torch's HIP runtime)
    path = path or os.environ.get("AF_LIB_PATH") or _LIB_PATH      # AF_LIB_PATH: an experiment build of the same library (tools/ab_step.sh), never a fallback
    if not os.path.exists(path):
        raise AtlasFitError(-100, "libatlasfit.so not built: run `python %s`" % os.path.join(_HERE, "build.py"))
    lib = C.CDLL(path)
    vp, i32, i64, u64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t
    fp = C.POINTER(C.c_float)
    sig = {
        "af_create": (i32, [C.POINTER(AfConfig), i32, C.POINTER(vp)]),
        "af_destroy": (None, [vp]),
        "af_last_error": (C.c_char_p, [vp]),
        "af_upload_video": (i32, [vp, vp, vp, vp, vp, vp, vp, i32]),
        "af_param_count": (sz, [vp, i32]),
        "af_set_params": (i32, [vp, i32, vp, sz]),
        "af_get_params": (i32, [vp, i32, vp, sz]),
        "af_get_adam_state": (i32, [vp, i32, vp, vp, C.POINTER(i64)]),
        "af_set_adam_state": (i32, [vp, i32, vp, vp, i64]),
        "af_pretrain": (i32, [vp, i32, i32, vp, vp, u64, vp]),
        "af_train_steps": (i32, [vp, i32, i32, vp, u64, vp]),
        "af_render_frame": (i32, [vp, i32, vp, C.POINTER(C.c_double)]),
        "af_psnr": (i32, [vp, C.POINTER(C.c_double), vp]),
        "af_sync": (i32, [vp]),
        "af_debug_forward": (i32, [vp, i32, vp, i32, vp]),
        "af_set_debug": (i32, [vp, i32]),
        "af_get_last_grads": (i32, [vp, i32, vp, sz]),
        "af_set_timing": (i32, [vp, i32]),
        "af_get_timing": (i32, [vp, vp, vp, vp, i32]),
        "af_step_work": (i32, [vp, i32, C.POINTER(i64 * 4), C.POINTER(C.c_double)]),
        "af_loss_width": (i32, [vp]),
        "af_config_size": (sz, []),
        "af_debug_records": (i32, [vp, vp, i32, vp]),
        "af_debug_plan": (i32, [i32, i32, i32, i32, C.POINTER(i32 * 3)]),
        "af_resize_bilinear": (i32, [i32, vp, i32, i32, i32, i32, vp, i32, i32, i64, i64, i64, C.c_double, C.c_double, i32]),
        "af_flow_consistency": (i32, [i32, vp, vp, i32, i32, vp, i64, i64, C.c_float, i32]),
        "af_debug_dw_clocks": (i32, [vp, i32, vp, i32]),
        "af_debug_step_clocks": (i32, [vp, i32, vp, i32]),
        "af_debug_dw_schedule": (i32, [vp, i32, vp, i32]),
        "af_set_dw_mode": (i32, [vp, i32]),
        "af_set_mlp_mode": (i32, [vp, i32]),
        "af_debug_set_dw_cost": (i32, [vp, vp, C.c_double]),
        "af_debug_tiles": (i32, [vp, i32, i32, i32, i32, i32, i32, vp]),
        "af_get_modes": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    if lib.af_config_size() != C.sizeof(AfConfig):
        raise AtlasFitError(-101, "af_config mirror (%d B) does not match libatlasfit.so (%d B): rebuild" % (C.sizeof(AfConfig), lib.af_config_size()))
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# layer shapes of the IMLPs (stage1_neural_atlas.py:112-128; stage1_neural_atlas_seg.py:127-161; IMLP.__init__,
# implicit_neural_networks.py:16-60: skip layers [4, 7] on the atlas net only)
def imlp_shapes(net, cfg=None, pe_atlas=10, pe_alpha=5):
    """[(out_features, in_features)] per layer.  With `cfg` (an AfConfig) the widths / depths / PE sizes come from it —
    AtlasFit.__init__ checks the result against the library's own parameter count, so a config the library cannot run
    fails loudly in Python, before anything reaches the C side."""
    hid = {NET_MAPPING1: 256, NET_MAPPING2: 256, NET_ATLAS: 256, NET_ALPHA: 256}
    nl = {NET_MAPPING1: 6, NET_MAPPING2: 4, NET_ATLAS: 8, NET_ALPHA: 8}
    if cfg is not None:
        hid = {NET_MAPPING1: cfg.number_of_channels_mapping1, NET_MAPPING2: cfg.number_of_channels_mapping2,
               NET_ATLAS: cfg.number_of_channels_atlas, NET_ALPHA: cfg.number_of_channels_alpha}
        nl = {NET_MAPPING1: cfg.number_of_layers_mapping1, NET_MAPPING2: cfg.number_of_layers_mapping2,
              NET_ATLAS: cfg.number_of_layers_atlas, NET_ALPHA: cfg.number_of_layers_alpha}
        pe_atlas, pe_alpha = cfg.positional_encoding_num_atlas, cfg.positional_encoding_num_alpha
    if net not in hid:
        raise ValueError("net")
    h, L = int(hid[net]), int(nl[net])
    enc = {NET_MAPPING1: 3, NET_MAPPING2: 3, NET_ATLAS: 4 * pe_atlas, NET_ALPHA: 6 * pe_alpha}[net]
    if cfg is not None:     # a mapping net with positional encoding reads 2 * 3 * K Fourier features (implicit_neural_networks.py:28-33)
        if net == NET_MAPPING1 and cfg.use_positional_encoding_mapping1:
            enc = 6 * int(cfg.number_of_positional_encoding_mapping1)
        if net == NET_MAPPING2 and cfg.use_positional_encoding_mapping2:
            enc = 6 * int(cfg.number_of_positional_encoding_mapping2)
    out = {NET_MAPPING1: 2, NET_MAPPING2: 2, NET_ATLAS: 3, NET_ALPHA: 1}[net]
    skips = (4, 7) if net == NET_ATLAS else ()
    dims = []
    for i in range(L):
        fan_in = enc if i == 0 else (h + enc if i in skips else h)
        dims.append((out if i == L - 1 else h, fan_in))
    return dims


def flatten_state_dict(sd, net, cfg=None):
    """IMLP.state_dict() (torch tensors or arrays) -> flat fp32 vector in state_dict order."""
    parts = []
    for i, (o, k) in enumerate(imlp_shapes(net, cfg)):
        w = np.asarray(sd["hidden.%d.weight" % i].detach().cpu().numpy() if hasattr(sd["hidden.%d.weight" % i], "detach") else sd["hidden.%d.weight" % i], dtype=np.float32)
        b = np.asarray(sd["hidden.%d.bias" % i].detach().cpu().numpy() if hasattr(sd["hidden.%d.bias" % i], "detach") else sd["hidden.%d.bias" % i], dtype=np.float32)
        assert w.shape == (o, k) and b.shape == (o,), (i, w.shape, b.shape)
        parts += [w.reshape(-1), b]
    return np.concatenate(parts)


def unflatten_state_dict(flat, net, cfg=None):
    out, off = {}, 0
    for i, (o, k) in enumerate(imlp_shapes(net, cfg)):
        out["hidden.%d.weight" % i] = flat[off:off + o * k].reshape(o, k).copy(); off += o * k
        out["hidden.%d.bias" % i] = flat[off:off + o].copy(); off += o
    assert off == flat.size
    return out


# ---- input builder on the device (unwrap_utils.py:10-38,105-163) ------------------------------------------------
def _util_chk(rc):
    if rc != 0:
        raise AtlasFitError(rc, load_library().af_last_error(None).decode())


def resize_bilinear_device(src, dst, dh, dw, pix_stride, ch_stride, offset, scale=(1.0, 1.0), device=0):
    """cv2.resize(src, (dw, dh)) (INTER_LINEAR) on the GPU.  src: torch CUDA tensor (H, W, C), uint8 or float32,
    contiguous.  dst: torch CUDA float32 tensor; element (y, x, c) is written at
    dst.view(-1)[(y*dw + x)*pix_stride + c*ch_stride + offset] (so a frame can land directly in (resy,resx,C,F))."""
    import torch
    assert src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and dst.dtype == torch.float32
    assert src.dtype in (torch.uint8, torch.float32) and src.dim() == 3
    torch.cuda.synchronize()
    sh, sw, ch = src.shape
    _util_chk(load_library().af_resize_bilinear(int(device), C.c_void_p(src.data_ptr()), int(src.dtype == torch.uint8), sh, sw, ch,
                                                C.c_void_p(dst.data_ptr()), int(dh), int(dw), int(pix_stride), int(ch_stride), int(offset),
                                                float(scale[0]), float(scale[1]), 1))


def flow_consistency_device(flow12, flow21, out, pix_stride, offset, thresh=1.0, device=0):
    """unwrap_utils.py:10-23,151-159: out.view(-1)[(y*w + x)*pix_stride + offset] = (|| f12 + remap(f21, f12) || < thresh)
    as 1.0 / 0.0 (thresh <= 0: the norm itself).  flow12 / flow21: torch CUDA float32 (h, w, 2) contiguous."""
    import torch
    assert flow12.is_cuda and flow21.is_cuda and out.is_cuda and flow12.is_contiguous() and flow21.is_contiguous() and out.is_contiguous()
    assert flow12.dtype == torch.float32 and flow12.shape == flow21.shape and flow12.shape[2] == 2
    torch.cuda.synchronize()
    h, w, _ = flow12.shape
    _util_chk(load_library().af_flow_consistency(int(device), C.c_void_p(flow12.data_ptr()), C.c_void_p(flow21.data_ptr()), h, w,
                                                 C.c_void_p(out.data_ptr()), int(pix_stride), int(offset), float(thresh), 1))


class AtlasFit:
    """One video on one MI355X.  Mirrors the objects of stage1_neural_atlas.main()."""

    LOSS_NAMES = ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total", "valid_fwd", "valid_bwd")
    LOSS_NAMES_TWO_LAYER = ("rgb", "gradient", "rigidity1", "rigidity2", "global_rigidity1", "global_rigidity2", "flow1", "flow2",
                            "flow_alpha", "alpha_bootstrapping", "sparsity", "total", "valid_fwd", "valid_bwd", "_", "_")
    TIMING_NAMES = ("prep", "fwd_1", "fwd_2", "loss", "bwd_1", "bwd_2", "dw", "adam")

    def __init__(self, cfg, device=0):
        self.lib = load_library()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.af_create(C.byref(cfg), int(device), C.byref(h))
        if rc != 0:
            raise AtlasFitError(rc, self.lib.af_last_error(None).decode())
        self.h = h
        self.N = cfg.samples_batch
        self.two_layer = bool(cfg.two_layer)
        self.nets = (NET_MAPPING1, NET_MAPPING2, NET_ATLAS, NET_ALPHA) if self.two_layer else (NET_MAPPING1, NET_ATLAS)
        self.loss_width = int(self.lib.af_loss_width(h))
        for net in self.nets:                       # the Python view of the architecture must be the library's
            n = sum(o * k + o for o, k in imlp_shapes(net, cfg))
            built = self.param_count(net)
            if n != built:
                self.close()
                raise AtlasFitError(-1, "net %d: config describes %d parameters, libatlasfit.so built %d" % (net, n, built))
        self._apply_experiment_env()

    def _apply_experiment_env(self):
        """The A/B tools' environment switches, mapped onto the explicit calls (libatlasfit.so itself reads no environment):
        AF_MLP_MODE / AF_MLP_FP32, AF_DW_MODE / AF_DW_FP32 (include/atlasfit.h: af_set_mlp_mode, af_set_dw_mode) and
        AF_DW_COST="c8x8,c8x2,c8x1,c1x8,c1x2[,seg]" (af_debug_set_dw_cost; tools/dw_cost_sweep.sh).
        Honoured ONLY under AF_EXPERIMENT=1 (a stale exported AF_*MODE must not switch the arithmetic of a production run silently), every
        override is reported on stderr, and `self.arithmetic` records what is in force (stage1.py writes it into the results' config.json)."""
        env = os.environ
        m, d = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.af_get_modes(self.h, C.byref(m), C.byref(d)))
        self.arithmetic = {"mlp_mode": int(m.value), "dw_mode": int(d.value), "dw_cost": None, "overrides": []}
        asked = [k for k in ("AF_MLP_FP32", "AF_MLP_MODE", "AF_DW_FP32", "AF_DW_MODE", "AF_DW_COST") if env.get(k)]
        if not asked:
            return
        if env.get("AF_EXPERIMENT", "0") in ("", "0"):
            sys.stderr.write("[atlasfit] ignoring %s: experiment switches need AF_EXPERIMENT=1\n" % ", ".join("%s=%s" % (k, env[k]) for k in asked))
            return
        try:
            if env.get("AF_MLP_FP32") and int(env["AF_MLP_FP32"]):
                self.set_mlp_mode(0)
            elif env.get("AF_MLP_MODE"):
                self.set_mlp_mode(int(env["AF_MLP_MODE"]))
            if env.get("AF_DW_FP32") and int(env["AF_DW_FP32"]):
                self.set_dw_mode(0)
            elif env.get("AF_DW_MODE"):
                self.set_dw_mode(int(env["AF_DW_MODE"]))
            if env.get("AF_DW_COST"):
                self.set_dw_cost(env["AF_DW_COST"])
        except Exception:
            self.close()
            raise
        self.arithmetic["overrides"] = ["%s=%s" % (k, env[k]) for k in asked]
        sys.stderr.write("[atlasfit] AF_EXPERIMENT: %s -> mlp_mode %d, dw_mode %d\n" % (", ".join(self.arithmetic["overrides"]), self.arithmetic["mlp_mode"], self.arithmetic["dw_mode"]))

    def close(self):
        if getattr(self, "h", None):
            self.lib.af_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise AtlasFitError(rc, self.lib.af_last_error(self.h).decode())

    # ---- data (load_input_data_single outputs, unwrap_utils.py:105-163)
    def upload_video(self, video_frames, optical_flows, optical_flows_reverse, optical_flows_mask,
                     optical_flows_reverse_mask, mask_frames=None):
        c = self.cfg
        shp = (c.resy, c.resx, 3, c.number_of_frames)
        if hasattr(video_frames, "is_cuda") and video_frames.is_cuda:      # torch tensors already in HBM
            ts = [video_frames, optical_flows, optical_flows_reverse, optical_flows_mask, optical_flows_reverse_mask, mask_frames]
            ts = [None if t is None else t.contiguous().float() for t in ts]
            assert tuple(ts[0].shape) == shp, (tuple(ts[0].shape), shp)
            import torch
            torch.cuda.synchronize()
            ptrs = [None if t is None else C.c_void_p(t.data_ptr()) for t in ts]
            self._chk(self.lib.af_upload_video(self.h, *ptrs, 1))
            return
        arrs = [video_frames, optical_flows, optical_flows_reverse, optical_flows_mask, optical_flows_reverse_mask, mask_frames]
        arrs = [None if a is None else _f32(a.numpy() if hasattr(a, "numpy") else a) for a in arrs]
        assert arrs[0].shape == shp, (arrs[0].shape, shp)
        assert arrs[1].size == c.resy * c.resx * 2 * c.number_of_frames
        assert arrs[3].size == c.resy * c.resx * c.number_of_frames
        self._chk(self.lib.af_upload_video(self.h, *[_ptr(a) for a in arrs], 0))

    # ---- parameters
    def param_count(self, net):
        return int(self.lib.af_param_count(self.h, net))

    def load_state_dict(self, net, sd):
        flat = flatten_state_dict(sd, net, self.cfg)
        self._chk(self.lib.af_set_params(self.h, net, _ptr(flat), flat.size))

    def state_dict(self, net):
        flat = np.empty(self.param_count(net), np.float32)
        self._chk(self.lib.af_get_params(self.h, net, _ptr(flat), flat.size))
        return unflatten_state_dict(flat, net, self.cfg)

    def get_params_flat(self, net):
        flat = np.empty(self.param_count(net), np.float32)
        self._chk(self.lib.af_get_params(self.h, net, _ptr(flat), flat.size))
        return flat

    def adam_state(self, net):
        n = self.param_count(net)
        m, v, step = np.empty(n, np.float32), np.empty(n, np.float32), C.c_int64(0)
        self._chk(self.lib.af_get_adam_state(self.h, net, _ptr(m), _ptr(v), C.byref(step)))
        return m, v, int(step.value)

    def set_adam_state(self, net, m, v, step):
        m, v = _f32(m), _f32(v)
        self._chk(self.lib.af_set_adam_state(self.h, net, _ptr(m), _ptr(v), int(step)))

    # ---- pre_train_mapping (unwrap_utils.py:176-198)
    def pre_train_mapping(self, pretrain_iters, ys=None, xs=None, seed=0, net=NET_MAPPING1, return_losses=False):
        steps = pretrain_iters * self.cfg.number_of_frames
        losses = np.zeros(steps, np.float32) if return_losses else None
        if ys is not None:
            ys = np.ascontiguousarray(ys, np.int64); xs = np.ascontiguousarray(xs, np.int64)
            assert ys.size == steps * self.cfg.pretrain_batch == xs.size
        self._chk(self._range_fallback(self.lib.af_pretrain(self.h, net, int(pretrain_iters), _ptr(ys), _ptr(xs), int(seed), _ptr(losses)), "af_pretrain"))
        return losses

    # ---- the loop body (stage1_neural_atlas.py:151-231)
    def train_steps(self, first_iter, n_iters, inds=None, seed=0, return_losses=True):
        losses = np.zeros((n_iters, self.loss_width), np.float32) if return_losses else None
        if inds is not None:
            inds = np.ascontiguousarray(inds, np.int64)
            assert inds.size == n_iters * self.N, (inds.shape, n_iters, self.N)
        self._chk(self._range_fallback(self.lib.af_train_steps(self.h, int(first_iter), int(n_iters), _ptr(inds), int(seed), _ptr(losses)), "af_train_steps"))
        return losses

    range_fallback = False

    def _range_fallback(self, rc, what):
        """AF_ERANGE (include/atlasfit.h): in the f16x3 arithmetic k_adam saw a hidden-layer weight at |w| >= 8 — half of what the fp16 weight images
        hold.  The steps of the call that reports it are complete and valid (the images are finite below 16 and Adam moves a weight by <= lr per step;
        the losses are already written), so a caller that must not stop — the stage-1 CLIs set `range_fallback = True` — goes on from the same state
        on the bf16x6 chains, which have no range limit; said on stderr and recorded in `self.arithmetic` (-> config.json).  Off by default: the
        library's own behaviour is the error."""
        if rc != AF_ERANGE or not self.range_fallback:
            return rc
        msg = self.lib.af_last_error(self.h).decode()
        self.set_mlp_mode(1)
        self.arithmetic.setdefault("overrides", []).append("range fallback after %s: mlp_mode 3 -> 1" % what)
        sys.stderr.write("[atlasfit] %s -- continuing from the same state on the bf16x6 chains (af_set_mlp_mode(h, 1))\n" % msg)
        return 0

    # ---- evaluate_model_single core (evaluate.py:640-743)
    def render_frame(self, f):
        rgb = np.empty((self.cfg.resy, self.cfg.resx, 3), np.float32)
        sse = C.c_double(0)
        self._chk(self.lib.af_render_frame(self.h, int(f), _ptr(rgb), C.byref(sse)))
        return rgb, float(sse.value)

    def psnr(self):
        per = np.zeros(self.cfg.number_of_frames, np.float64)
        mean = C.c_double(0)
        self._chk(self.lib.af_psnr(self.h, C.byref(mean), _ptr(per)))
        return float(mean.value), per

    # ---- hooks
    def debug_forward(self, net, rows):
        rows = _f32(rows)
        assert rows.ndim == 2 and rows.shape[1] == 4
        out = np.empty_like(rows)
        self._chk(self.lib.af_debug_forward(self.h, net, _ptr(rows), rows.shape[0], _ptr(out)))
        return out

    def debug_tiles(self, net, which, layer, rows, tile0=0, ntiles=None):
        """Tensors the last training step left for the weight-gradient GEMMs (include/atlasfit.h af_debug_tiles): which = "acts" (plane `layer` =
        relu(Z_layer), (ntiles, 256, 32)), "dz" (dZ_layer), "masks" ((ntiles, 64, 4) uint32), "pe" ((ntiles, 64, 32)), "dz_last" / "x0" ((ntiles, 32, 32)).
        `rows` = rows of that net's batch in the step (its planes are ceil(rows / 32) tiles apart)."""
        kinds = {"acts": (0, (256, 32)), "dz": (1, (256, 32)), "masks": (2, (64, 4)), "pe": (3, (64, 32)), "dz_last": (4, (32, 32)), "x0": (5, (32, 32))}
        code, shp = kinds[which]
        stride = (int(rows) + 31) // 32
        ntiles = stride - tile0 if ntiles is None else int(ntiles)
        out = np.empty((ntiles,) + shp, np.float32)
        self._chk(self.lib.af_debug_tiles(self.h, net, code, int(layer), stride, int(tile0), ntiles, _ptr(out)))
        return out.view(np.uint32) if which == "masks" else out

    def read_records(self, inds):
        """(n, 16) packed pixel records for pixel-frame indices `inds` (column numbers of get_tuples' table)."""
        inds = np.ascontiguousarray(inds, np.int64)
        out = np.empty((inds.size, 16), np.float32)
        self._chk(self.lib.af_debug_records(self.h, _ptr(inds), inds.size, _ptr(out)))
        return out

    def dw_clocks(self, enable=True):
        """(#CUs, 2) start / end s_memrealtime ticks (100 MHz) per workgroup of the most recent k_dw launch."""
        out = np.zeros((1024, 2), np.uint64)
        n = self.lib.af_debug_dw_clocks(self.h, int(enable), _ptr(out), 1024)
        if n < 0:
            self._chk(n)
        return out[:n]

    STEP_CLOCK_LAUNCHES = ("fwd_1", "fwd_2", "bwd_1", "bwd_2", "dw")

    def step_clocks(self, enable=True):
        """{launch: (#workgroups, 4) uint64 = s_memrealtime (100 MHz), s_memtime (shader clock) at workgroup start, then at its end} of
        the most recent training step for the five hot launches (include/atlasfit.h: af_debug_step_clocks); the first call enables the stamps."""
        out = np.zeros((5, 4096, 4), np.uint64)
        n = self.lib.af_debug_step_clocks(self.h, int(enable), _ptr(out), 4096)
        if n < 0:
            self._chk(n)
        return {name: out[i][out[i, :, 2] > 0] for i, name in enumerate(self.STEP_CLOCK_LAUNCHES)}

    def dw_schedule(self, which):
        """(#workgroups, 16, 4) int32 segments {shape, t0, t1, job} of k_dw's static schedule `which` (0: 9 segments, 1: 7)."""
        n = self.lib.af_debug_dw_schedule(self.h, int(which), None, 0)
        if n < 0:
            self._chk(n)
        out = np.zeros((n, 16, 4), np.int32)
        self.lib.af_debug_dw_schedule(self.h, int(which), _ptr(out), n)
        return out

    def set_dw_mode(self, mode):
        """k_dw arithmetic: 1 = bf16x6 (fp32-faithful; the default), 2 = bf16x3 (two bf16 per operand, three products: narrower than fp32, opt-in), 0 = fp32 MFMA."""
        self._chk(self.lib.af_set_dw_mode(self.h, int(mode)))
        if hasattr(self, "arithmetic"):
            self.arithmetic["dw_mode"] = int(mode)

    def set_dw_cost(self, row, seg_cost=0.0):
        """Another split-K partition of k_dw (af_debug_set_dw_cost): row = five tile costs (8x8, 8x2, 8x1, 1x8, 1x2), a sequence or the
        "a,b,c,d,e[,seg]" string of the sweep tools; None = the shipped row.  Same arithmetic, another summation order."""
        if row is None:
            self._chk(self.lib.af_debug_set_dw_cost(self.h, None, 0.0))
            return
        if isinstance(row, str):
            vals = [float(x) for x in row.split(",")]
            if len(vals) == 6:
                seg_cost = vals[5]
            row = vals[:5]
        if len(row) != 5:
            raise ValueError("set_dw_cost: five tile costs (8x8, 8x2, 8x1, 1x8, 1x2)")
        arr = (C.c_double * 5)(*[float(x) for x in row])
        self._chk(self.lib.af_debug_set_dw_cost(self.h, arr, float(seg_cost)))

    def set_mlp_mode(self, mode):
        """Hidden-layer products of the MLP chains: 3 = f16x3 on the fp16 matrix pipe (default since round 6), 1 = bf16x6 on the bf16 matrix pipe,
        0 = fp32 MFMA (cross-check), 2 = bf16x6 forward, three-product bf16 backward chain (experiment)."""
        self._chk(self.lib.af_set_mlp_mode(self.h, int(mode)))
        if hasattr(self, "arithmetic"):
            self.arithmetic["mlp_mode"] = int(mode)

    def set_debug(self, on=True):
        self._chk(self.lib.af_set_debug(self.h, int(on)))

    def last_grads(self, net):
        g = np.empty(self.param_count(net), np.float32)
        self._chk(self.lib.af_get_last_grads(self.h, net, _ptr(g), g.size))
        return g

    def set_timing(self, mask=0xFFFF, every=1):
        """HIP events around the launch classes in `mask`; `every` = P: only every P-th step of a train_steps call is timed (an event
        costs ~5 us in-stream)."""
        m = int(mask) if not isinstance(mask, bool) else (0xFFFF if mask else 0)
        if not 1 <= int(every) <= 255:          # the period travels in bits 16..23 of af_set_timing's argument: 256 would wrap to "every step"
            raise ValueError("set_timing: every must be 1..255, got %r" % (every,))
        self._chk(self.lib.af_set_timing(self.h, (m & 0xFFFF) | ((max(1, int(every)) & 0xFF) << 16)))

    def timing(self, reset=True):
        """{launch class: (milliseconds, launches, algorithmic FLOPs)} accumulated since the last reset."""
        ms = np.zeros(16, np.float64); cnt = np.zeros(16, np.int64); fl = np.zeros(16, np.float64)
        self._chk(self.lib.af_get_timing(self.h, _ptr(ms), _ptr(cnt), _ptr(fl), int(reset)))
        return {n: (float(m), int(c), float(f)) for n, m, c, f in zip(self.TIMING_NAMES, ms, cnt, fl)}

    def step_work(self, it):
        """(rows per net indexed by NET_*, fwd+bwd FLOPs) of one loop iteration."""
        rows, fl = (C.c_int64 * 4)(), C.c_double(0)
        self._chk(self.lib.af_step_work(self.h, int(it), C.byref(rows), C.byref(fl)))
        return [int(r) for r in rows], float(fl.value)

    def sync(self):
        self._chk(self.lib.af_sync(self.h))
