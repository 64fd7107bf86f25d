"""Python owner of an `lns_engine` handle (include/lns.h).

Translates the reference's `args` namespace into `lns_config`, exposes the
engine's parameter table (= the reference state_dict keys/shapes) and runs the
hot path on CUDA/HIP tensors.  torch is used for device memory and streams only.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (LNS_AE_HALF_PERIODIC, LNS_AE_NONE, LNS_AE_NONSQUARED, LNS_AE_SQUARE,
                   LNS_PAD_CIRCULAR, LNS_PAD_ZEROS, LNS_PROP_CONDITIONAL, LNS_PROP_NONE,
                   LNS_PROP_PLAIN, LnsConfig, LnsError)

# family -> (ae kind, propagator kind)
_FAMILIES = {
    "ns2d": (LNS_AE_SQUARE, LNS_PROP_PLAIN),
    "sw_half_periodic": (LNS_AE_HALF_PERIODIC, LNS_PROP_PLAIN),
    "sw_nonsquared": (LNS_AE_NONSQUARED, LNS_PROP_PLAIN),
    "twophase": (LNS_AE_NONSQUARED, LNS_PROP_PLAIN),
    "twophase_cond": (LNS_AE_NONSQUARED, LNS_PROP_CONDITIONAL),
}


def _fill_list(cfg, name, values):
    values = list(values or [])
    if len(values) > _lib.LNS_MAX_STAGES:
        raise ValueError("%s: at most %d entries" % (name, _lib.LNS_MAX_STAGES))
    arr = getattr(cfg, name)
    for i, v in enumerate(values):
        arr[i] = int(v)
    setattr(cfg, "n_" + name, len(values))


def make_config(args, ae_kind=None, prop_kind=None, ae_prefix="", prop_prefix="",
                prop_pad=None) -> LnsConfig:
    """`args` carries the reference's YAML keys (modules/autoencoder2d.py:19-27,78-92;
    train_stage2_ns2d.py:94-104).  `family` (optional) selects the AE / propagator files."""
    fam = getattr(args, "family", None)
    if ae_kind is None or prop_kind is None:
        if fam not in _FAMILIES:
            raise ValueError("args.family must be one of %s" % sorted(_FAMILIES))
        ak, pk = _FAMILIES[fam]
        ae_kind = ak if ae_kind is None else ae_kind
        prop_kind = pk if prop_kind is None else prop_kind
    c = LnsConfig()
    c.abi_version = _lib.LNS_ABI_VERSION
    c.ae_kind, c.prop_kind = ae_kind, prop_kind
    c.latent_dim = int(args.latent_dim)
    if ae_kind != LNS_AE_NONE:
        c.in_channels = int(args.in_channels)
        c.Ly, c.Lx = int(args.Ly), int(args.Lx)
        if ae_kind == LNS_AE_SQUARE:
            c.res_h = c.res_w = int(args.resolution)
            heads, dim = args.attn_heads, args.attn_dim
            per = bool(args.is_periodic)
            c.ae_pad_y = c.ae_pad_x = LNS_PAD_CIRCULAR if per else LNS_PAD_ZEROS
            c.use_attn_enc = int(bool(getattr(args, "use_attn_enc", False)))
        else:
            c.res_h, c.res_w = int(args.resolutions[0]), int(args.resolutions[1])
            heads, dim = args.decoder_attn_heads, args.decoder_attn_dim
            c.hw_ratio = float(getattr(args, "hw_ratio", c.res_w / c.res_h))
            if ae_kind == LNS_AE_HALF_PERIODIC:
                pd = args.periodic_direction
                if pd not in ("x", "y"):
                    raise ValueError("periodic_direction must be x or y")
                c.ae_pad_y = LNS_PAD_CIRCULAR if pd == "y" else LNS_PAD_ZEROS
                c.ae_pad_x = LNS_PAD_CIRCULAR if pd == "x" else LNS_PAD_ZEROS
            else:
                per = bool(args.is_periodic)
                c.ae_pad_y = c.ae_pad_x = LNS_PAD_CIRCULAR if per else LNS_PAD_ZEROS
        c.latent_resolution = int(args.latent_resolution)
        _fill_list(c, "encoder_channels", args.encoder_channels)
        _fill_list(c, "decoder_channels", args.decoder_channels)
        _fill_list(c, "attn_resolutions", args.attn_resolutions)
        _fill_list(c, "fourier_resolutions", getattr(args, "fourier_resolutions", []))
        c.encoder_res_blocks = int(args.encoder_res_blocks)
        c.decoder_res_blocks = int(args.decoder_res_blocks)
        c.use_fa = int(bool(args.use_fa))
        c.final_smoothing = int(bool(args.final_smoothing))
        dca = getattr(args, "disable_coarse_attn", None)
        c.disable_coarse_attn = int(bool(dca)) if dca is not None else 0
        c.attn_heads, c.attn_dim = int(heads), int(dim)
    if prop_kind != LNS_PROP_NONE:
        c.prop_n_block = int(args.prop_n_block)
        c.prop_n_embd = int(args.prop_n_embd)
        c.prop_dilation = int(args.dilation)
        if prop_pad is None:
            # train_stage2_ns2d.py:75 circular; train_stage2_SW.py:76 periodic_direction='x';
            # train_stage2_twophase.py:76 / _conditional.py:106 zeros
            prop_pad = {"ns2d": (LNS_PAD_CIRCULAR, LNS_PAD_CIRCULAR),
                        "sw_half_periodic": (LNS_PAD_ZEROS, LNS_PAD_CIRCULAR),
                        "sw_nonsquared": (LNS_PAD_ZEROS, LNS_PAD_CIRCULAR)}.get(
                            fam, (LNS_PAD_ZEROS, LNS_PAD_ZEROS))
        c.prop_pad_y, c.prop_pad_x = prop_pad
        c.cond_emb_dim = int(getattr(args, "cond_emb_dim", args.latent_dim))
    if getattr(args, "cond_encoder", False):
        # ConditionalSimpleAutoencoder (modules/autoencoder2d_nonsquared.py:279-305): CondEncoder + plain Decoder
        if ae_kind != LNS_AE_NONSQUARED:
            raise ValueError("cond_encoder is defined for the non-squared autoencoder only")
        c.cond_encoder = 1
        c.cond_emb_channels = int(args.cond_emb_channels)
    c.ae_prefix = ae_prefix.encode()
    c.prop_prefix = prop_prefix.encode()
    return c


class Engine:
    """One lns_engine handle (one per GPU; not thread-safe)."""

    def __init__(self, cfg: LnsConfig):
        L = _lib.lib()
        self._L = L
        self.cfg = cfg
        h = ctypes.c_void_p()
        rc = L.lns_create(ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            raise LnsError("lns_create failed: %s" % L.lns_create_error().decode())
        self._h = h
        self._ws = {}
        self.device_index = None
        self.params = self._param_table()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._L.lns_destroy(h)
            except Exception:
                pass

    def _check(self, rc, what):
        if rc != 0:
            raise LnsError("%s failed (%d): %s" % (what, rc, self._L.lns_last_error(self._h).decode()))

    def _param_table(self):
        out = []
        n = self._L.lns_num_params(self._h)
        key = ctypes.create_string_buffer(_lib_key_cap())
        shape = (ctypes.c_int64 * 8)()
        nd, isb = ctypes.c_int(), ctypes.c_int()
        for i in range(n):
            self._check(self._L.lns_param_info(self._h, i, key, len(key), shape, ctypes.byref(nd),
                                               ctypes.byref(isb)), "lns_param_info")
            out.append((key.value.decode(), tuple(int(shape[j]) for j in range(nd.value)), bool(isb.value)))
        return out

    def param_shapes(self):
        return {k: s for k, s, _ in self.params}

    # -- weights ---------------------------------------------------------------
    def load_weights(self, weights: dict, device_index: int):
        """weights: {key: float32 ndarray}; all keys of the table must be present."""
        for key, shape, _ in self.params:
            if key not in weights:
                raise KeyError("missing key in state_dict: %s" % key)
            a = np.ascontiguousarray(weights[key], dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError("size mismatch for %s: %s vs %s" % (key, a.shape, shape))
            shp = (ctypes.c_int64 * 8)(*shape)
            self._check(self._L.lns_set_weight(self._h, key.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                               shp, len(shape)), "lns_set_weight")
        self._check(self._L.lns_finalize_weights(self._h, int(device_index)), "lns_finalize_weights")
        self.device_index = int(device_index)
        self._ws.clear()

    def set_option(self, name, value):
        """Scheduling options of the rollout (include/lns.h lns_set_option): decode_group, decode_streams, overlap,
        prop_priority, track_nonfinite, fa_chunk_mb, fa_fused_gpb.  Results never depend on them.  "fa_fused" (default 2; 1 = single-buffered kernel, same bits; 0 = off) selects
        the arithmetic form of FABlock2D at 64 x 64 planes (in_proj inside the sandwich kernel): ~2e-7 relative on the fields."""
        self._check(self._L.lns_set_option(self._h, name.encode(), int(value)), "lns_set_option")
        self._ws.clear()                      # the workspace size depends on the options

    def latent_shape(self):
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._check(self._L.lns_latent_shape(self._h, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)),
                    "lns_latent_shape")
        return c.value, h.value, w.value

    # -- execution ---------------------------------------------------------------
    def _workspace(self, B, device, min_bytes=0):
        import torch
        n = ctypes.c_size_t(0)
        if self.cfg.ae_kind != LNS_AE_NONE:
            self._check(self._L.lns_prepare(self._h, int(B), ctypes.byref(n)), "lns_prepare")
        need = max(int(n.value), int(min_bytes), 1 << 20)
        ws = self._ws.get((B, device))
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws[(B, device)] = ws
        return ws

    @staticmethod
    def _stream(t):
        """The current HIP stream of the device `t` lives on (not of whatever device is current)."""
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def _param(self, param, like):
        """Conditional propagator: one normalised parameter per trajectory -> fp32 [B] on like's device."""
        import torch
        if param is None:
            return None
        B = like.shape[0]
        p = torch.as_tensor(param)
        if p.numel() != B:
            raise LnsError("param must hold one value per trajectory: got %d values for batch %d" % (p.numel(), B))
        if p.device != like.device:
            raise LnsError("param is on %s but the fields are on %s" % (p.device, like.device))
        return p.reshape(B).to(torch.float32).contiguous()

    @staticmethod
    def _dev(t):
        import torch
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise LnsError("the LNS engine runs on HIP device tensors only (got %s); there is no CPU "
                           "fallback" % (t.device if hasattr(t, "device") else type(t)))
        if t.dtype != torch.float32:
            raise LnsError("fp32 tensors expected, got %s" % t.dtype)
        return t.contiguous()

    def encode(self, x, param=None, scale_shift=None):
        """scale_shift [B, in_channels, 2] (device): encode(x * scale + shift) with the affine map applied in the first
        convolution's prologue (lns_encode_affine; the dataset normalisation of encode_dataset)."""
        import torch
        x = self._dev(x)
        B = x.shape[0]
        C, H, W = self.latent_shape()
        z = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        ws = self._workspace(B, x.device)
        if scale_shift is not None:
            ss = self._dev(scale_shift)
            if tuple(ss.shape) != (B, self.cfg.in_channels, 2) or ss.device != x.device:
                raise LnsError("scale_shift must be [B, in_channels, 2] on the input's device")
            if bool(self.cfg.cond_encoder) != (param is not None):
                raise LnsError("param must be given exactly for a conditional encoder")
            p = self._param(param, x) if param is not None else None
            self._check(self._L.lns_encode_affine(self._h, x.data_ptr(), ss.data_ptr(), p.data_ptr() if p is not None else None,
                                                  B, z.data_ptr(), ws.data_ptr(), ws.numel(), self._stream(x)),
                        "lns_encode_affine")
            return z
        if self.cfg.cond_encoder:
            if param is None:
                raise LnsError("this autoencoder's encoder is conditional: encode(x, param)")
            p = self._param(param, x)
            self._check(self._L.lns_encode_cond(self._h, x.data_ptr(), p.data_ptr(), B, z.data_ptr(), ws.data_ptr(),
                                                ws.numel(), self._stream(x)), "lns_encode_cond")
            return z
        self._check(self._L.lns_encode(self._h, x.data_ptr(), B, z.data_ptr(), ws.data_ptr(), ws.numel(),
                                       self._stream(x)), "lns_encode")
        return z

    def decode(self, z):
        import torch
        z = self._dev(z)
        B = z.shape[0]
        c = self.cfg
        y = torch.empty((B, c.in_channels, c.Ly, c.Lx), dtype=torch.float32, device=z.device)
        ws = self._workspace(B, z.device)
        self._check(self._L.lns_decode(self._h, z.data_ptr(), B, y.data_ptr(), ws.data_ptr(), ws.numel(),
                                       self._stream(z)), "lns_decode")
        return y

    def propagate(self, z, param=None):
        import torch
        z = self._dev(z)
        B, C, H, W = z.shape
        out = torch.empty_like(z)
        p = self._param(param, z)
        # propagator-only engines have no lns_prepare(): size generously from the activations
        ws = self._workspace(B, z.device, min_bytes=64 * B * max(C, self.cfg.prop_n_embd) * H * W * 4 + (1 << 22))
        self._check(self._L.lns_propagate(self._h, z.data_ptr(), p.data_ptr() if p is not None else None,
                                          B, H, W, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                          self._stream(z)), "lns_propagate")
        return out

    def rollout(self, x, steps, param=None, to_x=True, return_latents=False, out=None):
        import torch
        x = self._dev(x)
        B = x.shape[0]
        c = self.cfg
        C, H, W = self.latent_shape()
        shape = (B, steps, c.in_channels, c.Ly, c.Lx) if to_x else (B, steps, C, H, W)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != shape or not out.is_contiguous():
            raise LnsError("preallocated output must be contiguous with shape %s" % (shape,))
        lat = torch.empty((B, steps, C, H, W), dtype=torch.float32, device=x.device) if return_latents else None
        p = self._param(param, x)
        ws = self._workspace(B, x.device)
        self._check(self._L.lns_rollout(self._h, x.data_ptr(), p.data_ptr() if p is not None else None, B,
                                        int(steps), int(bool(to_x)), out.data_ptr(),
                                        lat.data_ptr() if lat is not None else None, ws.data_ptr(),
                                        ws.numel(), self._stream(x)), "lns_rollout")
        return (out, lat) if return_latents else out

    def rollout_latent(self, z, steps, param=None, to_x=True, out=None):
        """Continue from latent z: returns (out [B,steps,...], z after the last step)."""
        import torch
        z = self._dev(z)
        B = z.shape[0]
        c = self.cfg
        C, H, W = self.latent_shape()
        shape = (B, steps, c.in_channels, c.Ly, c.Lx) if to_x else (B, steps, C, H, W)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=z.device)
        elif tuple(out.shape) != shape or not out.is_contiguous():
            raise LnsError("preallocated output must be contiguous with shape %s" % (shape,))
        z_last = torch.empty_like(z)
        p = self._param(param, z)
        ws = self._workspace(B, z.device)
        self._check(self._L.lns_rollout_latent(self._h, z.data_ptr(), p.data_ptr() if p is not None else None, B,
                                               int(steps), int(bool(to_x)), out.data_ptr(), z_last.data_ptr(),
                                               ws.data_ptr(), ws.numel(), self._stream(z)), "lns_rollout_latent")
        return out, z_last

    # -- training rollout of the propagator (include/lns.h "training rollout") ------------------------------------
    def _ptr_array(self, tensors):
        """ctypes array of device pointers in parameter-table order; tensors: {key: fp32 contiguous device tensor}."""
        arr = (ctypes.c_void_p * len(self.params))()
        for i, (key, shape, _) in enumerate(self.params):
            t = tensors.get(key)
            if t is None:
                arr[i] = None
                continue
            if not t.is_cuda or t.dtype is not __import__("torch").float32 or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
                raise LnsError("training rollout: %s must be a contiguous fp32 device tensor of shape %s" % (key, tuple(shape)))
            arr[i] = t.data_ptr()
        return arr

    def train_forward(self, params, z_in, T, param=None):
        """z_pred [B,T,c,h,w] of the latent rollout started at z_in [B,c,h,w], with the tape kept for train_backward.
        params: {state_dict key: device tensor} of (at least) the propagator's parameters; param: [B] (conditional)."""
        import torch
        z_in = self._dev(z_in)
        pc = self._param(param, z_in)
        B, c, h, w = z_in.shape
        n = ctypes.c_size_t(0)
        self._check(self._L.lns_train_workspace_bytes(self._h, B, h, w, int(T), ctypes.byref(n)), "lns_train_workspace_bytes")
        ws = torch.empty(int(n.value), dtype=torch.uint8, device=z_in.device)
        z_pred = torch.empty((B, int(T), c, h, w), dtype=torch.float32, device=z_in.device)
        self._check(self._L.lns_train_forward(self._h, self._ptr_array(params), z_in.data_ptr(),
                                              pc.data_ptr() if pc is not None else None, B, h, w, int(T), z_pred.data_ptr(),
                                              ws.data_ptr(), ws.numel(), self._stream(z_in)), "lns_train_forward")
        return z_pred, ws

    def train_backward(self, params, z_in, z_pred, grad_z_pred, ws, need_z_grad=False):
        """{key: gradient tensor} of every propagator parameter (+ grad of z_in if asked) for dL/dz_pred."""
        import torch
        z_in = self._dev(z_in)
        g = self._dev(grad_z_pred)
        B, T, c, h, w = z_pred.shape
        prefix = self.cfg.prop_prefix.decode()
        grads = {k: torch.empty(tuple(shp), dtype=torch.float32, device=z_in.device)
                 for k, shp, _ in self.params if k.startswith(prefix)}
        gz = torch.empty_like(z_in) if need_z_grad else None
        self._check(self._L.lns_train_backward(self._h, self._ptr_array(params), z_in.data_ptr(), z_pred.data_ptr(), g.data_ptr(),
                                               B, h, w, T, self._ptr_array(grads), gz.data_ptr() if gz is not None else None,
                                               ws.data_ptr(), ws.numel(), self._stream(z_in)), "lns_train_backward")
        return grads, gz

    def check_finite(self, B, device=None):
        """Raises LnsError naming the first layer / sample whose output held inf or NaN in the LAST call (encode /
        decode / propagate / rollout) for batch B.  Coverage: every tensor a layer of that call wrote, including
        the plan outputs; for a rollout the amax records are per plan RUN, so what is seen is the last propagator
        step and the last decode on each decode stream -- `set_option("track_nonfinite", 1)` adds a sticky word that
        also remembers the earlier steps / decode groups (one tiny extra launch per plan run).
        Synchronises the device's current stream; reads a few KB of the workspace back."""
        import torch
        dev = torch.device("cuda" if device is None else device)
        if dev.type != "cuda":
            raise LnsError("check_finite: %s is not a HIP device" % (dev,))
        dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        ws = self._ws.get((B, dev))
        if ws is None:
            raise LnsError("no run for batch %d on %s yet" % (B, dev))
        stream = ctypes.c_void_p(torch.cuda.current_stream(ws.device).cuda_stream)
        self._check(self._L.lns_check_finite(self._h, int(B), ws.data_ptr(), ws.numel(), stream), "lns_check_finite")

    # -- diagnostics ----------------------------------------------------------------
    def trace_enable(self, on=True):
        self._check(self._L.lns_trace_enable(self._h, int(on)), "lns_trace_enable")

    def trace(self):
        out = []
        name = ctypes.create_string_buffer(_lib_key_cap())
        shp = (ctypes.c_int64 * 4)()
        for i in range(self._L.lns_trace_count(self._h)):
            self._check(self._L.lns_trace_info(self._h, i, name, len(name), shp), "lns_trace_info")
            a = np.empty(tuple(int(s) for s in shp), np.float32)
            self._check(self._L.lns_trace_copy(self._h, i, a.ctypes.data_as(ctypes.c_void_p)), "lns_trace_copy")
            out.append((name.value.decode(), a))
        return out

    def timing_enable(self, on=True):
        self._check(self._L.lns_timing_enable(self._h, int(on)), "lns_timing_enable")

    def timing_event_overhead_us(self):
        """Per-launch overhead the engine measured for its HIP-event timing and subtracted from every timed launch."""
        v = ctypes.c_double()
        try:
            self._check(self._L.lns_timing_mfma_flops(self._h, -1, ctypes.byref(v)), "lns_timing_mfma_flops")
        except _lib.LnsLibraryError:
            return None
        return v.value if v.value >= 0 else None

    def timing(self):
        out = {}
        name = ctypes.create_string_buffer(_lib_key_cap())
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_int64()
        for i in range(self._L.lns_timing_count(self._h)):
            self._check(self._L.lns_timing_info(self._h, i, name, len(name), ctypes.byref(ms), ctypes.byref(n),
                                                ctypes.byref(fl), ctypes.byref(by)), "lns_timing_info")
            mf = ctypes.c_double()
            try:
                self._check(self._L.lns_timing_mfma_flops(self._h, i, ctypes.byref(mf)), "lns_timing_mfma_flops")
            except _lib.LnsLibraryError:      # an earlier round's build under LNS_HIP_LIB (A/B runs): no executed-FLOP records
                pass
            if n.value:
                out[name.value.decode()] = dict(ms=ms.value, launches=int(n.value), flops=fl.value, bytes=by.value,
                                                mfma_flops=mf.value)
        return out


def _lib_key_cap():
    return 160


def param_shapes(args, **kw):
    """{state_dict key: shape} of the model `args` describes (no GPU needed)."""
    return Engine(make_config(args, **kw)).param_shapes()
