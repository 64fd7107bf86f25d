"""ctypes binding of liblns_hip.so (the C ABI of include/lns.h).

There is NO fallback: if the HIP library is missing the import of any compute
path fails loudly (LnsLibraryError).  `build()` compiles it in-tree with hipcc
for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# LNS_HIP_LIB: A/B runs of two builds of the library inside one GPU call (tools/ab_bench.sh); never set in production
LIB_PATH = os.environ.get("LNS_HIP_LIB") or os.path.join(_HERE, "liblns_hip.so")
CSRC = os.path.join(_HERE, "csrc")

LNS_MAX_STAGES = 8
LNS_ABI_VERSION = 2
LNS_AE_NONE, LNS_AE_SQUARE, LNS_AE_NONSQUARED, LNS_AE_HALF_PERIODIC = 0, 1, 2, 3
LNS_PROP_NONE, LNS_PROP_PLAIN, LNS_PROP_CONDITIONAL = 0, 1, 2
LNS_PAD_ZEROS, LNS_PAD_CIRCULAR = 0, 1
# status codes (include/lns.h)
LNS_OK, LNS_EINVAL, LNS_ENOKEY, LNS_ESTATE, LNS_ENOMEM, LNS_EHIP, LNS_ENONFINITE = 0, -1, -2, -3, -4, -5, -6


class LnsLibraryError(RuntimeError):
    pass


class LnsError(RuntimeError):
    pass


_I32 = ctypes.c_int32
_I32x8 = ctypes.c_int32 * LNS_MAX_STAGES


class LnsConfig(ctypes.Structure):
    """Mirror of `struct lns_config` (include/lns.h)."""
    _fields_ = [
        ("abi_version", _I32), ("ae_kind", _I32), ("prop_kind", _I32),
        ("in_channels", _I32), ("latent_dim", _I32), ("Ly", _I32), ("Lx", _I32),
        ("res_h", _I32), ("res_w", _I32), ("latent_resolution", _I32),
        ("ae_pad_y", _I32), ("ae_pad_x", _I32),
        ("n_encoder_channels", _I32), ("encoder_channels", _I32x8),
        ("encoder_res_blocks", _I32), ("use_attn_enc", _I32),
        ("n_decoder_channels", _I32), ("decoder_channels", _I32x8),
        ("decoder_res_blocks", _I32),
        ("n_attn_resolutions", _I32), ("attn_resolutions", _I32x8),
        ("n_fourier_resolutions", _I32), ("fourier_resolutions", _I32x8),
        ("use_fa", _I32), ("final_smoothing", _I32), ("disable_coarse_attn", _I32),
        ("attn_heads", _I32), ("attn_dim", _I32),
        ("hw_ratio", ctypes.c_float),
        ("prop_n_block", _I32), ("prop_n_embd", _I32), ("prop_dilation", _I32),
        ("prop_pad_y", _I32), ("prop_pad_x", _I32), ("cond_emb_dim", _I32),
        ("ae_prefix", ctypes.c_char * 32), ("prop_prefix", ctypes.c_char * 32),
        ("cond_encoder", _I32), ("cond_emb_channels", _I32),
    ]


# every symbol include/lns.h declares (tests check the library exports them all)
SYMBOLS = [
    "lns_create_error", "lns_create", "lns_destroy", "lns_last_error", "lns_num_params",
    "lns_param_info", "lns_set_weight", "lns_finalize_weights", "lns_latent_shape", "lns_prepare",
    "lns_encode", "lns_encode_cond", "lns_encode_affine", "lns_decode", "lns_propagate", "lns_rollout", "lns_rollout_latent", "lns_check_finite", "lns_set_option",
    "lns_train_workspace_bytes", "lns_train_forward", "lns_train_backward",
    "lns_trace_enable", "lns_trace_count", "lns_trace_info", "lns_trace_copy",
    "lns_timing_enable", "lns_timing_count", "lns_timing_info", "lns_timing_mfma_flops", "lns_build_has",
    "lns_op_conv2d", "lns_op_conv_pair_stress", "lns_op_groupnorm_stats", "lns_op_attention", "lns_op_fa_sandwich", "lns_op_fourier_block",
    "lns_fourier_block_create", "lns_fourier_block_forward", "lns_fourier_block_destroy", "lns_metric_rel_l2", "lns_metric_rel_l2_ch",
]

_lib = None


def build(force=False):
    """Compile liblns_hip.so in-tree (hipcc --offload-arch=gfx950)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith((".hip", ".cpp", ".h"))] + [os.path.join(_HERE, "..", "include", "lns.h")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LnsLibraryError(
            "HIP extension %s is missing; run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the LNS hot path)" % LIB_PATH)
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise LnsLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    c = ctypes
    vp, i, i64p, fp = c.c_void_p, c.c_int, c.POINTER(c.c_int64), c.POINTER(c.c_float)
    if os.environ.get("LNS_HIP_LIB"):
        # same-box A/B against an EARLIER round's build (tools/ab_*.sh; never set in production): entry points that build
        # does not export yet become stubs that fail when called, so the core path (rollout, timing, op tests) still binds
        def _absent(*_a, **_k):
            raise LnsLibraryError("entry point missing from the library selected by LNS_HIP_LIB")
        for sym in SYMBOLS:
            if not hasattr(L, sym):
                setattr(L, sym, type("Stub", (), {"__call__": staticmethod(_absent), "argtypes": None, "restype": None})())
    L.lns_create_error.restype = c.c_char_p
    L.lns_create.argtypes = [c.POINTER(LnsConfig), c.POINTER(vp)]
    L.lns_destroy.argtypes = [vp]
    L.lns_destroy.restype = None
    L.lns_last_error.argtypes = [vp]
    L.lns_last_error.restype = c.c_char_p
    L.lns_num_params.argtypes = [vp]
    L.lns_param_info.argtypes = [vp, i, c.c_char_p, i, i64p, c.POINTER(i), c.POINTER(i)]
    L.lns_set_weight.argtypes = [vp, c.c_char_p, vp, i64p, i]
    L.lns_finalize_weights.argtypes = [vp, i]
    L.lns_latent_shape.argtypes = [vp, c.POINTER(i), c.POINTER(i), c.POINTER(i)]
    L.lns_prepare.argtypes = [vp, i, c.POINTER(c.c_size_t)]
    L.lns_encode.argtypes = [vp, vp, i, vp, vp, c.c_size_t, vp]
    L.lns_decode.argtypes = [vp, vp, i, vp, vp, c.c_size_t, vp]
    if hasattr(L, "lns_encode_cond"):
        L.lns_encode_cond.argtypes = [vp, vp, vp, i, vp, vp, c.c_size_t, vp]
    L.lns_encode_affine.argtypes = [vp, vp, vp, vp, i, vp, vp, c.c_size_t, vp]
    L.lns_propagate.argtypes = [vp, vp, vp, i, i, i, vp, vp, c.c_size_t, vp]
    L.lns_rollout.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp, c.c_size_t, vp]
    L.lns_rollout_latent.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp, c.c_size_t, vp]
    if hasattr(L, "lns_check_finite"):      # (absent from older builds loaded through LNS_HIP_LIB for A/B runs)
        L.lns_check_finite.argtypes = [vp, i, vp, c.c_size_t, vp]
    if hasattr(L, "lns_set_option"):
        L.lns_set_option.argtypes = [vp, c.c_char_p, c.c_long]
    if hasattr(L, "lns_train_forward"):
        L.lns_train_workspace_bytes.argtypes = [vp, i, i, i, i, c.POINTER(c.c_size_t)]
        L.lns_train_forward.argtypes = [vp, vp, vp, vp, i, i, i, i, vp, vp, c.c_size_t, vp]
        L.lns_train_backward.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp, vp, vp, c.c_size_t, vp]
    L.lns_build_has.argtypes = [c.c_char_p]
    L.lns_trace_enable.argtypes = [vp, i]
    L.lns_trace_count.argtypes = [vp]
    L.lns_trace_info.argtypes = [vp, i, c.c_char_p, i, i64p]
    L.lns_trace_copy.argtypes = [vp, i, vp]
    L.lns_timing_enable.argtypes = [vp, i]
    L.lns_timing_count.argtypes = [vp]
    L.lns_timing_info.argtypes = [vp, i, c.c_char_p, i, c.POINTER(c.c_double), i64p,
                                  c.POINTER(c.c_double), c.POINTER(c.c_double)]
    L.lns_timing_mfma_flops.argtypes = [vp, i, c.POINTER(c.c_double)]
    L.lns_op_conv2d.argtypes = [vp, i, i, i, i, i, i, vp, vp, i, i, i, i, i, i, i, i, i, i,
                                vp, i, i, vp, vp, vp, i, vp, vp]
    L.lns_op_groupnorm_stats.argtypes = [vp, i, i, i, i, c.c_float, vp, vp, vp, vp, vp]
    L.lns_op_attention.argtypes = [vp, i, i, i, i, c.c_float, vp, vp]
    L.lns_op_fa_sandwich.argtypes = [vp, vp, vp, i, i, i, i, i, c.c_float, i, vp, vp]
    L.lns_op_fourier_block.argtypes = [vp, i, i, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp, vp]
    L.lns_fourier_block_create.argtypes = [i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, c.POINTER(vp)]
    L.lns_fourier_block_forward.argtypes = [vp, vp, vp, i, i, i, vp, vp]
    L.lns_fourier_block_destroy.argtypes = [vp]
    L.lns_fourier_block_destroy.restype = None
    L.lns_metric_rel_l2.argtypes = [vp, vp, i, i, i, i, c.c_float, c.c_float, c.c_float, vp, vp, vp, vp]
    L.lns_metric_rel_l2.restype = i
    L.lns_metric_rel_l2_ch.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, c.c_float, c.c_float, c.c_float, vp, vp, vp, vp]
    L.lns_metric_rel_l2_ch.restype = i
    _lib = L
    return L
