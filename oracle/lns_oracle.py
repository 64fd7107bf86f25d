"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the LNS rollout hot path.

A restatement (numpy reshapes + the plain-C ops of oracle/lns_oracle.c) of the
reference's `LatentDynamics.predict` path:
    encode once -> T x (propagate ; decode)                train_stage2_ns2d.py:143-158
It consumes a reference-format state_dict ({key: float32 ndarray}) and the same
`args` namespace the reference reads.  Nothing here is imported by the product
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
use it, and only as the checker / reported CPU baseline.

PARITY PIN: this oracle is validated against outputs of the REAL reference
(imported on CPU in the build container through oracle/ref_shim.py); the
fixtures are committed under tests/golden/ (tools/make_golden.py).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblns_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    src = os.path.join(_HERE, "lns_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "liblns_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.lo_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().lo_num_threads())


def set_num_threads(n):
    lib().lo_set_num_threads(int(n))


# ----------------------------------------------------------------------------
# primitive ops (thin wrappers over lns_oracle.c)
# ----------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, dil=1, pad=(0, 0, 0, 0), mode=(0, 0)):
    """pad = (top, bottom, left, right); mode = (mode_y, mode_x), 0 zeros / 1 circular."""
    x = _c(x)
    w = _c(w)
    B, Cin, H, W = x.shape
    Cout, Cin2, KH, KW = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    Hp, Wp = H + pad[0] + pad[1], W + pad[2] + pad[3]
    Ho = (Hp - dil * (KH - 1) - 1) // stride + 1
    Wo = (Wp - dil * (KW - 1) - 1) // stride + 1
    y = np.empty((B, Cout, Ho, Wo), np.float32)
    bb = _c(b) if b is not None else None
    lib().lo_conv2d(_p(x), B, Cin, H, W, _p(w), _p(bb), Cout, KH, KW, stride, dil,
                    pad[0], pad[1], pad[2], pad[3], mode[0], mode[1], _p(y))
    return y


def groupnorm(x, groups, eps, gamma=None, beta=None):
    x = _c(x)
    B, C = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    g = _c(gamma) if gamma is not None else None
    be = _c(beta) if beta is not None else None
    lib().lo_groupnorm(_p(x), B, C, HW, groups, ctypes.c_float(eps), _p(g), _p(be), _p(y))
    return y


def layernorm(x, gamma, beta, eps=1e-5):
    x = _c(x)
    C = x.shape[-1]
    rows = x.size // C
    y = np.empty_like(x)
    lib().lo_layernorm(_p(x), ctypes.c_long(rows), C, ctypes.c_float(eps), _p(_c(gamma)),
                       _p(_c(beta)), _p(y))
    return y


def swish(x):
    x = _c(x)
    y = np.empty_like(x)
    lib().lo_swish(_p(x), ctypes.c_long(x.size), _p(y))
    return y


def gelu(x):
    x = _c(x)
    y = np.empty_like(x)
    lib().lo_gelu(_p(x), ctypes.c_long(x.size), _p(y))
    return y


def linear(x, w, b=None):
    x = _c(x)
    w = _c(w)
    Out, In = w.shape
    assert x.shape[-1] == In
    rows = x.size // In
    y = np.empty(x.shape[:-1] + (Out,), np.float32)
    bb = _c(b) if b is not None else None
    lib().lo_linear(_p(x), ctypes.c_long(rows), In, _p(w), _p(bb), Out, _p(y))
    return y


def bmm(a, b, transB=False):
    """a: [..., M, K]; b: [..., K, N] (or [..., N, K] if transB)."""
    a = _c(a)
    b = _c(b)
    M, K = a.shape[-2:]
    N = b.shape[-2] if transB else b.shape[-1]
    batch = int(np.prod(a.shape[:-2]))
    c = np.empty(a.shape[:-2] + (M, N), np.float32)
    lib().lo_bmm(_p(a), _p(b), _p(c), ctypes.c_long(batch), M, N, K, int(transB),
                 ctypes.c_long(M * K), ctypes.c_long(K * N), ctypes.c_long(M * N))
    return c


def softmax_rows(x, scale):
    x = _c(x).copy()
    n = x.shape[-1]
    lib().lo_softmax_rows(_p(x), ctypes.c_long(x.size // n), n, ctypes.c_float(scale))
    return x


def upsample_nearest(x, Ho, Wo, scale_h=0.0, scale_w=0.0):
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty((B, C, Ho, Wo), np.float32)
    lib().lo_upsample_nearest(_p(x), B, C, H, W, Ho, Wo, ctypes.c_float(scale_h),
                              ctypes.c_float(scale_w), _p(y))
    return y


def fa_contract(u, kx, ky, heads):
    u = _c(u)
    kx = _c(kx)
    ky = _c(ky)
    B, HC, H, W = u.shape
    o = np.empty_like(u)
    lib().lo_fa_contract(_p(u), _p(kx), _p(ky), B, heads, HC // heads, H, W, _p(o))
    return o


# ----------------------------------------------------------------------------
# module restatements
# ----------------------------------------------------------------------------
ZEROS, CIRC = 0, 1


class _Net:
    """Holds the state_dict and the padding convention."""

    def __init__(self, sd, mode):
        self.sd = sd
        self.mode = mode  # (mode_y, mode_x) of the 'same' convs
        self.trace = None  # optional {module prefix: output} record for layer-by-layer tests

    def rec(self, name, val):
        if self.trace is not None:
            self.trace[name] = val
        return val

    def has(self, k):
        return k in self.sd

    def __getitem__(self, k):
        return self.sd[k]

    # nn.Conv2d(cin,cout,k,1,p,padding_mode) / HalfPeriodicConv2d
    def conv(self, x, pfx, stride=1, dil=1, pad=None, mode=None):
        w = self.sd[pfx + ".weight"]
        b = self.sd.get(pfx + ".bias")
        k = w.shape[-1]
        if pad is None:
            p = dil * (k - 1) // 2
            pad = (p, p, p, p)
        return self.rec(pfx, conv2d(x, w, b, stride, dil, pad, self.mode if mode is None else mode))

    # basics.GroupNorm wrapper: 32 groups, eps 1e-6  (modules/basics.py:18-24)
    def gn32(self, x, pfx):
        return groupnorm(x, 32, 1e-6, self.sd[pfx + ".gn.weight"], self.sd[pfx + ".gn.bias"])

    # raw nn.GroupNorm(groups, C) eps 1e-5
    def gn(self, x, pfx, groups):
        return groupnorm(x, groups, 1e-5, self.sd[pfx + ".weight"], self.sd[pfx + ".bias"])


def residual_block(net, x, pfx):
    """modules/basics.py:245-256,272-276"""
    h = net.conv(swish(net.gn32(x, pfx + ".block.0")), pfx + ".block.2")
    h = net.conv(swish(net.gn32(h, pfx + ".block.3")), pfx + ".block.5")
    if net.has(pfx + ".channel_up.weight"):
        x = net.conv(x, pfx + ".channel_up")
    return net.rec(pfx, x + h)


def hp_residual_block(net, x, pfx):
    """modules/autoencoder2d_half_periodic.py:77-103"""
    skip = net.conv(x, pfx + ".channel_up") if net.has(pfx + ".channel_up.weight") else x
    h = net.conv(swish(net.gn32(x, pfx + ".norm_act1.norm_act.0")), pfx + ".conv1")
    h = net.conv(swish(net.gn32(h, pfx + ".norm_act2.norm_act.0")), pfx + ".conv2")
    return net.rec(pfx, h + skip)


def cond_residual_block(net, x, pfx, cond_emb):
    """CondResidualBlock.forward, norm=True (GroupNorm(1, C), eps 1e-5), GELU, use_scale_shift_norm=False:
    modules/cond_utils.py:112-128"""
    sd = net.sd
    h = net.conv(gelu(net.gn(x, pfx + ".norm1", 1)), pfx + ".conv1")
    e = linear(cond_emb, sd[pfx + ".cond_emb.weight"], sd[pfx + ".cond_emb.bias"])[:, :, None, None]
    h = h + e
    h = net.conv(gelu(net.gn(h, pfx + ".norm2", 1)), pfx + ".conv2")
    sc = conv2d(x, sd[pfx + ".shortcut.weight"], sd[pfx + ".shortcut.bias"]) if net.has(pfx + ".shortcut.weight") else x
    return net.rec(pfx, h + sc)


def downsample_block(net, x, pfx):
    """modules/basics.py:302-328: F.pad then Conv2d(ch,ch,3,2,0)"""
    if net.mode == (CIRC, CIRC):
        pad = (1, 1, 1, 1)
    else:
        pad = (0, 1, 0, 1)
    return net.conv(x, pfx + ".conv_layer", stride=2, pad=pad)


def hp_downsample_block(net, x, pfx):
    """modules/autoencoder2d_half_periodic.py:68-74: HalfPeriodicConv2d(ch,ch,3,2,1)"""
    return net.conv(x, pfx + ".conv_layer", stride=2, pad=(1, 1, 1, 1))


def upsample_block(net, x, pfx):
    """modules/basics.py:295-299 / autoencoder2d_half_periodic.py:61-65"""
    B, C, H, W = x.shape
    x = upsample_nearest(x, 2 * H, 2 * W, 0.5, 0.5)
    return net.conv(x, pfx + ".conv_layer")


def sa_block(net, x, pfx, heads):
    """modules/basics.py:377-404"""
    sd = net.sd
    B, C, H, W = x.shape
    n = H * W
    t = np.ascontiguousarray(x.reshape(B, C, n).transpose(0, 2, 1))  # b n c
    x_in = t
    h = layernorm(t, sd[pfx + ".ln.weight"], sd[pfx + ".ln.bias"])
    if (pfx + ".pe") in sd:
        h = h + sd[pfx + ".pe"][:, :n]
    q = linear(h, sd[pfx + ".to_q.weight"])
    k = linear(h, sd[pfx + ".to_k.weight"])
    v = linear(h, sd[pfx + ".to_v.weight"], sd[pfx + ".to_v.bias"])
    d = q.shape[-1] // heads

    def split(a):
        return np.ascontiguousarray(a.reshape(B, n, heads, d).transpose(0, 2, 1, 3))  # b h n d
    q, k, v = split(q), split(k), split(v)
    attn = bmm(q, k, transB=True)                      # b h n n
    attn = softmax_rows(attn, float(int(d) ** (-0.5)))
    out = bmm(attn, v)                                 # b h n d
    out = np.ascontiguousarray(out.transpose(0, 2, 1, 3)).reshape(B, n, heads * d)
    out = linear(out, sd[pfx + ".proj_out.weight"], sd[pfx + ".proj_out.bias"])
    out = x_in + out
    return net.rec(pfx, np.ascontiguousarray(out.transpose(0, 2, 1)).reshape(B, C, H, W))


def _rotary(t, inv_freq):
    """modules/embedding.py:171-186 with pos = linspace(0,1,n) (factorized_attention.py:48);
    t: [b,h,n,d]."""
    n = t.shape[2]
    pos = np.linspace(0.0, 1.0, n, dtype=np.float32) if n > 1 else np.zeros(1, np.float32)
    tt = pos * np.float32(1.0 / (1.0 / 64.0))          # scale/min_freq, scale=1, min_freq=1/64
    freqs = tt[:, None] * inv_freq[None, :].astype(np.float32)
    freqs = np.concatenate([freqs, freqs], axis=-1).astype(np.float32)  # [n, d]
    d = t.shape[-1]
    x1, x2 = t[..., : d // 2], t[..., d // 2:]
    rot = np.concatenate([-x2, x1], axis=-1)
    return (t * np.cos(freqs) + rot * np.sin(freqs)).astype(np.float32)


def _low_rank_kernel(sd, pfx, u, heads):
    """modules/factorized_attention.py:43-69; u: [b,n,c] -> K [b,h,n,n] (no softmax, scaling 1)"""
    B, n, _ = u.shape
    qk = linear(u, sd[pfx + ".to_qk.weight"])
    half = qk.shape[-1] // 2
    q, k = qk[..., :half], qk[..., half:]
    d = half // heads

    def split(a):
        return np.ascontiguousarray(a.reshape(B, n, heads, d).transpose(0, 2, 1, 3))
    q, k = split(q), split(k)
    inv_freq = sd[pfx + ".pos_emb.inv_freq"]
    q = _rotary(q, inv_freq)
    k = _rotary(k, inv_freq)
    return bmm(q, k, transB=True)


def _pooling_reducer(sd, pfx, x):
    """modules/factorized_attention.py:86-94; x: [b,c,nx,ny] -> [b,nx,out]"""
    t = linear(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), sd[pfx + ".to_in.weight"])  # b nx ny c
    t = t.mean(axis=2, dtype=np.float32)
    t = layernorm(t, sd[pfx + ".out_ffn.0.weight"], sd[pfx + ".out_ffn.0.bias"])
    t = gelu(linear(t, sd[pfx + ".out_ffn.1.weight"]))
    return linear(t, sd[pfx + ".out_ffn.3.weight"], sd[pfx + ".out_ffn.3.bias"])


def fa_block(net, x, pfx, heads):
    """modules/factorized_attention.py:144-159"""
    sd = net.sd
    u_skip = x
    u = net.gn(x, pfx + ".in_norm", 1)
    u_phi = conv2d(u, sd[pfx + ".in_proj.weight"])
    u = conv2d(u, sd[pfx + ".to_in.0.weight"])
    u_x = _pooling_reducer(sd, pfx + ".to_x.0", u)
    u_y = _pooling_reducer(sd, pfx + ".to_y.1", np.ascontiguousarray(u.transpose(0, 1, 3, 2)))
    k_x = _low_rank_kernel(sd, pfx + ".low_rank_kernel_x", u_x, heads)
    k_y = _low_rank_kernel(sd, pfx + ".low_rank_kernel_y", u_y, heads)
    u_phi = fa_contract(u_phi, k_x, k_y, heads)
    C = u_phi.shape[1]
    h = groupnorm(u_phi, C, 1e-5)                       # InstanceNorm2d, no affine
    h = gelu(conv2d(h, sd[pfx + ".to_out.1.weight"]))
    h = conv2d(h, sd[pfx + ".to_out.3.weight"])
    return net.rec(pfx, h + u_skip)


def spectral_conv2d(sd, pfx, x, emb12=None):
    """modules/basics.py:126-149 ; conditional variant modules/fourier_cond.py:55-81"""
    w1 = sd[pfx + ".weights1"]
    w2 = sd[pfx + ".weights2"]
    w1c = w1[..., 0] + 1j * w1[..., 1]
    w2c = w2[..., 0] + 1j * w2[..., 1]
    m1, m2 = w1.shape[2], w1.shape[3]
    B, Cin, H, W = x.shape
    Cout = w1.shape[1]
    x_ft = np.fft.rfft2(x.astype(np.float64))
    out_ft = np.zeros((B, Cout, H, W // 2 + 1), np.complex128)
    lo = x_ft[:, :, :m1, :m2]
    hi = x_ft[:, :, -m1:, :m2]
    if emb12 is not None:
        lo = lo * emb12[..., 0][:, None]
        hi = hi * emb12[..., 1][:, None]
    out_ft[:, :, :m1, :m2] = np.einsum("bixy,ioxy->boxy", lo, w1c)
    out_ft[:, :, -m1:, :m2] = np.einsum("bixy,ioxy->boxy", hi, w2c)
    return np.fft.irfft2(out_ft, s=(H, W)).astype(np.float32)


def activation(x, name):
    """ACTIVATION_REGISTRY: modules/basics.py:10-16"""
    if name == "gelu":
        return gelu(x)
    if name == "silu":
        return swish(x)
    if name == "relu":
        return np.maximum(x, np.float32(0.0))
    if name == "tanh":
        return np.tanh(x).astype(np.float32)
    if name == "sigmoid":
        return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x))).astype(np.float32)
    raise NotImplementedError(name)


def fourier_basic_block(net, x, pfx, act="gelu", residual=True):
    """modules/basics.py:574-583 (in_planes != planes allowed when residual is False)"""
    x1 = spectral_conv2d(net.sd, pfx + ".fourier", x)
    x2 = conv2d(x, net.sd[pfx + ".conv.weight"], net.sd[pfx + ".conv.bias"])
    out = activation(x1 + x2, act)
    return x + out if residual else out


def cond_fourier_basic_block(sd, pfx, x, cond_emb, residual=True):
    """modules/fourier_cond.py:106-117 (+ FreqLinear :25-29)"""
    fw = sd[pfx + ".fourier.cond_emb.weights"]
    fb = sd[pfx + ".fourier.cond_emb.bias"]
    m1 = sd[pfx + ".fourier.weights1"].shape[2]
    m2 = sd[pfx + ".fourier.weights1"].shape[3]
    h = (cond_emb.astype(np.float32) @ fw + fb).reshape(cond_emb.shape[0], m1, m2, 2, 2)
    emb12 = h[..., 0] + 1j * h[..., 1]                  # view_as_complex over the last dim
    x1 = spectral_conv2d(sd, pfx + ".fourier", x, emb12)
    x2 = conv2d(x, sd[pfx + ".conv.weight"], sd[pfx + ".conv.bias"])
    e = linear(cond_emb, sd[pfx + ".cond_emb.weight"], sd[pfx + ".cond_emb.bias"])
    out = gelu(x1 + x2 + e[:, :, None, None])
    return x + out if residual else out


def fourier_embedding(t, dim, max_period=10000):
    """modules/cond_utils.py:19-38"""
    half = dim // 2
    freqs = np.exp(-math.log(max_period) * np.arange(half, dtype=np.float32) / half).astype(np.float32)
    a = t.astype(np.float32)[:, None] * freqs[None]
    emb = np.concatenate([np.cos(a), np.sin(a)], axis=-1).astype(np.float32)
    if dim % 2:
        emb = np.concatenate([emb, np.zeros_like(emb[:, :1])], axis=-1)
    return emb


# ----------------------------------------------------------------------------
# autoencoders
# ----------------------------------------------------------------------------
def _attn_layer(net, x, pfx, args, heads, res):
    if args.use_fa:
        return fa_block(net, x, pfx, heads)
    return sa_block(net, x, pfx, heads)


class OracleAutoencoder:
    """SimpleAutoencoder of the three AE files; `pfx` is '' or 'vq_ae.' / 'ae.'."""

    def __init__(self, args, sd, pfx=""):
        self.args = args
        self.pfx = pfx
        fam = args.family
        if fam == "sw_half_periodic":
            self.kind = "hp"
            mode = (ZEROS, CIRC) if args.periodic_direction == "x" else (CIRC, ZEROS)
        else:
            self.kind = "square" if fam == "ns2d" else "nonsq"
            mode = (CIRC, CIRC) if args.is_periodic else (ZEROS, ZEROS)
        self.net = _Net(sd, mode)

    # -- encode ---------------------------------------------------------------
    def encode(self, x):
        a, net, p = self.args, self.net, self.pfx + "encoder.model."
        ch = a.encoder_channels
        idx = 0
        x = conv2d(x, net[p + "0.weight"], net[p + "0.bias"])          # 1x1
        x = net.rec(p + "0", swish(x))      # engine fuses the Swish into this conv's epilogue
        idx = 2
        if self.kind == "hp":
            # autoencoder2d_half_periodic.py:120-139
            x = hp_residual_block(net, x, p + "2")
            idx = 3
            res_h = a.resolutions[0]
            for i in range(len(ch) - 1):
                for _ in range(a.encoder_res_blocks):
                    x = hp_residual_block(net, x, p + str(idx)); idx += 1
                if i != len(ch) - 2:
                    x = hp_downsample_block(net, x, p + str(idx)); idx += 1
            x = hp_residual_block(net, x, p + str(idx)); idx += 1
        else:
            # autoencoder2d.py:30-67 / autoencoder2d_nonsquared.py:36-63
            x = net.conv(x, p + "2")
            idx = 3
            res = a.resolution if self.kind == "square" else a.resolutions[0]
            for i in range(len(ch) - 1):
                for _ in range(a.encoder_res_blocks):
                    x = residual_block(net, x, p + str(idx)); idx += 1
                    if self.kind == "nonsq" and res in a.fourier_resolutions:
                        x = fourier_basic_block(net, x, p + str(idx)); idx += 1
                if self.kind == "square":
                    if res in a.attn_resolutions and a.use_attn_enc:
                        x = _attn_layer(net, x, p + str(idx), a, a.attn_heads, res); idx += 1
                    if res in a.fourier_resolutions:
                        x = fourier_basic_block(net, x, p + str(idx)); idx += 1
                if i != len(ch) - 2:
                    x = downsample_block(net, x, p + str(idx)); idx += 1
                    res //= 2
            if self.kind == "square":
                x = net.conv(x, p + str(idx)); idx += 1
            else:
                x = residual_block(net, x, p + str(idx)); idx += 1
        x = swish(net.gn32(x, p + str(idx))); idx += 2
        x = net.rec(p + str(idx), conv2d(x, net[p + f"{idx}.weight"], net[p + f"{idx}.bias"]))
        q = self.pfx + "quant_conv"
        return conv2d(x, net[q + ".weight"], net[q + ".bias"])

    # -- decode ---------------------------------------------------------------
    def decode(self, z):
        a, net, p = self.args, self.net, self.pfx + "decoder.model."
        q = self.pfx + "post_quant_conv"
        x = net.rec(q, conv2d(z, net[q + ".weight"], net[q + ".bias"]))
        ch = a.decoder_channels
        kind = self.kind
        heads = a.attn_heads if kind == "square" else a.decoder_attn_heads
        disable_coarse = bool(getattr(a, "disable_coarse_attn", False))
        rb = hp_residual_block if kind == "hp" else residual_block
        up = upsample_block
        x = net.conv(x, p + "0")   # 1x1 (square) or 3x3 (nonsq / hp) -- from the weight shape
        idx = 1
        if kind == "hp":
            # autoencoder2d_half_periodic.py:167-175
            if not disable_coarse:
                x = sa_block(net, x, p + "1", heads); idx = 2
                x = rb(net, x, p + "2"); idx = 3
            else:
                x = rb(net, x, p + "1"); x = rb(net, x, p + "2"); idx = 3
        else:
            x = rb(net, x, p + "1"); idx = 2
            if not disable_coarse:
                x = sa_block(net, x, p + "2", heads); idx = 3
            x = rb(net, x, p + str(idx)); idx += 1
        res = a.latent_resolution
        for i in range(len(ch)):
            for _ in range(a.decoder_res_blocks):
                x = rb(net, x, p + str(idx)); idx += 1
                if kind != "square" and res in a.attn_resolutions:
                    x = _attn_layer(net, x, p + str(idx), a, heads, res); idx += 1
            if kind == "square" and res in a.attn_resolutions:
                x = _attn_layer(net, x, p + str(idx), a, heads, res); idx += 1
            if i != 0 and i != len(ch) - 1:
                x = up(net, x, p + str(idx)); idx += 1
                res *= 2
        x = upsample_nearest(x, a.Ly, a.Lx); idx += 1
        res = a.Ly
        x = net.conv(x, p + str(idx)); idx += 1
        if a.final_smoothing:
            x = fourier_basic_block(net, x, p + str(idx)); idx += 1
        else:
            if res in a.attn_resolutions:
                x = _attn_layer(net, x, p + str(idx), a, heads, res); idx += 1
            x = net.conv(x, p + str(idx)); idx += 1   # 1x1 (square) or 3x3
        if kind == "square":
            x = net.gn(x, p + str(idx), 8)            # nn.GroupNorm(8, C): autoencoder2d.py:149
        else:
            x = net.gn32(x, p + str(idx))
        idx += 1
        x = swish(x); idx += 1
        return conv2d(x, net[p + f"{idx}.weight"], net[p + f"{idx}.bias"])


# ----------------------------------------------------------------------------
# propagators
# ----------------------------------------------------------------------------
class OracleCondAutoencoder(OracleAutoencoder):
    """ConditionalSimpleAutoencoder: modules/autoencoder2d_nonsquared.py:279-305 (CondEncoder.forward :126-145)."""

    def encode(self, x, param):
        a, net, sd, p = self.args, self.net, self.net.sd, self.pfx + "encoder."
        E = a.cond_emb_channels
        ce = fourier_embedding(np.asarray(param), E)
        ce = linear(ce, sd[p + "embed.0.weight"], sd[p + "embed.0.bias"])
        ce = linear(swish(ce), sd[p + "embed.2.weight"], sd[p + "embed.2.bias"])
        x = net.rec(p + "to_in.0", swish(conv2d(x, sd[p + "to_in.0.weight"], sd[p + "to_in.0.bias"])))
        x = net.conv(x, p + "to_in.2")
        n = len(a.encoder_channels)
        for i in range(n - 1):
            for j in range(a.encoder_res_blocks):
                x = cond_residual_block(net, x, p + f"layers.{i}.0.{j}", ce)
            if i != n - 2:
                x = downsample_block(net, x, p + f"layers.{i}.1")
        x = cond_residual_block(net, x, p + "to_out_conv", ce)
        x = swish(net.gn32(x, p + "to_out.0"))
        x = net.rec(p + "to_out.2", conv2d(x, sd[p + "to_out.2.weight"], sd[p + "to_out.2.bias"]))
        q = self.pfx + "quant_conv"
        return conv2d(x, sd[q + ".weight"], sd[q + ".bias"])

    def forward(self, x, param):
        return self.decode(self.encode(x, param))


class OraclePropagator:
    """SimpleCNN of train_stage2_{ns2d,SW,twophase,twophase_conditional}.py"""

    def __init__(self, args, sd, pfx="propagator."):
        self.args = args
        self.pfx = pfx
        fam = args.family
        if fam == "ns2d":
            mode = (CIRC, CIRC)                       # train_stage2_ns2d.py:75
        elif fam in ("sw_half_periodic", "sw_nonsquared"):
            mode = (ZEROS, CIRC)                      # train_stage2_SW.py:76 periodic_direction='x'
        else:
            mode = (ZEROS, ZEROS)                     # train_stage2_twophase.py:76 ; _conditional.py:106
        self.net = _Net(sd, mode)
        self.cond = fam == "twophase_cond"

    def _block(self, x, p, dil):
        """train_stage2_ns2d.py:50-53"""
        net = self.net
        h = net.gn(x, p + ".conv.0", 1)
        h = gelu(net.conv(h, p + ".conv.1"))
        h = gelu(net.conv(h, p + ".conv.3", dil=dil))
        h = net.conv(h, p + ".conv.5")
        x = x + h
        h = net.gn(x, p + ".ffn.0", 1)
        h = gelu(conv2d(h, net[p + ".ffn.1.weight"]))
        h = conv2d(h, net[p + ".ffn.3.weight"])
        return net.rec(p, x + h)

    def _cond_block(self, x, p, dil, cond_emb):
        """train_stage2_twophase_conditional.py:66-75"""
        net, sd = self.net, self.net.sd
        e = linear(cond_emb, sd[p + ".cond_emb.weight"], sd[p + ".cond_emb.bias"])[:, :, None, None]
        x_skip = x
        h = net.gn(x, p + ".conv1.0", 1)
        h = gelu(net.conv(h, p + ".conv1.1"))
        h = net.conv(h, p + ".conv1.3", dil=dil)
        h = h + e
        h = gelu(net.gn(h, p + ".cond_conv1.0", 1))
        h = net.conv(h, p + ".cond_conv1.2")
        x = x_skip + h
        m = net.gn(np.ascontiguousarray(e), p + ".cond_conv2.0", 1)
        m = gelu(net.conv(m, p + ".cond_conv2.1"))
        m = net.conv(m, p + ".cond_conv2.3")
        u = x * (np.float32(1.0) + m)
        h = net.gn(u, p + ".ffn.0", 1)
        h = gelu(conv2d(h, net[p + ".ffn.1.weight"]))
        h = conv2d(h, net[p + ".ffn.3.weight"])
        return net.rec(p, x + h)

    def forward(self, z, param=None):
        a, net, p = self.args, self.net, self.pfx
        x = net.rec(p + "in_proj", conv2d(z, net[p + "in_proj.weight"], net[p + "in_proj.bias"]))
        if self.cond:
            sd = net.sd
            ce = fourier_embedding(np.asarray(param), a.latent_dim)
            ce = linear(ce, sd[p + "cond_emb_proj.0.weight"], sd[p + "cond_emb_proj.0.bias"])
            ce = linear(gelu(ce), sd[p + "cond_emb_proj.2.weight"], sd[p + "cond_emb_proj.2.bias"])
            for i in range(a.prop_n_block):
                x = self._cond_block(x, p + f"net.{i}", a.dilation, ce)
        else:
            for i in range(a.prop_n_block):
                x = self._block(x, p + f"net.{i}", a.dilation)
        x = net.gn32(x, p + "out_proj.0")
        return conv2d(x, net[p + "out_proj.1.weight"], net[p + "out_proj.1.bias"])


class OracleDynamics:
    """LatentDynamics.predict: train_stage2_ns2d.py:143-158 (cond: _conditional.py:177-193)."""

    def __init__(self, args, sd):
        self.args = args
        ae_pfx = "ae." if args.family == "twophase_cond" else "vq_ae."
        self.ae = OracleAutoencoder(args, sd, ae_pfx)
        self.prop = OraclePropagator(args, sd)

    def x_to_z(self, x):
        return self.ae.encode(x)

    def z_to_x(self, z):
        return self.ae.decode(z)

    def predict(self, x, steps, param=None, to_x=False, return_latents=False):
        z = self.x_to_z(x)
        out, lat = [], []
        for _ in range(steps):
            z = self.prop.forward(z, param)
            lat.append(z)
            out.append(self.z_to_x(z) if to_x else z)
        out = np.stack(out, axis=1)
        if return_latents:
            return out, np.stack(lat, axis=1)
        return out


# ---------------------------------------------------------------------------------------------------
# "next rows" (SURVEY 8f): denormalise + relative_lp_loss, bulk encode
# ---------------------------------------------------------------------------------------------------
def relative_lp_loss(pred, gt, reduce_dim, eps=1e-8):
    """training_utils.py:9-23 with p=2, reduction='sum', reduce_all=False (float64 accumulation)."""
    pred = np.asarray(pred, np.float64)
    gt = np.asarray(gt, np.float64)
    gt_norm = (gt ** 2).sum(axis=reduce_dim)
    gt_norm = np.where(gt_norm < eps, eps, gt_norm)
    diff = ((pred - gt) ** 2).sum(axis=reduce_dim)
    return np.sqrt(diff / gt_norm)


def denormalize(x, mean, std):
    """dataset/ns2d_fno_stage2_simpleae.py:140-149 (affine, scalar stats)."""
    return np.asarray(x, np.float64) * std + mean


def denormalize_channels(x, mean, std):
    """dataset/Stage2_SW.py:60-72: per-channel affine map on [..., C, H, W]."""
    x = np.asarray(x, np.float64)
    m = np.asarray(mean, np.float64).reshape(-1, 1, 1)
    s = np.asarray(std, np.float64).reshape(-1, 1, 1)
    return x * s + m


def denormalize_twophase(x, vel_mean, vel_std, prs_mean, prs_std):
    """dataset/twophase_flow_stage2.py:370-390: velocities denormalised then zeroed on the four closed walls,
    pressure denormalised, VOF clamped to [0, 1+1e-8].  x [..., 4, H, W]."""
    x = np.array(x, np.float64, copy=True)
    x[..., :2, :, :] = x[..., :2, :, :] * vel_std + vel_mean
    x[..., :2, 0, :] = 0.0
    x[..., :2, -1, :] = 0.0
    x[..., :2, :, 0] = 0.0
    x[..., :2, :, -1] = 0.0
    x[..., 2, :, :] = x[..., 2, :, :] * prs_std + prs_mean
    x[..., 3, :, :] = np.clip(x[..., 3, :, :], 0.0, 1.0 + 1e-8)
    return x


def rollout_metrics(y_hat, y, mean=0.0, std=1.0, eps=1e-8, denorm=None):
    """The validate_loop metric pair of train_stage2_ns2d.py:253-257 on [B,T,C,H,W] arrays; `denorm` replaces the
    scalar affine map by another dataset's denormalize (a callable applied to both arrays)."""
    if denorm is None:
        yh, yt = denormalize(y_hat, mean, std), denormalize(y, mean, std)
    else:
        yh, yt = denorm(y_hat), denorm(y)
    return relative_lp_loss(yh, yt, (3, 4), eps), relative_lp_loss(yh, yt, (1, 3, 4), eps)


def smooth_l1_loss(pred, target, beta=1.0):
    """torch.nn.functional.smooth_l1_loss (mean reduction), the loss_fn of train_stage2_ns2d.py:209,227."""
    d = np.abs(np.asarray(pred, np.float64) - np.asarray(target, np.float64))
    return float(np.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta).mean())


def teacher_forced_loss(dyn, z_in, z_out, param=None, loss_fn=smooth_l1_loss):
    """LatentDynamics.forward (train_stage2_ns2d.py:126-141): roll the propagator t_out steps from z_in[:, 0] and
    compare the stacked latents with z_out.  dyn: OracleDynamics."""
    z = np.asarray(z_in)[:, 0]
    preds = []
    for _ in range(z_out.shape[1]):
        z = dyn.prop.forward(z, param)
        preds.append(z)
    return loss_fn(np.stack(preds, 1), z_out)


def normalize_frames(frames, mean=0.0, std=1.0, eps=1e-8):
    """Dataset normalisation ahead of encode_dataset: (u - mean) / (std + eps) with scalar statistics
    (ns2d_fno_stage2_simpleae.py:78-79, eps 1e-8) or per-channel ones over [N,C,H,W] -- Stage2_SW.normalize
    (dataset/Stage2_SW.py:74-78: u, v, pres, eps 0) and TwoPhaseFlow.normalize_data
    (dataset/twophase_flow_stage2.py:304-313: both velocity channels by the velocity statistics, pressure by its own,
    the VOF channel untouched, eps 0)."""
    u = np.asarray(frames, np.float32)
    m = np.asarray(mean, np.float32)
    sd = np.asarray(std, np.float32)
    if m.ndim == 1:
        m = m.reshape(1, -1, 1, 1)
    if sd.ndim == 1:
        sd = sd.reshape(1, -1, 1, 1)
    return ((u - m) / (sd + np.float32(eps))).astype(np.float32)


def encode_dataset(ae, frames, chunk=32, mean=0.0, std=1.0, eps=1e-8, param=None):
    """ns2d_fno_stage2_simpleae.py:81-93 / Stage2_SW.py:80-105 / twophase_flow_stage2.py:317-337: normalise, encode
    `chunk` frames at a time (the two-phase dataset uses chunks of 32, the others one case per call)."""
    outs = []
    for s in range(0, frames.shape[0], chunk):
        u = normalize_frames(frames[s:s + chunk], mean, std, eps)
        outs.append(ae.encode(u) if param is None else ae.encode(u, np.asarray(param)[s:s + chunk]))
    return np.concatenate(outs, 0)
