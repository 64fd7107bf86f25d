"""Shared GPU parity checks (used by the -m gpu tests and tools/gpu_diag.py).

Every check runs the HIP path through the C ABI (ctypes) and compares with the
CPU oracle (oracle/, test infrastructure) or the committed golden fixtures.
"""
import ctypes

import numpy as np
import torch

import lns_oracle
from helpers import rel_l2
from lns_amd import _lib


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _hp(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(ctypes.c_void_p) if a is not None else None


def rng(seed):
    return np.random.default_rng(seed)


def to_oct8(a):
    """[B, C, H, W] -> the engine's channel-octet-interleaved layout [B, C/8, H*W, 8] (C % 8 == 0)."""
    B, C, H, W = a.shape
    return np.ascontiguousarray(a.reshape(B, C // 8, 8, H * W).transpose(0, 1, 3, 2))


def from_oct8(a, C, H, W):
    B = a.shape[0]
    return np.ascontiguousarray(a.reshape(B, C // 8, H * W, 8).transpose(0, 1, 3, 2)).reshape(B, C, H, W)


def conv_case(B, Cin, Cout, H, W, k=3, stride=1, dil=1, pad=None, mode=(1, 1), up=None, ss=False, act_in=0,
              act_out=0, res=False, badd=False, bias=True, variant=-1, seed=0, xscale=1.0, sample_scales=None,
              heavy_w=False, xdist="normal", fp64=False, ret_y=False, xoct=False, yoct=False):
    """Returns (rel_l2 error, output shape) of lns_op_conv2d vs the oracle composition.
    xoct / yoct: the input / the output and residual are handed to the kernel in the OCT8 layout (variant | 0x100 / 0x200).
    xscale / sample_scales: magnitude of the activations (per sample); heavy_w: heavy-tailed (Student-t, 2 dof)
    weights; xdist "lognormal": activations spread over several decades; fp64: additionally returns the error
    against an fp64 evaluation of the same composition (third element)."""
    L = _lib.lib()
    r = rng(seed)
    x = r.standard_normal((B, Cin, H, W)).astype(np.float32)
    if xdist == "lognormal":
        x = (x * np.exp(3.0 * r.standard_normal((B, Cin, H, W)))).astype(np.float32)
    x = (x * np.float32(xscale)).astype(np.float32)
    if sample_scales is not None:
        x = (x * np.asarray(sample_scales, np.float32).reshape(B, 1, 1, 1)).astype(np.float32)
    if heavy_w:
        w = (r.standard_t(2.0, size=(Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    else:
        w = (r.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bv = (r.standard_normal(Cout) * 0.1 * xscale).astype(np.float32) if bias else None
    if pad is None:
        p = dil * (k - 1) // 2
        pad = (p, p, p, p)
    ssv = None
    xin = x
    if ss:
        ssv = np.stack([1.0 + 0.2 * r.standard_normal((B, Cin)), 0.1 * xscale * r.standard_normal((B, Cin))], -1).astype(np.float32)
        xin = x * ssv[:, :, 0][:, :, None, None] + ssv[:, :, 1][:, :, None, None]
    if act_in == 1:
        xin = lns_oracle.swish(xin)
    elif act_in == 2:
        xin = lns_oracle.gelu(xin)
    Hv, Wv = H, W
    if up is not None:
        Hv, Wv = up
        sch = 0.5 if Hv == 2 * H else 0.0
        scw = 0.5 if Wv == 2 * W else 0.0
        xin = lns_oracle.upsample_nearest(xin, Hv, Wv, sch, scw)
    ref = lns_oracle.conv2d(xin, w, bv, stride, dil, pad, mode)
    badd_v = None
    if badd:
        badd_v = r.standard_normal((B, Cout)).astype(np.float32)
        ref = ref + badd_v[:, :, None, None]
    if act_out == 1:
        ref = lns_oracle.swish(ref)
    elif act_out == 2:
        ref = lns_oracle.gelu(ref)
    res_v = None
    if res:
        res_v = r.standard_normal(ref.shape).astype(np.float32)
        ref = ref + res_v
    xd = _dev(to_oct8(x) if xoct else x)
    y = torch.full(ref.shape, float("nan"), dtype=torch.float32, device="cuda")
    ssd = _dev(ssv) if ssv is not None else None
    resd = _dev(to_oct8(res_v) if yoct else res_v) if res_v is not None else None
    if xoct or yoct:
        assert variant >= 0
        variant = variant | (0x100 if xoct else 0) | (0x200 if yoct else 0)
    baddd = _dev(badd_v) if badd_v is not None else None
    amax = torch.zeros((B, 16), dtype=torch.int32, device="cuda")
    rc = L.lns_op_conv2d(xd.data_ptr(), B, Cin, H, W, Hv, Wv, _hp(w), _hp(bv), Cout, k, stride, dil,
                         pad[0], pad[1], pad[2], pad[3], mode[0], mode[1],
                         ssd.data_ptr() if ssd is not None else None, act_in, act_out,
                         resd.data_ptr() if resd is not None else None,
                         baddd.data_ptr() if baddd is not None else None, y.data_ptr(), variant, _stream(),
                         amax.data_ptr())
    assert rc == 0, "lns_op_conv2d rc=%d" % rc
    torch.cuda.synchronize()
    out = y.cpu().numpy()
    if yoct:
        out = from_oct8(out.reshape(B, Cout // 8, -1, 8), Cout, ref.shape[2], ref.shape[3])
    assert np.isfinite(out).all(), "non-finite / unwritten outputs"
    # amax side channel: bit pattern of max |y| per sample, exactly
    got = amax.cpu().numpy().view(np.float32).max(1)      # [B][16] sub-slots: the maximum is the sample's
    want = np.abs(out).reshape(B, -1).max(1)
    assert np.array_equal(got, want), ("amax side channel", got, want)
    # every sample on its own (samples of one batch may differ by orders of magnitude)
    err = max(rel_l2(out[i], ref[i]) for i in range(B))
    if ret_y:
        return err, ref.shape, out
    if fp64:
        r64 = conv_case_fp64(x, w, bv, stride, dil, pad, mode, up, ssv, act_in, badd_v, act_out, res_v)
        return err, ref.shape, max(rel_l2(out[i], r64[i]) for i in range(B))
    return err, ref.shape


def conv_case_fp64(x, w, bv, stride, dil, pad, mode, up, ssv, act_in, badd_v, act_out, res_v):
    """The same composition evaluated in fp64 (torch CPU): the tie-breaker for the split-operand kernels."""
    import torch.nn.functional as F
    xt = torch.from_numpy(x).double()
    if ssv is not None:
        st = torch.from_numpy(ssv).double()
        xt = xt * st[:, :, 0][:, :, None, None] + st[:, :, 1][:, :, None, None]
    if act_in == 1:
        xt = xt * torch.sigmoid(xt)
    elif act_in == 2:
        xt = F.gelu(xt)
    if up is not None:
        Hv, Wv = up
        H, W = xt.shape[-2:]
        iy = torch.clamp(torch.floor(torch.arange(Hv).float() * (0.5 if Hv == 2 * H else np.float32(H) / np.float32(Hv))).long(), max=H - 1)
        ix = torch.clamp(torch.floor(torch.arange(Wv).float() * (0.5 if Wv == 2 * W else np.float32(W) / np.float32(Wv))).long(), max=W - 1)
        xt = xt[:, :, iy][:, :, :, ix]
    xt = F.pad(xt, (pad[2], pad[3], 0, 0), mode="circular" if mode[1] else "constant")
    xt = F.pad(xt, (0, 0, pad[0], pad[1]), mode="circular" if mode[0] else "constant")
    y = F.conv2d(xt, torch.from_numpy(w).double(), torch.from_numpy(bv).double() if bv is not None else None,
                 stride=stride, dilation=dil)
    if badd_v is not None:
        y = y + torch.from_numpy(badd_v).double()[:, :, None, None]
    if act_out == 1:
        y = y * torch.sigmoid(y)
    elif act_out == 2:
        y = F.gelu(y)
    if res_v is not None:
        y = y + torch.from_numpy(res_v).double()
    return y.numpy()


CONV_CASES = [
    dict(B=2, Cin=64, Cout=64, H=32, W=32, k=3, mode=(1, 1)),
    dict(B=2, Cin=64, Cout=64, H=32, W=32, k=3, mode=(0, 0)),
    dict(B=2, Cin=64, Cout=64, H=24, W=48, k=3, mode=(0, 1)),
    dict(B=2, Cin=64, Cout=64, H=24, W=48, k=3, mode=(1, 0)),
    dict(B=2, Cin=128, Cout=128, H=16, W=16, k=3, dil=2, mode=(1, 1)),
    dict(B=2, Cin=128, Cout=128, H=12, W=24, k=3, dil=3, mode=(0, 1)),
    dict(B=2, Cin=128, Cout=128, H=7, W=15, k=3, dil=2, mode=(0, 0)),
    dict(B=2, Cin=64, Cout=64, H=32, W=32, k=3, stride=2, pad=(1, 1, 1, 1), mode=(1, 1)),
    dict(B=2, Cin=64, Cout=64, H=61, W=121, k=3, stride=2, pad=(0, 1, 0, 1), mode=(0, 0)),
    dict(B=2, Cin=64, Cout=64, H=24, W=48, k=3, stride=2, pad=(1, 1, 1, 1), mode=(0, 1)),
    dict(B=2, Cin=3, Cout=64, H=32, W=32, k=1, act_out=1),
    dict(B=2, Cin=16, Cout=128, H=16, W=16, k=1),
    dict(B=2, Cin=128, Cout=16, H=16, W=16, k=1, ss=True, act_in=1),
    dict(B=2, Cin=64, Cout=3, H=64, W=64, k=1, ss=True, act_in=1),
    dict(B=2, Cin=64, Cout=512, H=32, W=32, k=1, ss=True, bias=False),
    dict(B=2, Cin=512, Cout=64, H=32, W=32, k=1, act_out=2, bias=False),
    dict(B=2, Cin=64, Cout=64, H=16, W=16, k=3, up=(32, 32), mode=(1, 1)),
    dict(B=2, Cin=64, Cout=64, H=28, W=60, k=3, up=(61, 121), mode=(0, 0)),
    dict(B=2, Cin=64, Cout=128, H=32, W=32, k=3, ss=True, act_in=1, res=True),
    dict(B=2, Cin=128, Cout=128, H=16, W=16, k=3, ss=True, act_in=1, act_out=2, badd=True),
    dict(B=2, Cin=64, Cout=64, H=61, W=121, k=1, ss=True, act_in=1, res=True),
    dict(B=2, Cin=128, Cout=64, H=30, W=60, k=1, bias=False),
    dict(B=3, Cin=64, Cout=2048, H=1, W=64, k=1, bias=False),
    dict(B=1, Cin=32, Cout=32, H=8, W=8, k=3, mode=(1, 1)),
]
# bf16x3 kernel (variant 6): every stride-1 3x3 shape class, prologue / epilogue features included
for _c in [dict(B=2, Cin=64, Cout=64, H=32, W=32, mode=(1, 1)), dict(B=2, Cin=64, Cout=64, H=24, W=48, mode=(0, 1)),
           dict(B=2, Cin=128, Cout=128, H=16, W=16, dil=2, mode=(1, 1)), dict(B=2, Cin=128, Cout=128, H=12, W=24, dil=3, mode=(0, 1)),
           dict(B=2, Cin=128, Cout=128, H=7, W=15, dil=2, mode=(0, 0)), dict(B=2, Cin=64, Cout=64, H=16, W=16, up=(32, 32), mode=(1, 1)),
           dict(B=2, Cin=64, Cout=64, H=28, W=60, up=(61, 121), mode=(0, 0)),
           dict(B=2, Cin=64, Cout=128, H=32, W=32, ss=True, act_in=1, res=True),
           dict(B=2, Cin=128, Cout=128, H=16, W=16, ss=True, act_in=1, act_out=2, badd=True),
           dict(B=2, Cin=40, Cout=72, H=20, W=36, mode=(0, 0)), dict(B=1, Cin=512, Cout=64, H=32, W=32, mode=(1, 1), ss=True)]:
    CONV_CASES.append(dict(k=3, variant=6, **_c))
    CONV_CASES.append(dict(k=3, variant=11, **_c))        # two-term fp16 split (f16x2) of the same kernel
    if not (_c.get('Cout') == 64 and _c.get('Cin') == 512):
        CONV_CASES.append(dict(k=3, variant=9, **_c))     # 32-cout tiles of the same kernel
        CONV_CASES.append(dict(k=3, variant=13, **_c))    # f16x2 with 32-cout tiles
        CONV_CASES.append(dict(k=3, variant=14, **_c))    # f16x2 with 256-pixel tiles (a wave owns 64 couts x 64 pixels)
# producer / consumer form of the f16x2 3x3 kernel (variants 15: 128-pixel tiles, 16: 256-pixel tiles; Cin_pad 64..256,
# Wout % 4 == 0): every padding mode, dilation, up-sampling gather, prologue / epilogue feature, ragged cout and pixel
# tiles, and launches with several tiles per block (more than 256 tiles)
PC_CASES = [dict(B=2, Cin=64, Cout=64, H=32, W=32, mode=(1, 1)), dict(B=2, Cin=64, Cout=64, H=24, W=48, mode=(0, 1)),
            dict(B=2, Cin=128, Cout=128, H=16, W=16, dil=2, mode=(1, 1)), dict(B=2, Cin=128, Cout=128, H=12, W=24, dil=3, mode=(0, 1)),
            dict(B=2, Cin=64, Cout=64, H=16, W=16, up=(32, 32), mode=(1, 1)),
            dict(B=2, Cin=64, Cout=128, H=32, W=32, ss=True, act_in=1, res=True),
            dict(B=2, Cin=128, Cout=128, H=16, W=16, ss=True, act_in=1, act_out=2, badd=True),
            dict(B=2, Cin=72, Cout=100, H=20, W=36, mode=(0, 0), ss=True), dict(B=1, Cin=256, Cout=64, H=32, W=32, mode=(1, 1), ss=True),
            dict(B=5, Cin=64, Cout=128, H=64, W=64, ss=True, act_in=1, res=True, mode=(1, 1)),
            dict(B=9, Cin=64, Cout=64, H=64, W=64, act_out=1, mode=(0, 0), bias=False)]
for _c in PC_CASES:
    CONV_CASES.append(dict(k=3, variant=15, **_c))
    CONV_CASES.append(dict(k=3, variant=16, **_c))
# phase-decomposed form of a 3x3 conv over an exactly 2x nearest-upsampled tensor (variant 17: four summed taps per
# output phase, f16x2 kernel): every padding mode, prologue / epilogue features, ragged source tiles and cout tiles
UP2_CASES = [dict(B=2, Cin=64, Cout=64, H=16, W=16, up=(32, 32), mode=(1, 1)), dict(B=2, Cin=64, Cout=64, H=16, W=16, up=(32, 32), mode=(0, 0)),
             dict(B=2, Cin=64, Cout=64, H=12, W=24, up=(24, 48), mode=(0, 1)), dict(B=2, Cin=128, Cout=128, H=16, W=16, up=(32, 32), mode=(1, 1), ss=True, act_in=1, res=True),
             dict(B=2, Cin=64, Cout=100, H=10, W=20, up=(20, 40), mode=(0, 0), act_out=2, badd=True),
             dict(B=3, Cin=64, Cout=64, H=32, W=32, up=(64, 64), mode=(1, 1), bias=False), dict(B=1, Cin=40, Cout=64, H=7, W=15, up=(14, 30), mode=(0, 0), ss=True)]
for _c in UP2_CASES:
    CONV_CASES.append(dict(k=3, variant=17, **_c))
# ... its resident-patch form (variant 18: Cin_pad <= 64, one block per source tile walks the four phases)
UP2R_CASES = [c for c in UP2_CASES if c["Cin"] <= 64] + [
    dict(B=2, Cin=64, Cout=64, H=64, W=64, up=(128, 128), mode=(1, 1), ss=True, act_in=1),
    dict(B=2, Cin=24, Cout=64, H=9, W=31, up=(18, 62), mode=(1, 0), ss=True, act_in=1, res=True),
    dict(B=1, Cin=64, Cout=128, H=16, W=32, up=(32, 64), mode=(0, 0), badd=True, act_out=2)]
for _c in UP2R_CASES:
    CONV_CASES.append(dict(k=3, variant=18, **_c))
# ... its quad-phase form (variant 20, conv3_up2q.inc: Cin_pad 16 .. 64; the patch staged once, four phases per block, the phase
# weights streamed through an LDS ring, statistics scratch in halves)
UP2Q_CASES = [c for c in UP2R_CASES if c["Cin"] in (16, 32, 48, 64) and not c.get("res")] + [
    dict(B=2, Cin=16, Cout=64, H=8, W=16, up=(16, 32), mode=(1, 1), ss=True, act_in=1),
    dict(B=3, Cin=48, Cout=72, H=21, W=13, up=(42, 26), mode=(0, 1), badd=True, act_out=1),
    dict(B=2, Cin=32, Cout=128, H=12, W=24, up=(24, 48), mode=(0, 0), ss=True)]
for _c in UP2Q_CASES:
    CONV_CASES.append(dict(k=3, variant=20, **_c))
# f16x2 3x3 kernel, 8-wave form for small launches (variant 19)
for _c in [dict(B=2, Cin=128, Cout=128, H=16, W=16, ss=True, act_in=1, mode=(1, 1)), dict(B=2, Cin=128, Cout=128, H=7, W=15, ss=True, mode=(0, 0), badd=True),
           dict(B=2, Cin=40, Cout=100, H=21, W=37, ss=True, act_in=1, res=True, mode=(0, 0)), dict(B=1, Cin=64, Cout=64, H=16, W=16, dil=2, mode=(1, 1), act_out=2)]:
    CONV_CASES.append(dict(k=3, variant=19, **_c))
# OCT8 layouts of the f16x2 3x3 kernel (variants 11 / 13 / 17 with | 0x100 input, | 0x200 output + residual): every padding
# mode, dilation, the phase form, prologue / epilogue features, ragged pixel tiles, ragged cout tiles (Cout % 64 != 0)
OCT_CASES = [dict(B=2, Cin=64, Cout=64, H=32, W=32, mode=(1, 1), ss=True, act_in=1, res=True),
             dict(B=2, Cin=128, Cout=128, H=16, W=16, dil=2, mode=(1, 1), ss=True, act_out=2),
             dict(B=2, Cin=72, Cout=104, H=21, W=37, mode=(0, 0), ss=True, act_in=1, badd=True, res=True),
             dict(B=3, Cin=128, Cout=128, H=7, W=15, mode=(0, 0), ss=True, act_out=2),
             dict(B=2, Cin=64, Cout=128, H=12, W=24, dil=3, mode=(0, 1)),
             dict(B=2, Cin=320, Cout=64, H=20, W=20, mode=(1, 1), ss=True, act_in=1)]
for _c in OCT_CASES:
    for _lay in (dict(xoct=True), dict(yoct=True), dict(xoct=True, yoct=True)):
        CONV_CASES.append(dict(k=3, variant=11, **_c, **_lay))
    CONV_CASES.append(dict(k=3, variant=13, xoct=True, yoct=True, **_c))
for _c in UP2_CASES:
    if _c["Cin"] % 8 == 0 and _c["Cout"] % 8 == 0:
        CONV_CASES.append(dict(k=3, variant=17, xoct=True, yoct=True, **_c))
# bf16x3 1x1 kernel (variant 7): channel counts below / above / not multiples of the 32-channel stage, ragged pixel
# counts, prologue and epilogue features
for _c in [dict(B=2, Cin=3, Cout=64, H=32, W=32, act_out=1), dict(B=2, Cin=16, Cout=128, H=16, W=16),
           dict(B=2, Cin=128, Cout=16, H=16, W=16, ss=True, act_in=1), dict(B=2, Cin=64, Cout=3, H=64, W=64, ss=True, act_in=1),
           dict(B=2, Cin=64, Cout=512, H=32, W=32, ss=True, bias=False), dict(B=2, Cin=512, Cout=64, H=32, W=32, act_out=2, bias=False),
           dict(B=2, Cin=64, Cout=64, H=61, W=121, ss=True, act_in=1, res=True), dict(B=2, Cin=128, Cout=64, H=30, W=60, bias=False),
           dict(B=3, Cin=64, Cout=2048, H=1, W=64, bias=False), dict(B=2, Cin=40, Cout=72, H=7, W=15, ss=True, badd=True),
           dict(B=1, Cin=96, Cout=96, H=20, W=36, ss=True, act_in=1, act_out=2)]:
    CONV_CASES.append(dict(k=1, variant=7, **_c))
# thin streaming projection (variant 12): <= 4 output channels
for _c in [dict(B=2, Cin=64, Cout=3, H=64, W=64, ss=True, act_in=1), dict(B=3, Cin=16, Cout=1, H=8, W=12),
           dict(B=2, Cin=37, Cout=4, H=20, W=36, ss=True, bias=False), dict(B=1, Cin=128, Cout=2, H=32, W=32, ss=True, act_in=1)]:
    CONV_CASES.append(dict(k=1, variant=12, **_c))
# the same kernel in its input-stationary form (variant code 8: blocks walk over 3 cout tiles, ragged last chunk)
for _c in [dict(B=2, Cin=64, Cout=512, H=32, W=32, ss=True, bias=False), dict(B=2, Cin=16, Cout=128, H=16, W=16),
           dict(B=2, Cin=40, Cout=200, H=7, W=15, ss=True, act_in=1, act_out=2, badd=True),
           dict(B=3, Cin=64, Cout=2048, H=1, W=64, bias=False), dict(B=1, Cin=64, Cout=448, H=61, W=121, ss=True, act_in=1, res=True)]:
    CONV_CASES.append(dict(k=1, variant=8, **_c))
for _v in range(6):   # every tile variant on the same problem
    CONV_CASES.append(dict(B=2, Cin=64, Cout=(32 if _v == 5 else 128 if _v in (0, 2) else 64), H=32, W=32, k=3,
                           mode=(1, 1), variant=_v))
    CONV_CASES.append(dict(B=2, Cin=64, Cout=(32 if _v == 5 else 128 if _v in (0, 2) else 64), H=20, W=36, k=1,
                           variant=_v, ss=True, act_in=1))


# Domain of the split-operand (f16x2) kernels -- variants 11 / 13 (3x3) and 7 / 8 (1x1): activation magnitudes from
# 1e-5 to 3e3 and beyond (the fixed x16 scale of round 1 overflowed at 4094 and lost the low term below 8e-3),
# different magnitudes per sample of one batch, activations spread over several decades, heavy-tailed weights,
# un-normalised inputs (no GroupNorm prologue) and inputs behind a scale/shift + Swish prologue.
DOMAIN_CASES = []
for _xs in (1e-12, 1e-5, 1e-3, 3e-2, 1.0, 60.0, 3e3, 1e6, 1e12):
    DOMAIN_CASES.append(dict(B=2, Cin=64, Cout=64, H=16, W=16, k=3, variant=11, xscale=_xs))
    DOMAIN_CASES.append(dict(B=2, Cin=64, Cout=64, H=16, W=16, k=3, variant=11, xscale=_xs, ss=True, act_in=1, res=False))
    DOMAIN_CASES.append(dict(B=2, Cin=64, Cout=128, H=16, W=16, k=1, variant=7, xscale=_xs))
    DOMAIN_CASES.append(dict(B=2, Cin=64, Cout=192, H=8, W=16, k=1, variant=8, xscale=_xs, ss=True))
DOMAIN_CASES += [
    dict(B=4, Cin=64, Cout=64, H=16, W=16, k=3, variant=11, sample_scales=[1e-4, 1.0, 2e3, 3e-2]),
    dict(B=4, Cin=64, Cout=64, H=16, W=16, k=3, variant=13, sample_scales=[1e-4, 1.0, 2e3, 3e-2], ss=True, act_in=1),
    dict(B=4, Cin=32, Cout=64, H=16, W=16, k=1, variant=7, sample_scales=[5e3, 1e-5, 1.0, 40.0]),
    dict(B=2, Cin=64, Cout=64, H=16, W=16, k=3, variant=11, heavy_w=True),
    dict(B=2, Cin=128, Cout=128, H=16, W=16, k=3, variant=11, heavy_w=True, xscale=500.0, act_out=2),
    dict(B=2, Cin=64, Cout=128, H=16, W=16, k=1, variant=7, heavy_w=True, xscale=1e-3),
    dict(B=2, Cin=64, Cout=64, H=16, W=16, k=3, variant=11, xdist="lognormal"),
    dict(B=2, Cin=64, Cout=64, H=16, W=16, k=1, variant=7, xdist="lognormal", heavy_w=True),
]


def conv_nonfinite_case():
    """One inf in sample 1: sample 0 stays finite and exact, sample 1's outputs and amax are non-finite."""
    L = _lib.lib()
    r = rng(5)
    B, C, H, W = 2, 64, 16, 16
    x = r.standard_normal((B, C, H, W)).astype(np.float32)
    x[1, 3, 5, 7] = np.inf
    w = (r.standard_normal((C, C, 3, 3)) / 24.0).astype(np.float32)
    xd = _dev(x)
    y = torch.zeros((B, C, H, W), dtype=torch.float32, device="cuda")
    amax = torch.zeros((B, 16), dtype=torch.int32, device="cuda")
    rc = L.lns_op_conv2d(xd.data_ptr(), B, C, H, W, H, W, _hp(w), None, C, 3, 1, 1, 1, 1, 1, 1, 1, 1, None, 0, 0, None, None,
                         y.data_ptr(), 11, _stream(), amax.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    out = y.cpu().numpy()
    am = amax.cpu().numpy().max(1).view(np.float32)           # integer max of the bit patterns (a NaN pattern wins)
    ref0 = lns_oracle.conv2d(x[:1], w, None, 1, 1, (1, 1, 1, 1), (1, 1))
    return bool(np.isfinite(out[0]).all() and rel_l2(out[0], ref0[0]) < 2e-6 and not np.isfinite(out[1]).all()
                and not np.isfinite(am[1]) and np.isfinite(am[0]))


def gn_case(B, C, HW, groups, eps, premul=False, seed=0):
    L = _lib.lib()
    r = rng(seed)
    x = (r.standard_normal((B, C, HW)) * 1.5 + 0.7).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(C)).astype(np.float32)
    b = (0.1 * r.standard_normal(C)).astype(np.float32)
    pm = (1 + 0.3 * r.standard_normal((B, C))).astype(np.float32) if premul else None
    xin = x * pm[:, :, None] if premul else x
    ref = lns_oracle.groupnorm(xin.reshape(B, C, HW, 1), groups, eps, g, b).reshape(B, C, HW)
    ss = torch.empty((B, C, 2), dtype=torch.float32, device="cuda")
    xd = _dev(x)
    pmd = _dev(pm) if premul else None
    rc = L.lns_op_groupnorm_stats(xd.data_ptr(), B, C, HW, groups, eps, _hp(g), _hp(b),
                                  pmd.data_ptr() if premul else None, ss.data_ptr(), _stream())
    assert rc == 0
    s = ss.cpu().numpy()
    out = x * s[:, :, 0:1] + s[:, :, 1:2]
    return rel_l2(out, ref)


def attention_case(B, heads, D, n, seed=0, qk_scale=1.0, v_scale=1.0, sample_scales=None, fp64=False):
    """rel. L2 of lns_op_attention vs the oracle.  qk_scale / v_scale: magnitude of q, k / of v (the f16x2 form scales all
    three by the sample's max |qkv|: a large v leaves q, k far below the maximum, a large q k^T saturates the softmax);
    sample_scales: per-sample factors on v; fp64: also returns the error against an fp64 evaluation."""
    L = _lib.lib()
    r = rng(seed)
    qkv = r.standard_normal((B, 3, heads, D, n)).astype(np.float32)
    qkv[:, :2] *= np.float32(qk_scale)
    qkv[:, 2] *= np.float32(v_scale)
    if sample_scales is not None:
        qkv[:, 2] *= np.asarray(sample_scales, np.float32).reshape(B, 1, 1, 1)
    q = np.ascontiguousarray(qkv[:, 0].transpose(0, 1, 3, 2))  # b h n d
    k = np.ascontiguousarray(qkv[:, 1].transpose(0, 1, 3, 2))
    v = np.ascontiguousarray(qkv[:, 2].transpose(0, 1, 3, 2))
    scale = float(D) ** -0.5
    attn = lns_oracle.softmax_rows(lns_oracle.bmm(q, k, transB=True), scale)
    ref = lns_oracle.bmm(attn, v).transpose(0, 1, 3, 2)  # b h d n
    qd = _dev(qkv)
    o = torch.full((B, heads, D, n), float("nan"), dtype=torch.float32, device="cuda")
    rc = L.lns_op_attention(qd.data_ptr(), B, heads, D, n, scale, o.data_ptr(), _stream())
    assert rc == 0
    out = o.cpu().numpy()
    assert np.isfinite(out).all()
    err = max(rel_l2(out[i], ref[i]) for i in range(B))
    if fp64:
        q64, k64, v64 = q.astype(np.float64), k.astype(np.float64), v.astype(np.float64)
        s64 = np.einsum("bhid,bhjd->bhij", q64, k64) * scale
        s64 -= s64.max(axis=-1, keepdims=True)
        p64 = np.exp(s64)
        p64 /= p64.sum(axis=-1, keepdims=True)
        r64 = np.einsum("bhij,bhjd->bhid", p64, v64).transpose(0, 1, 3, 2)
        return err, max(rel_l2(out[i], r64[i].astype(np.float32)) for i in range(B))
    return err


def sandwich_case(B, heads, C, H, W, instnorm=True, seed=0, uscale=1.0, kscale=1.0, plane_spread=0.0, heavy_k=False,
                  sample_scales=None, fp64=False):
    """rel. L2 of lns_op_fa_sandwich vs the oracle.  uscale / kscale: magnitude of the planes / of Kx, Ky; plane_spread:
    the channel planes of a sample differ by up to 10^(+-plane_spread) in magnitude (the f16x2 form scales by the
    SAMPLE's maximum); heavy_k: heavy-tailed (Student-t, 2 dof) kernels, i.e. large absolute row sums against their
    typical entry (the bound of U); sample_scales: per-sample factors; fp64: also returns the error against fp64."""
    L = _lib.lib()
    r = rng(seed)
    u = r.standard_normal((B, heads * C, H, W)).astype(np.float32)
    if plane_spread:
        u = (u * (10.0 ** r.uniform(-plane_spread, plane_spread, size=(B, heads * C, 1, 1)))).astype(np.float32)
    u = (u * np.float32(uscale)).astype(np.float32)
    if sample_scales is not None:
        u = (u * np.asarray(sample_scales, np.float32).reshape(B, 1, 1, 1)).astype(np.float32)
    if heavy_k:
        kx = (r.standard_t(2.0, size=(B, heads, H, H)) / np.sqrt(H) * kscale).astype(np.float32)
        ky = (r.standard_t(2.0, size=(B, heads, W, W)) / np.sqrt(W) * kscale).astype(np.float32)
    else:
        kx = (r.standard_normal((B, heads, H, H)) / np.sqrt(H) * kscale).astype(np.float32)
        ky = (r.standard_normal((B, heads, W, W)) / np.sqrt(W) * kscale).astype(np.float32)
    ref = lns_oracle.fa_contract(u, kx, ky, heads)
    if instnorm:
        ref = lns_oracle.groupnorm(ref, heads * C, 1e-5)
    ud, kxd, kyd = _dev(u), _dev(kx), _dev(ky)
    o = torch.full(u.shape, float("nan"), dtype=torch.float32, device="cuda")
    rc = L.lns_op_fa_sandwich(ud.data_ptr(), kxd.data_ptr(), kyd.data_ptr(), B, heads, C, H, W, 1e-5,
                              int(instnorm), o.data_ptr(), _stream())
    assert rc == 0
    out = o.cpu().numpy()
    assert np.isfinite(out).all()
    # every sample on its own (samples of one batch may differ by orders of magnitude)
    err = max(rel_l2(out[i], ref[i]) for i in range(B))
    if fp64:
        ud64 = u.astype(np.float64).reshape(B, heads, C, H, W)
        r64 = np.einsum("bhij,bhcjm,bhlm->bhcil", kx.astype(np.float64), ud64, ky.astype(np.float64)).reshape(B, heads * C, H, W)
        if instnorm:
            mu = r64.mean(axis=(2, 3), keepdims=True)
            r64 = (r64 - mu) / np.sqrt(r64.var(axis=(2, 3), keepdims=True) + 1e-5)
        return err, max(rel_l2(out[i], r64[i].astype(np.float32)) for i in range(B))
    return err


CV_F64 = 11


def conv2d_gpu(x, w, bias, k, pad=0, variant=-1):
    """lns_op_conv2d of a DEVICE tensor x on torch's current stream (no host synchronisation before the launch);
    stride 1, zero padding `pad` on every side.  Returns the device output."""
    L = _lib.lib()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    Ho, Wo = H + 2 * pad - (k - 1), W + 2 * pad - (k - 1)
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    rc = L.lns_op_conv2d(x.data_ptr(), B, Cin, H, W, H, W, _hp(w), _hp(bias), Cout, k, 1, 1, pad, pad, pad, pad, 0, 0,
                         None, 0, 0, None, None, y.data_ptr(), variant, _stream(), None)
    assert rc == 0, "lns_op_conv2d rc=%d" % rc
    return y


def build_models(args, weight_seed, variant=None):
    """(drop-in model on cuda, oracle) with identical deterministic weights (variant: lns_amd.filler, "stable")."""
    from helpers import synthetic_state_dict
    from lns_amd import dropin
    model = dropin.build_dynamics(args)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synthetic_state_dict(shapes, weight_seed, variant)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    orc = lns_oracle.OracleDynamics(args, sd)
    return model, orc


def layer_trace_compare(args, weight_seed, x, param=None):
    """Runs encode / one propagator step / decode with the engine's layer trace on and
    returns [(stage, layer name, rel_l2 vs oracle)] for every layer both sides know."""
    model, orc = build_models(args, weight_seed)
    rows = []
    xd = _dev(x)
    pd = _dev(param) if param is not None else None
    eng = model._engine(xd)
    # oracle with recording
    orc.ae.net.trace = {}
    z_ref = orc.x_to_z(x)
    enc_ref = dict(orc.ae.net.trace)
    orc.prop.net.trace = {}
    z1_ref = orc.prop.forward(z_ref, param)
    prop_ref = dict(orc.prop.net.trace)
    orc.ae.net.trace = {}
    y_ref = orc.z_to_x(z1_ref)
    dec_ref = dict(orc.ae.net.trace)
    eng.trace_enable(True)
    z = eng.encode(xd)
    torch.cuda.synchronize()
    for name, a in eng.trace():
        if name in enc_ref and enc_ref[name].shape == a.shape:
            rows.append(("encode", name, rel_l2(a, enc_ref[name])))
    rows.append(("encode", "OUT z0", rel_l2(z.cpu().numpy(), z_ref)))
    eng.trace_enable(True)
    z1 = eng.propagate(_dev(z_ref), pd)
    torch.cuda.synchronize()
    for name, a in eng.trace():
        if name in prop_ref and prop_ref[name].shape == a.shape:
            rows.append(("propagate", name, rel_l2(a, prop_ref[name])))
    rows.append(("propagate", "OUT z1", rel_l2(z1.cpu().numpy(), z1_ref)))
    eng.trace_enable(True)
    y = eng.decode(_dev(z1_ref))
    torch.cuda.synchronize()
    for name, a in eng.trace():
        if name in dec_ref and dec_ref[name].shape == a.shape:
            rows.append(("decode", name, rel_l2(a, dec_ref[name])))
    rows.append(("decode", "OUT y", rel_l2(y.cpu().numpy(), y_ref)))
    eng.trace_enable(False)
    return rows
