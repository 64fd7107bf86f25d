"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against
(a) the CPU oracle on seeded inputs, (b) the committed golden fixtures generated
from the real reference, (c) size-independent properties at full bench size.

Tolerances (fp32; north star: decoded fields within 1e-4 rel-L2 of the reference):
  single kernel / single stage  2e-6 .. 2e-5   (fp32 accumulation-order noise)
  rollout, T <= 64               1e-4
"""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import load_golden, rel_l2, case_args, case_inputs  # noqa: E402

pytestmark = pytest.mark.gpu

KERNEL_TOL = 2e-6
STAGE_TOL = 2e-5
ROLLOUT_TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


EXPERIMENTAL_VARIANTS = (15, 16, 18, 19, 20)  # kernel forms only compiled with -DLNS_EXPERIMENTAL (csrc/conv3_{pc,up2r,up2q,w8}.inc)


def _experimental():
    from lns_amd import _lib
    return _lib.lib().lns_build_has(b"experimental") == 1


def test_native_library_is_loaded():
    _need_gpu()
    from lns_amd import _lib
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "liblns_hip.so" in maps


def _conv_cases():
    import gpu_checks as gc
    return list(enumerate(gc.CONV_CASES))


@pytest.mark.parametrize("idx,case", _conv_cases())
def test_conv_kernel(idx, case):
    _need_gpu()
    import gpu_checks as gc
    if case.get("variant") in EXPERIMENTAL_VARIANTS and not _experimental():
        pytest.skip("variant %d is only compiled with -DLNS_EXPERIMENTAL (measured slower; DESIGN.md 6d)" % case["variant"])
    if case.get("variant") == 20:
        try:
            err, _ = gc.conv_case(seed=idx, **case)
        except AssertionError as ex:         # the quad-phase form refuses patches that do not fit its 80 KB (LNS_EINVAL): the
            if "rc=-1" in str(ex):           # planner then keeps the per-phase form
                pytest.skip("patch does not fit the quad-phase form's LDS budget")
            raise
        assert err < KERNEL_TOL, (case, err)
        return
    err, _ = gc.conv_case(seed=idx, **case)
    assert err < KERNEL_TOL, (case, err)


def test_conv_tile_variants_same_bits():
    """The f16x2 3x3 kernel accumulates every output element in the same order whatever the block tile: 64 couts x 128
    pixels (11), 32 x 128 (13) and 64 x 256 (14) agree bit for bit -- full tiles, ragged edges and a ragged cout tile;
    so do the producer / consumer forms (15: 128-pixel tiles, 16: 256-pixel tiles; one or several tiles per block)."""
    _need_gpu()
    import gpu_checks as gc
    import numpy as np
    for kw in (dict(B=2, Cin=64, Cout=64, H=32, W=32, ss=True, act_in=1, res=True),
               dict(B=2, Cin=40, Cout=100, H=21, W=37, ss=True, act_in=1, badd=True, mode=(0, 0)),
               dict(B=2, Cin=64, Cout=128, H=28, W=60, up=(61, 121), mode=(0, 0))):
        ys = [gc.conv_case(k=3, variant=v, seed=5, ret_y=True, **kw)[2] for v in (11, 13, 14)]
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2]), kw
    if not _experimental():      # the measured-slower forms (15 / 16 / 19) only exist in -DLNS_EXPERIMENTAL builds
        return
    for kw in (dict(B=2, Cin=64, Cout=64, H=32, W=32, ss=True, act_in=1, res=True),
               dict(B=2, Cin=72, Cout=100, H=20, W=36, ss=True, act_in=1, badd=True, mode=(0, 0)),
               dict(B=2, Cin=64, Cout=128, H=30, W=60, up=(60, 120), mode=(0, 1), act_out=2),
               dict(B=2, Cin=128, Cout=128, H=16, W=16, dil=2, ss=True, act_in=1, act_out=2, mode=(1, 1)),
               dict(B=5, Cin=64, Cout=128, H=64, W=64, ss=True, act_in=1, res=True, mode=(1, 1))):
        ys = [gc.conv_case(k=3, variant=v, seed=6, ret_y=True, **kw)[2] for v in (11, 15, 16)]
        assert np.array_equal(ys[0], ys[1]), ("variant 15", kw)
        assert np.array_equal(ys[0], ys[2]), ("variant 16", kw)
    # the 8-wave form for small launches (19: two wave groups of 32 couts share one patch and one slab)
    for kw in (dict(B=2, Cin=64, Cout=64, H=16, W=16, ss=True, act_in=1, res=True, mode=(1, 1)),
               dict(B=2, Cin=72, Cout=100, H=20, W=36, ss=True, act_in=1, badd=True, mode=(0, 0)),
               dict(B=3, Cin=128, Cout=128, H=7, W=15, ss=True, act_out=2, mode=(0, 0)),
               dict(B=2, Cin=128, Cout=128, H=16, W=16, dil=4, ss=True, act_in=1, act_out=2, mode=(1, 1)),
               dict(B=2, Cin=40, Cout=64, H=12, W=24, mode=(0, 1), bias=False),
               dict(B=1, Cin=128, Cout=192, H=32, W=32, ss=True, act_in=1, res=True, mode=(1, 1))):
        y11 = gc.conv_case(k=3, variant=11, seed=8, ret_y=True, **kw)[2]
        y19 = gc.conv_case(k=3, variant=19, seed=8, ret_y=True, **kw)[2]
        assert np.array_equal(y11, y19), ("variant 19", kw)


def test_conv_oct8_layouts_same_bits():
    """The channel-octet-interleaved layout only changes where a value is loaded from / stored to: the f16x2 3x3 kernel gives
    the SAME bits with an OCT8 input, an OCT8 output (+ residual), or both, as with planar tensors (64- and 32-cout tiles,
    the phase form of the upsampling conv)."""
    _need_gpu()
    import gpu_checks as gc
    import numpy as np
    for kw in gc.OCT_CASES:
        for v in (11, 13):
            y0 = gc.conv_case(k=3, variant=v, seed=12, ret_y=True, **kw)[2]
            for lay in (dict(xoct=True), dict(yoct=True), dict(xoct=True, yoct=True)):
                y1 = gc.conv_case(k=3, variant=v, seed=12, ret_y=True, **kw, **lay)[2]
                assert np.array_equal(y0, y1), (v, kw, lay)
    for kw in gc.UP2_CASES:
        if kw["Cin"] % 8 or kw["Cout"] % 8:
            continue
        y0 = gc.conv_case(k=3, variant=17, seed=13, ret_y=True, **kw)[2]
        y1 = gc.conv_case(k=3, variant=17, seed=13, ret_y=True, xoct=True, yoct=True, **kw)[2]
        assert np.array_equal(y0, y1), kw


def test_upsampling_conv_quad_phase_same_bits():
    """The quad-phase form of the phase-decomposed upsampling conv (op-level variant 20, conv3_up2q.inc: one block stages the
    split source patch of all channels once and walks the four output phases, streaming the phase weights through a two-slot
    LDS ring; GroupNorm tile statistics through a half-size scratch) gives the bits of the per-phase form (17)."""
    _need_gpu()
    if not _experimental():
        pytest.skip("conv3_up2q.inc is only compiled with -DLNS_EXPERIMENTAL (measured neutral to slower; DESIGN.md 6e)")
    import gpu_checks as gc
    import numpy as np
    ran = 0
    for kw in gc.UP2Q_CASES:
        try:
            y20 = gc.conv_case(k=3, variant=20, seed=9, ret_y=True, **kw)[2]
        except AssertionError as ex:
            if "rc=-1" in str(ex):           # patch beyond the form's LDS budget: refused, the planner keeps the per-phase form
                continue
            raise
        y17 = gc.conv_case(k=3, variant=17, seed=9, ret_y=True, **kw)[2]
        assert np.array_equal(y17, y20), kw
        ran += 1
    assert ran >= 5, ran


def test_quad_phase_upsampling_conv_in_the_decoder(monkeypatch):
    """The planner's rule (decoder layers behind a 2x nearest upsample with <= 64 input channels: the last layer with its fused
    1x1 and GroupNorm tile statistics, the 64-channel UpSampleBlock) against the per-phase form (LNS_NO_UP2_QUAD=1, read when
    the FIRST plan of a process is built: a child process runs that arm): the decoded fields agree to rounding."""
    _need_gpu()
    if not _experimental():
        pytest.skip("conv3_up2q.inc is only compiled with -DLNS_EXPERIMENTAL (measured neutral to slower; DESIGN.md 6e)")
    import subprocess
    import sys
    from helpers import ROOT
    code = ("import sys, torch, numpy as np; sys.path[:0] = [%r, %r, %r]\n"
            "import gpu_checks as gc\nfrom lns_amd import config, filler\n"
            "args = config.preset('ns2d_128'); model, _ = gc.build_models(args, 1)\n"
            "x = torch.from_numpy(filler.normal('xq', (3, args.in_channels, args.Ly, args.Lx), 5)).cuda()\n"
            "y = model.predict(x, 2, to_x=True); torch.cuda.synchronize(); np.save(sys.argv[1], y.cpu().numpy())\n"
            % (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')))
    import tempfile
    outs = []
    for env_extra in ({"LNS_UP2_QUAD": "1"}, {}):
        f = tempfile.mktemp(suffix=".npy")
        env = dict(os.environ, **env_extra)
        p = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        outs.append(np.load(f))
        os.remove(f)
    # (not bit for bit: the quad-phase form tiles the source 8 x 16 where the per-phase form tiles it 4 x 32, so the GroupNorm
    #  tile statistics of the layer's output are merged from other partial sums -- a rounding-level difference, 1e-7)
    assert np.isfinite(outs[0]).all() and rel_l2(outs[0], outs[1]) < 1e-6, rel_l2(outs[0], outs[1])


def test_upsampling_conv_resident_patch_same_bits():
    """The resident-patch form of the phase-decomposed upsampling conv (op-level variant 18: the split source patch of all
    channels staged once per source tile, four phases per block) gives the bits of the per-phase form (17)."""
    _need_gpu()
    if not _experimental():
        pytest.skip("conv3_up2r.inc is only compiled with -DLNS_EXPERIMENTAL (measured slower; DESIGN.md 6d)")
    import gpu_checks as gc
    import numpy as np
    for kw in gc.UP2R_CASES:
        y17 = gc.conv_case(k=3, variant=17, seed=9, ret_y=True, **kw)[2]
        y18 = gc.conv_case(k=3, variant=18, seed=9, ret_y=True, **kw)[2]
        assert np.array_equal(y17, y18), kw


def _domain_cases():
    import gpu_checks as gc
    return list(enumerate(gc.DOMAIN_CASES))


@pytest.mark.parametrize("idx,case", _domain_cases())
def test_split_operand_domain(idx, case):
    """f16x2 kernels over the whole fp32-like input domain (activation magnitudes 1e-12 .. 1e12, per-sample
    magnitudes 5e3 vs 1e-5 in one batch, heavy-tailed weights, decades-wide activations; with and without a
    GroupNorm-style prologue): every SAMPLE within the kernel tolerance (2e-6 rel-L2) of the fp32 oracle AND of an
    fp64 evaluation of the same composition."""
    _need_gpu()
    import gpu_checks as gc
    err, _, e64 = gc.conv_case(seed=100 + idx, bias=False, fp64=True, **case)
    assert err < KERNEL_TOL, (case, err, e64)
    assert e64 < KERNEL_TOL, (case, err, e64)


def test_split_operand_extremes():
    """Full accuracy down to |x| ~ 1e-25; at the edge of fp32's own range (1e-33, beyond the clamp of the dynamic
    scale) the kernels degrade gracefully; a non-finite input gives non-finite outputs and a non-finite amax (loud,
    as in fp32), never a silently wrong field."""
    _need_gpu()
    import gpu_checks as gc
    err, _ = gc.conv_case(B=2, Cin=64, Cout=64, H=16, W=16, k=3, variant=11, xscale=1e-25, bias=False, seed=7)
    assert err < KERNEL_TOL
    err, _ = gc.conv_case(B=2, Cin=64, Cout=64, H=16, W=16, k=1, variant=7, xscale=1e-33, bias=False, seed=8)
    assert err < 1e-3
    assert gc.conv_nonfinite_case()


@pytest.mark.parametrize("case", [
    dict(B=2, C=64, HW=1024, groups=32, eps=1e-6), dict(B=2, C=128, HW=256, groups=1, eps=1e-5),
    dict(B=3, C=64, HW=4097, groups=8, eps=1e-5), dict(B=2, C=128, HW=105, groups=1, eps=1e-5, premul=True),
    dict(B=2, C=64, HW=16384, groups=8, eps=1e-5)])
def test_groupnorm_stats_kernel(case):
    _need_gpu()
    import gpu_checks as gc
    assert gc.gn_case(**case) < KERNEL_TOL


@pytest.mark.parametrize("case", [
    dict(B=2, heads=8, D=64, n=256), dict(B=2, heads=8, D=64, n=105), dict(B=2, heads=2, D=32, n=16),
    dict(B=1, heads=8, D=64, n=288)])
def test_attention_kernel(case):
    _need_gpu()
    import gpu_checks as gc
    assert gc.attention_case(**case) < KERNEL_TOL


@pytest.mark.parametrize("case", [
    dict(qk_scale=1e-3, v_scale=1e-3), dict(qk_scale=1.0, v_scale=1e3), dict(qk_scale=4.0), dict(qk_scale=0.05, v_scale=200.0),
    dict(sample_scales=[1e-4, 3e3])])
def test_attention_f16x2_domain(case):
    """q, k, v share one power-of-two scale per sample (max |qkv|): values far below the maximum, saturated and flat
    softmaxes, samples of very different magnitude -- against the oracle and an fp64 evaluation."""
    _need_gpu()
    import gpu_checks as gc
    for shape in (dict(B=2, heads=4, D=64, n=256), dict(B=2, heads=2, D=32, n=80)):
        err, e64 = gc.attention_case(seed=21, fp64=True, **shape, **case)
        assert err < KERNEL_TOL and e64 < KERNEL_TOL, (shape, case, err, e64)


@pytest.mark.parametrize("case", [
    dict(B=2, heads=8, C=64, H=64, W=64), dict(B=2, heads=8, C=64, H=32, W=32),
    dict(B=1, heads=8, C=64, H=24, W=48), dict(B=1, heads=8, C=64, H=48, W=96),
    dict(B=2, heads=2, C=32, H=16, W=16), dict(B=2, heads=2, C=32, H=8, W=8),
    dict(B=2, heads=8, C=64, H=64, W=64, instnorm=False)])
def test_fa_sandwich_kernel(case):
    _need_gpu()
    import gpu_checks as gc
    assert gc.sandwich_case(**case) < KERNEL_TOL


@pytest.mark.parametrize("case", [
    dict(uscale=1e-5), dict(uscale=3e3), dict(kscale=1e-3), dict(kscale=50.0), dict(heavy_k=True),
    dict(sample_scales=[1e-4, 2e3]), dict(plane_spread=2.0), dict(plane_spread=2.0, heavy_k=True, uscale=30.0)])
def test_fa_sandwich_f16x2_domain(case):
    """The f16x2 sandwich scales P by the sample's max |u|, Kx / Ky by their maxima and U by max|P| x the largest absolute
    row sum of Ky: no input range, samples independent, planes far below the sample's maximum still normalise correctly.
    Checked against the oracle and against an fp64 evaluation of the same contraction (the tie-breaker)."""
    _need_gpu()
    import gpu_checks as gc
    for shape in (dict(B=2, heads=4, C=16, H=64, W=64), dict(B=2, heads=2, C=8, H=48, W=96)):
        err, e64 = gc.sandwich_case(seed=11, fp64=True, **shape, **case)
        # planes three to four decades below the sample's maximum keep fewer bits of the split: 1e-5 there, 2e-6 otherwise
        tol = 1e-5 if case.get("plane_spread") else KERNEL_TOL
        assert err < tol and e64 < tol, (shape, case, err, e64)


# constructor flags no shipped config sets (modules/autoencoder2d.py:36-47): attention blocks in the encoder (FABlock2D /
# SABlock with use_fa=False), two residual blocks per level -- on the ns2d_mini shape
FLAG_VARIANTS = {"ns2d_mini+attn_enc": dict(use_attn_enc=True), "ns2d_mini+attn_enc_sa": dict(use_attn_enc=True, use_fa=False),
                 "ns2d_mini+res2": dict(encoder_res_blocks=2, decoder_res_blocks=2)}


@pytest.mark.parametrize("preset", ["ns2d_mini", "ns2d_128", "sw_half_periodic", "sw_96x192x5", "twophase",
                                    "twophase_cond"] + sorted(FLAG_VARIANTS))
def test_every_layer_matches_oracle(preset):
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    args = config.preset(preset.split("+")[0], **FLAG_VARIANTS.get(preset, {}))
    x = filler.normal("x", (2, args.in_channels, args.Ly, args.Lx), 7)
    param = filler.uniform01("param", 2, 7).astype(np.float32) if args.family == "twophase_cond" else None
    rows = gc.layer_trace_compare(args, 1, x, param)
    assert len(rows) > 20
    bad = [r for r in rows if not (r[2] < STAGE_TOL)]
    assert not bad, bad


GOLDEN_CASES = ["ns2d_mini", "ns2d_mini_zeros", "ns2d_mini_sa", "ns2d_mini_nocoarse", "ns2d_64", "ns2d_128",
                "sw_half_periodic", "sw_96x192x5", "twophase", "twophase_cond", "ns2d_mini_fourier",
                "ns2d_mini_attn_enc", "ns2d_mini_attn_enc_sa", "ns2d_mini_res2"]


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_rollout_matches_reference_golden(case):
    """HIP rollout vs the REAL reference's outputs (fixtures from tools/make_golden.py)."""
    _need_gpu()
    import gpu_checks as gc
    meta, g = load_golden(case)
    args = case_args(meta)
    model, _ = gc.build_models(args, meta["weight_seed"])
    x, param = case_inputs(meta, args)
    xd = torch.from_numpy(x).cuda()
    T = meta["T"]
    extra = (torch.from_numpy(param).cuda(),) if param is not None else ()
    dec, lat = model.predict(xd, T, *extra, to_x=True, return_latents=True)
    torch.cuda.synchronize()
    dec, lat = dec.cpu().numpy(), lat.cpu().numpy()
    assert dec.shape == (meta["B"], T, args.in_channels, args.Ly, args.Lx)
    sub = meta["sub"]
    report = []
    for i, s in enumerate(meta["steps"]):
        e_lat = rel_l2(lat[:, s - 1], g["lat"][:, i])
        e_dec = rel_l2(dec[:, s - 1][..., ::sub, ::sub], g["dec"][:, i])
        e64 = rel_l2(dec[:, s - 1][..., ::sub, ::sub], g["dec_f64"][:, i])
        report.append((s, e_lat, e_dec, e64, float(g["ref_self_err"][s - 1])))
        assert e_dec < ROLLOUT_TOL and e_lat < ROLLOUT_TOL, report
    print(case, report)
    nrm = np.sqrt((dec.astype(np.float64) ** 2).sum((-1, -2)))
    np.testing.assert_allclose(nrm, g["dec_norm"], rtol=5e-4)
    # to_x=False returns exactly the latents of the same rollout
    lat2 = model.predict(xd, T, *extra, to_x=False).cpu().numpy()
    assert np.array_equal(lat2, lat)


# Long horizons of BASELINE configs 3 / 4 / 5 (T = 64 / 128 / 256).  The random-init dynamics amplify rounding noise
# exponentially, so ONE fp32 run of the reference says little about how far a correct fp32 implementation may sit
# from the fp64 trajectory.  The fixtures therefore hold an ENSEMBLE of the REAL reference's fp32 runs
# (tools/make_golden.py `ens`: 8 / 1 threads, oneDNN / ATen convolutions, the input moved by +-1 ulp per element --
# ten members, each a valid fp32 evaluation of the same model) as rel-L2 to the reference's fp64 run per step
# (`ref_ens_err_sub`, computed on the same sub-sampled fields as the engine's distance below).  Measured spread at
# NS2d t=256: 1.5e-4 ... 1.6e-3 (x10 between members), SW-5ch t=64: 7e-4 ... 2.2e-3.
# Gate, against the reference's fp64 run: at most 2x the ensemble's MAXIMUM at that step (floor 2e-5) wherever that
# maximum is below 1e-2 (linearised regime); the north star's 1e-4 against the reference's fp32 run applies wherever
# the reference itself is reproducible to 3e-5; later steps are reported, not gated (chaotic regime: every member has
# left the fp64 trajectory).
LONG_CASES = ["sw_96x192x5_T64", "twophase_cond_T128", "ns2d_128_T256"]
LINEAR_REGIME = 1e-2
ENSEMBLE_FACTOR = 2.0


@pytest.mark.parametrize("case", LONG_CASES)
def test_long_horizon_rollout_vs_reference(case):
    _need_gpu()
    import gpu_checks as gc
    meta, g = load_golden(case)
    args = case_args(meta)
    model, _ = gc.build_models(args, meta["weight_seed"])
    x, param = case_inputs(meta, args)
    xd = torch.from_numpy(x).cuda()
    T = meta["T"]
    extra = (torch.from_numpy(param).cuda(),) if param is not None else ()
    dec = model.predict(xd, T, *extra, to_x=True)
    torch.cuda.synchronize()
    assert torch.isfinite(dec).all()
    dec = dec.cpu().numpy()
    sub = meta["sub"]
    noise = g["ref_self_err"]
    ens = g["ref_ens_err_sub"]                      # [members, stored steps]
    assert ens.shape[0] >= 8
    report, gated = [], 0
    for i, s in enumerate(meta["steps"]):
        f = dec[:, s - 1][..., ::sub, ::sub]
        e_dec, e64 = rel_l2(f, g["dec"][:, i]), rel_l2(f, g["dec_f64"][:, i])
        emax, emed = float(ens[:, i].max()), float(np.median(ens[:, i]))
        report.append((s, e_dec, e64, emed, emax))
        if emax <= LINEAR_REGIME:
            gated += 1
            assert e64 <= max(ENSEMBLE_FACTOR * emax, 2e-5), report
        if float(noise[s - 1]) <= 3e-5:
            assert e_dec < ROLLOUT_TOL, report
    print(case, "(step, vs fp32 run, vs fp64 run, ensemble median, ensemble max)", report)
    assert gated >= 2, report
    # per-frame norms over the whole horizon, where every member of the ensemble is reproducible
    nrm = np.sqrt((dec.astype(np.float64) ** 2).sum((-1, -2)))
    ok = g["ref_ens_err"].max(0) <= 1e-4
    np.testing.assert_allclose(nrm[:, ok], g["dec_norm_f64"][:, ok], rtol=1e-3)


# FULL horizons of BASELINE configs 3 / 4 / 5 in a regime where parity means something: on the `stable`
# filler variant (lns_amd.filler; the last convolution of every residual branch of the propagator x 0.25) the latent chain is
# non-expansive, every member of the real reference's ten-member fp32 ensemble stays within 3e-5 of its fp64 run at every
# step (tests/test_oracle_golden.py::test_stable_fixtures_are_reproducible_over_the_full_horizon), and the engine is gated
# at the north star's 1e-4 against the reference's fp32 run at EVERY stored step, first to last -- no `reported, not gated`.
STABLE_CASES = ["sw_96x192x5_T64_stable", "twophase_cond_T128_stable", "ns2d_128_T256_stable"]


@pytest.mark.parametrize("case", STABLE_CASES)
def test_full_horizon_rollout_vs_reference_stable(case):
    _need_gpu()
    import gpu_checks as gc
    from helpers import case_variant
    meta, g = load_golden(case)
    assert meta["steps"][-1] == meta["T"] and case_variant(meta) == "stable"
    args = case_args(meta)
    model, _ = gc.build_models(args, meta["weight_seed"], case_variant(meta))
    x, param = case_inputs(meta, args)
    xd = torch.from_numpy(x).cuda()
    T = meta["T"]
    extra = (torch.from_numpy(param).cuda(),) if param is not None else ()
    dec, lat = model.predict(xd, T, *extra, to_x=True, return_latents=True)
    torch.cuda.synchronize()
    assert torch.isfinite(dec).all()
    dec, lat = dec.cpu().numpy(), lat.cpu().numpy()
    sub = meta["sub"]
    ens = g["ref_ens_err_sub"]
    report = []
    for i, s in enumerate(meta["steps"]):
        f = dec[:, s - 1][..., ::sub, ::sub]
        e_dec, e64, e_lat = rel_l2(f, g["dec"][:, i]), rel_l2(f, g["dec_f64"][:, i]), rel_l2(lat[:, s - 1], g["lat"][:, i])
        report.append((s, e_dec, e64, e_lat, float(ens[:, i].max())))
        assert float(ens[:, i].max()) <= 3e-5, report          # the premise: the reference itself is reproducible here
        assert e_dec < ROLLOUT_TOL and e64 < ROLLOUT_TOL and e_lat < ROLLOUT_TOL, report
    print(case, "(step, decoded vs fp32 run, vs fp64 run, latent vs fp32 run, reference ensemble max)", report)
    nrm = np.sqrt((dec.astype(np.float64) ** 2).sum((-1, -2)))
    np.testing.assert_allclose(nrm, g["dec_norm_f64"], rtol=5e-4)


@pytest.mark.timeout(300)
def test_rccl_all_gather_with_one_rank_equals_the_shard():
    """RCCL readiness on a one-GPU box: init_process_group("nccl", world_size=1) in THIS process, then
    lns_amd.parallel.EndGatherRollout with the collective forced -- the HIP rollout writes the shard, ONE
    all_gather_into_tensor (RCCL) copies it into the separate [world*B, T, C, H, W] buffer, which must equal the shard bit
    for bit and the ungathered rollout.  (Nothing crosses xGMI with one rank: no scaling claim.)"""
    _need_gpu()
    import datetime
    import socket
    import torch.distributed as dist
    import gpu_checks as gc
    from lns_amd import config, filler, parallel
    args = config.preset("ns2d_mini")
    model, _ = gc.build_models(args, 1)
    B, T = 2, 4
    x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 7)).cuda()
    ref = model.predict(x, T, to_x=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            timeout=datetime.timedelta(seconds=120), device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        eng = model._engine(x)

        def rollout(xx, oo):
            eng.rollout(xx, T, to_x=True, out=oo)
        r = parallel.EndGatherRollout(rollout, (args.in_channels, args.Ly, args.Lx), B, T, x.device, gather=True,
                                      collective_at_world1=True)
        r.run(x)
        torch.cuda.synchronize()
        r.finish_timing()
        full = r.assemble()
        assert full.data_ptr() != r.out.data_ptr() and tuple(full.shape) == (B, T, args.in_channels, args.Ly, args.Lx)
        assert torch.equal(full, r.out) and torch.equal(full, ref)
        assert len(r.exposed_ms) == 1 and r.exposed_ms[0] >= 0.0
        maps = open("/proc/self/maps").read()
        assert "librccl" in maps or "libnccl" in maps or "libtorch_hip" in maps
    finally:
        dist.destroy_process_group()


def test_batch_shard_equivalence_and_b1():
    """Trajectories are independent: a sample's result must not depend on the batch it
    rides in (this is what makes trajectory sharding over GPUs collective-free), and B=1
    works (the reference's squeeze() bug, SURVEY.md F9, is not reproduced)."""
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    args = config.preset("ns2d_64")
    model, _ = gc.build_models(args, 3)
    x = torch.from_numpy(filler.normal("xb", (5, args.in_channels, args.Ly, args.Lx), 11)).cuda()
    full = model.predict(x, 3, to_x=True).cpu().numpy()
    part = model.predict(x[1:3].contiguous(), 3, to_x=True).cpu().numpy()
    one = model.predict(x[4:5].contiguous(), 3, to_x=True).cpu().numpy()
    assert np.array_equal(full[1:3], part)
    assert np.array_equal(full[4:5], one)


def test_module_api_surface():
    """encode / decode / forward / SimpleCNN.forward / rollout_latent compose to predict."""
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    args = config.preset("ns2d_mini")
    model, orc = gc.build_models(args, 1)
    x = torch.from_numpy(filler.normal("x", (3, args.in_channels, args.Ly, args.Lx), 7)).cuda()
    z = model.x_to_z(x)
    # the propagator's parameters require grad (LatentDynamics.forward trains them): its plain forward has no backward
    # and says so when autograd is on, instead of returning a tensor that only fails at loss.backward()
    with pytest.raises(Exception) as ei:
        model.propagator(z)
    assert "no backward" in str(ei.value)
    assert not any(p.requires_grad for p in model.vq_ae.parameters()) and all(p.requires_grad for p in model.propagator.parameters())
    torch.set_grad_enabled(False)
    z1 = model.propagator(z)
    y1 = model.z_to_x(z1)
    full = model.predict(x, 2, to_x=True)
    assert torch.equal(full[:, 0], y1)
    eng = model._engine(x)
    out, zl = eng.rollout_latent(z, 2, to_x=True)
    assert torch.equal(out, full)
    assert torch.equal(zl, model.propagator(z1))
    assert rel_l2(model.vq_ae(x).cpu().numpy(), orc.z_to_x(orc.x_to_z(x.cpu().numpy()))) < STAGE_TOL
    # standalone autoencoder / propagator objects with their own engines
    from lns_amd.modules.autoencoder2d import SimpleAutoencoder
    from lns_amd.train_stage2_ns2d import SimpleCNN
    ae = SimpleAutoencoder(args)
    ae.load_state_dict(model.vq_ae.state_dict())
    assert torch.equal(ae.cuda().encode(x), z)
    cnn = SimpleCNN(args.latent_dim, args.prop_n_block, args.prop_n_embd, args.dilation)
    cnn.load_state_dict(model.propagator.state_dict())
    assert torch.equal(cnn.cuda()(z), z1)
    # weights changed after the first call are picked up
    getattr(model.propagator.out_proj, "1").bias.add_(1.0)
    assert not torch.equal(model.propagator(z), z1)
    torch.set_grad_enabled(True)


def test_check_finite_names_the_layer():
    """lns_check_finite: clean after a normal rollout; after a rollout whose input holds an inf it names the first
    layer whose output went non-finite (and the sample) instead of leaving silently broken fields."""
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    from lns_amd._lib import LnsError
    args = config.preset("ns2d_mini")
    model, _ = gc.build_models(args, 1)
    x = filler.normal("x", (3, args.in_channels, args.Ly, args.Lx), 7)
    xd = torch.from_numpy(x).cuda()
    eng = model._engine(xd)
    y = model.predict(xd, 3, to_x=True)
    eng.check_finite(3, xd.device)
    assert torch.isfinite(y).all()
    x[2, 1, 5, 9] = np.inf
    xb = torch.from_numpy(x).cuda()
    y = model.predict(xb, 3, to_x=True)
    with pytest.raises(LnsError, match=r"non-finite values in the output of .*encoder.*sample 2"):
        eng.check_finite(3, xb.device)
    assert torch.isfinite(y[:2]).all() and not torch.isfinite(y[2]).all()
    assert torch.equal(y[:2], model.predict(xd, 3, to_x=True)[:2])       # the clean samples are untouched, bit for bit


def test_check_finite_state_and_coverage():
    """lns_check_finite resolves the last run's records through the workspace handed in (never through
    remembered plan / arena pointers), refuses a stale or foreign workspace with LNS_ESTATE, accepts 'cuda' for
    'cuda:0', and sees a NaN born in the LAST layer of a plan (the plan outputs record an amax too, the thin final
    projection included); "track_nonfinite" changes no bit."""
    _need_gpu()
    import ctypes
    import gpu_checks as gc
    from lns_amd import config, filler, _lib
    from lns_amd._lib import LnsError
    args = config.preset("ns2d_mini")
    from helpers import synthetic_state_dict
    model, _ = gc.build_models(args, 1)
    sd = synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1)
    x = filler.normal("x", (3, args.in_channels, args.Ly, args.Lx), 7)
    xd = torch.from_numpy(x).cuda()
    eng = model._engine(xd)
    y0 = model.predict(xd, 3, to_x=True)
    eng.check_finite(3, "cuda")                                  # torch.device('cuda') != torch.device('cuda:0')
    eng.check_finite(3)
    ws = eng._ws[(3, xd.device)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = eng._L
    other = torch.empty_like(ws)
    assert L.lns_check_finite(eng._h, 3, other.data_ptr(), other.numel(), stream) == _lib.LNS_ESTATE    # foreign workspace
    assert L.lns_check_finite(eng._h, 2, ws.data_ptr(), ws.numel(), stream) == _lib.LNS_ESTATE          # other batch
    assert L.lns_check_finite(eng._h, 3, ws.data_ptr(), 64, stream) == _lib.LNS_ENOMEM                  # records not covered
    assert L.lns_check_finite(eng._h, 3, ws.data_ptr(), ws.numel(), stream) == _lib.LNS_OK
    # lns_finalize_weights drops the plans: the records of the last run are gone with them (was: use after free)
    eng.load_weights({k: t.detach().cpu().numpy() for k, t in model.state_dict().items()}, xd.device.index)
    assert L.lns_check_finite(eng._h, 3, ws.data_ptr(), ws.numel(), stream) == _lib.LNS_ESTATE
    assert b"no run" in L.lns_last_error(eng._h)
    # a NaN born in the very last layer of the decoder (its bias): the plan output is recorded
    last = sorted((k for k in sd if k.startswith("vq_ae.decoder.model.") and k.endswith(".bias")),
                  key=lambda k: int(k.split(".")[3]))[-1]
    bad = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    bad[last][0] = float("inf")
    model.load_state_dict(bad)
    yb = model.predict(xd, 2, to_x=True)
    assert not torch.isfinite(yb).all()
    with pytest.raises(LnsError, match=r"non-finite values in the output of .*decoder"):
        model._engine(xd).check_finite(3, xd.device)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    eng = model._engine(xd)
    eng.set_option("track_nonfinite", 1)
    y1 = model.predict(xd, 3, to_x=True)
    eng.check_finite(3)
    assert torch.equal(y0, y1)
    xi = x.copy()
    xi[1, 0, 3, 3] = np.inf
    model.predict(torch.from_numpy(xi).cuda(), 5, to_x=True)
    with pytest.raises(LnsError, match=r"non-finite"):
        eng.check_finite(3)
    eng.set_option("track_nonfinite", 0)


def test_op_conv_amax_runs_on_the_callers_stream():
    """lns_op_conv2d takes the input's per-sample maximum on the CALLER's stream -- with x produced on a
    non-blocking side stream right before the call, a maximum taken on the null stream would be too small and the
    f16x2 scale would overflow."""
    _need_gpu()
    import gpu_checks as gc
    rng = np.random.default_rng(3)
    B, C, H, W = 4, 64, 32, 32
    w = (rng.standard_normal((64, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    side = torch.cuda.Stream()
    big = torch.from_numpy((rng.standard_normal((B, C, H, W)) * 3e4).astype(np.float32)).cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        x = torch.zeros_like(big)
        for _ in range(20):                       # keep the side stream busy so that a null-stream amax would run early
            x = x * 0.5 + big * 0.5
        y = gc.conv2d_gpu(x, w, None, 3, pad=1, variant=gc.CV_F64)
        side.synchronize()
    ref = gc.conv2d_gpu(x.clone(), w, None, 3, pad=1, variant=-1)      # fp32 MFMA kernel
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert rel_l2(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6


def test_overlapped_rollout_equals_single_stream():
    """The multi-stream rollout (propagator stream + decode streams, kernels of different steps co-resident
    on the CUs) must reproduce the single-stream rollout bit for bit at the bench batch size."""
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    args = config.preset("ns2d_128")
    model, _ = gc.build_models(args, 1)
    x = filler.normal("xfull", (64, args.in_channels, args.Ly, args.Lx), 5)
    xd = torch.from_numpy(x).cuda()
    eng = model._engine(xd)
    eng.set_option("decode_group", 1)
    eng.timing_enable(True)            # diagnostics mode = everything on the caller's stream
    ref = model.predict(xd, 6, to_x=True).clone()
    eng.timing_enable(False)
    # scheduling options never change a bit: steps decoded per launch set (batch 64 * k, ragged last group: 6 = 4 + 2),
    # number of decode streams, stream priority of the latent chain, single stream
    # ... and the FABlock's in_proj -> sandwich -> to_out chain issued per group of samples (fa_chunk_mb: 537 MB at 64 x 64 -> 16
    # samples per chunk at 128 MB; 60 MB: 7 samples, ragged last chunk, and the 32 x 32 block is chunked too)
    for opts in (dict(decode_group=1), dict(decode_group=4), dict(decode_group=4, decode_streams=2, prop_priority=1),
                 dict(decode_group=3, overlap=0), dict(decode_group=0), dict(decode_group=1, decode_streams=3, overlap=1, prop_priority=0),
                 dict(fa_chunk_mb=128), dict(fa_chunk_mb=60, decode_group=2), dict(fa_chunk_mb=60, overlap=0), dict(fa_chunk_mb=0, overlap=1)):
        for k, v in opts.items():
            eng.set_option(k, v)
        for _ in range(2):
            y = model.predict(xd, 6, to_x=True)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), opts


@pytest.mark.gpu
@pytest.mark.parametrize("B", [3, 16])
def test_fablock_in_proj_inside_the_sandwich(B):
    """FABlock2D at 64 x 64 planes (64 channels) and 32 x 32 planes (128 channels): in_proj computed inside the sandwich kernel
    (csrc/fa_fused.inc, the default) against the
    three-kernel form (in_proj convolution -> sandwich -> to_out) on the same latents: rounding-level agreement of the decoded
    fields, both within the oracle's tolerance; the plane groups a block walks (1, 2, 4) and the batch-dependent block order
    (B % 8 == 0 or not) never change a bit."""
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    args = config.preset("ns2d_128")
    model, orc = gc.build_models(args, 1)
    x = filler.normal("xfa", (B, args.in_channels, args.Ly, args.Lx), 7)
    xd = torch.from_numpy(x).cuda()
    eng = model._engine(xd)
    z = eng.encode(xd)
    try:
        eng.set_option("fa_fused", 0)
        y3 = eng.decode(z).clone()
        eng.set_option("fa_fused", 2)
        yf = eng.decode(z).clone()
        assert torch.isfinite(yf).all()
        for form in (1, 3):                     # the single-buffered and the generic form of the 64 x 64 kernel: the same arithmetic
            eng.set_option("fa_fused", form)
            assert torch.equal(eng.decode(z), yf), form
        eng.set_option("fa_fused", 2)
        assert rel_l2(yf.cpu().numpy(), y3.cpu().numpy()) < 1e-6
        assert not torch.equal(yf, y3)          # (the fused form really ran: another summation order)
        for gpb in (1, 2, 4, 0):
            eng.set_option("fa_fused_gpb", gpb)
            assert torch.equal(eng.decode(z), yf), gpb
        # a sample's result does not depend on the batch around it
        sub = eng.decode(z[1:2].contiguous())
        assert torch.equal(sub, yf[1:2])
        ref = orc.z_to_x(z[:2].cpu().numpy())
        assert rel_l2(yf[:2].cpu().numpy(), ref) < STAGE_TOL * 2
        assert rel_l2(y3[:2].cpu().numpy(), ref) < STAGE_TOL * 2
    finally:
        eng.set_option("fa_fused", 2)
        eng.set_option("fa_fused_gpb", 0)


def test_full_size_rollout_properties():
    """BASELINE config 2 shape (B=64, T=64 is the bench; here T=4 to bound memory/time):
    finite, per-sample independent (bitwise) and close to the oracle on a sub-batch."""
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    args = config.preset("ns2d_128")
    model, orc = gc.build_models(args, 1)
    x = filler.normal("xfull", (64, args.in_channels, args.Ly, args.Lx), 5)
    xd = torch.from_numpy(x).cuda()
    y = model.predict(xd, 4, to_x=True)
    assert torch.isfinite(y).all()
    sub = model.predict(xd[10:12].contiguous(), 4, to_x=True)
    assert torch.equal(y[10:12], sub)
    ref = orc.predict(x[10:12], 4, to_x=True)
    assert rel_l2(sub.cpu().numpy(), ref) < STAGE_TOL * 2


@pytest.mark.parametrize("preset,B,T", [("sw_96x192x5", 64, 2), ("twophase_cond", 32, 3)])
def test_full_size_properties_configs_3_and_4(preset, B, T):
    """BASELINE configs 3 (SW 96x192x5, B=64: 48x96 sandwich planes, one block per CU) and 4 (conditional two-phase,
    B=32: ragged 61x121 tiles, the conditional embedding pool) at their FULL batch, a few steps: finite; a sub-batch is
    bit-identical to the same trajectories inside the full batch (what makes trajectory sharding exact); the sub-batch
    is within 4e-5 of the oracle; the multi-stream rollout equals the single-stream one bit for bit under several
    schedules (reference shapes: modules/autoencoder2d_nonsquared.py:148-247)."""
    _need_gpu()
    import gpu_checks as gc
    from lns_amd import config, filler
    args = config.preset(preset)
    model, orc = gc.build_models(args, 1)
    x = filler.normal("xfull_" + preset, (B, args.in_channels, args.Ly, args.Lx), 5)
    param = filler.uniform01("pfull", B, 5).astype(np.float32) if args.family == "twophase_cond" else None
    xd = torch.from_numpy(x).cuda()
    pd = torch.from_numpy(param).cuda() if param is not None else None
    extra = (pd,) if pd is not None else ()
    eng = model._engine(xd)
    eng.set_option("decode_group", 1)
    eng.timing_enable(True)            # diagnostics mode = everything on the caller's stream
    ref = model.predict(xd, T, *extra, to_x=True).clone()
    eng.timing_enable(False)
    assert torch.isfinite(ref).all()
    for opts in (dict(decode_group=1), dict(decode_group=2, decode_streams=2), dict(decode_group=0), dict(decode_group=1, decode_streams=3, overlap=1)):
        for k, v in opts.items():
            eng.set_option(k, v)
        y = model.predict(xd, T, *extra, to_x=True)
        torch.cuda.synchronize()
        assert torch.equal(y, ref), (preset, opts)
    lo = B // 3
    sub_extra = (pd[lo:lo + 2].contiguous(),) if pd is not None else ()
    sub = model.predict(xd[lo:lo + 2].contiguous(), T, *sub_extra, to_x=True)
    assert torch.equal(ref[lo:lo + 2], sub), preset
    o = orc.predict(x[lo:lo + 2], T, param=param[lo:lo + 2] if param is not None else None, to_x=True)
    assert rel_l2(sub.cpu().numpy(), o) < 4e-5, preset


# ---------------------------------------------------------------------------------------------------
# "next rows" (SURVEY 8f): fused denormalise + relative-L2 metric, bulk dataset encode
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 5, 3, 32, 32), (2, 4, 5, 24, 48), (2, 3, 2, 61, 121), (1, 2, 3, 7, 9)])
def test_metric_rel_l2_matches_oracle(shape):
    """lns_metric_rel_l2 vs the oracle restatement of denormalize + relative_lp_loss (frame- and sequence-wise),
    including a plane whose ground truth is ~0 (the eps clamp)."""
    _need_gpu()
    import lns_oracle
    from lns_amd import metrics
    r = np.random.default_rng(3)
    y = r.standard_normal(shape).astype(np.float32)
    yh = (y + 0.05 * r.standard_normal(shape)).astype(np.float32)
    mean, std = 0.37, 1.9
    y[0, 0, 0] = -mean / std                     # denormalises to ~0 -> the eps clamp path
    f_ref, s_ref = lns_oracle.rollout_metrics(yh, y, mean, std)
    f, s = metrics.relative_l2(torch.from_numpy(yh).cuda(), torch.from_numpy(y).cuda(), mean, std)
    f, s = f.cpu().numpy().astype(np.float64), s.cpu().numpy().astype(np.float64)
    big = f_ref > 1e3                            # clamp-dominated entries: compare in log scale
    assert np.allclose(f[~big], f_ref[~big], rtol=2e-5, atol=1e-7)
    assert np.allclose(np.log(f[big]), np.log(f_ref[big]), rtol=1e-2) if big.any() else True
    assert np.allclose(s, s_ref, rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ns2d", "sw", "twophase"])
def test_metric_matches_reference_golden(name):
    """Fused denormalise + relative-L2 kernel (scalar stats; per-channel stats; closed-wall zeroing + VOF clamp)
    against the real reference's dataset.denormalize + relative_lp_loss (tests/golden/metrics.npz)."""
    _need_gpu()
    import os
    from helpers import GOLDEN, METRIC_STATS, metric_inputs
    from lns_amd import metrics
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    st = METRIC_STATS[name]
    yh, y = metric_inputs(name)
    a, b = torch.from_numpy(yh).cuda(), torch.from_numpy(y).cuda()
    if name == "twophase":
        f, s = metrics.relative_l2(a, b, **metrics.twophase_spec(st["vel_mean"], st["vel_std"], st["prs_mean"], st["prs_std"]))
    else:
        f, s = metrics.relative_l2(a, b, st["mean"], st["std"])
    f, s = f.cpu().numpy().astype(np.float64), s.cpu().numpy().astype(np.float64)
    f_ref, s_ref = g[name + "_frame_f64"], g[name + "_seq_f64"]
    big = f_ref > 1e3                            # eps-clamp-dominated entries
    assert np.allclose(f[~big], f_ref[~big], rtol=2e-5, atol=1e-7)
    assert np.allclose(np.log(f[big]), np.log(f_ref[big]), rtol=1e-2) if big.any() else True
    assert np.allclose(s, s_ref, rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["ns2d_mini", "twophase_cond"])
def test_teacher_forced_loss_matches_oracle(preset):
    """LatentDynamics.forward(z_in, z_out[, param], loss_fn): loss value of the latent rollout against pre-encoded
    targets (train_stage2_ns2d.py:126-141), vs the oracle.  With autograd enabled the unconditional model runs the HIP
    training rollout (same loss, differentiable: test_training_rollout_*), plain and conditional."""
    _need_gpu()
    import gpu_checks as gc
    import lns_oracle
    import torch.nn.functional as F
    from lns_amd import config, filler
    args = config.preset(preset)
    model, orc = gc.build_models(args, 1)
    B, T = 2, 4
    c, h, w = model._engine(torch.zeros(1, device="cuda")).latent_shape()
    z_in = filler.normal("tf_zin", (B, 1, c, h, w), 5).astype(np.float32) * 0.5
    z_out = filler.normal("tf_zout", (B, T, c, h, w), 6).astype(np.float32) * 0.5
    param = filler.uniform01("tf_param", B, 5).astype(np.float32) if args.family == "twophase_cond" else None
    ref = lns_oracle.teacher_forced_loss(orc, z_in, z_out, param)
    zi, zo = torch.from_numpy(z_in).cuda(), torch.from_numpy(z_out).cuda()
    extra = (torch.from_numpy(param).cuda(),) if param is not None else ()
    with torch.no_grad():
        loss = model(zi, zo, *extra, F.smooth_l1_loss)
    assert abs(float(loss) - ref) <= 2e-5 * abs(ref), (float(loss), ref)
    lt = model(zi, zo, *extra, F.smooth_l1_loss)
    assert lt.requires_grad and abs(float(lt) - ref) <= 2e-5 * abs(ref), (float(lt), ref)


@pytest.mark.gpu
def test_encode_dataset_matches_oracle():
    """Bulk pre-encoding (chunks of frames through the encoder, dataset normalisation applied first)."""
    _need_gpu()
    import gpu_checks as gc
    import lns_oracle
    from lns_amd import config, filler, metrics
    args = config.preset("ns2d_mini")
    model, orc = gc.build_models(args, 1)
    frames = filler.normal("frames", (11, args.in_channels, args.Ly, args.Lx), 9) * 2.0 + 0.3
    z = metrics.encode_dataset(model.vq_ae, frames, chunk=4, mean=0.3, std=2.0).numpy()
    ref = lns_oracle.encode_dataset(orc.ae, frames, chunk=4, mean=0.3, std=2.0)
    assert z.shape == ref.shape
    assert rel_l2(z, ref) < STAGE_TOL
    with pytest.raises(RuntimeError):
        metrics.relative_l2(torch.zeros(1, 1, 1, 2, 2), torch.zeros(1, 1, 1, 2, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("preset,spec", [("sw_half_periodic", "sw"), ("twophase", "twophase")])
def test_encode_dataset_per_channel_statistics(preset, spec):
    """The other datasets' pre-encoding: per-channel normalisation (dataset/Stage2_SW.py:74-105: u / v / pres; dataset/
    twophase_flow_stage2.py:304-337: velocities / pressure / VOF untouched, chunks of 32) + chunked encode, vs oracle."""
    _need_gpu()
    import gpu_checks as gc
    import lns_oracle
    from lns_amd import config, filler, metrics
    args = config.preset(preset)
    model, orc = gc.build_models(args, 1)
    N = 5
    raw = filler.normal("frames_" + spec, (N, args.in_channels, args.Ly, args.Lx), 9)
    if spec == "sw":
        kw = metrics.sw_norm(0.4, 2.1, -0.2, 1.7, 9.5, 0.6)
        kw["chunk"] = 2
    else:
        kw = metrics.twophase_norm(0.013, 0.21, 310.0, 180.0)
        raw[:, 3] = 0.5 + 0.3 * raw[:, 3]
    # raw fields in physical units: x_phys = x * std + mean
    m = np.asarray(kw["mean"], np.float32).reshape(1, -1, 1, 1)
    sd = np.asarray(kw["std"], np.float32).reshape(1, -1, 1, 1)
    frames = (raw * sd + m).astype(np.float32)
    z = metrics.encode_dataset(model._ae, frames, **kw).numpy()
    ref = lns_oracle.encode_dataset(orc.ae, frames, chunk=kw["chunk"], mean=kw["mean"], std=kw["std"], eps=kw["eps"])
    assert z.shape == ref.shape == (N,) + tuple(model._engine(torch.zeros(1, device="cuda")).latent_shape())
    assert rel_l2(z, ref) < STAGE_TOL
    with pytest.raises(ValueError):
        metrics.encode_dataset(model._ae, frames, mean=[0.0, 1.0], std=1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cond_ae_mini", "twophase_cond_ae"])
def test_conditional_autoencoder_matches_reference_golden(case):
    """ConditionalSimpleAutoencoder.encode(x, param) / decode / forward on the HIP engine vs the REAL
    reference's outputs (tests/golden/<case>.npz) and vs the oracle; B=1 works; the unconditional entry point refuses."""
    _need_gpu()
    from helpers import manifest
    from lns_amd import filler
    from lns_amd._lib import LnsError
    from lns_amd.modules.autoencoder2d_nonsquared import ConditionalSimpleAutoencoder
    import lns_oracle
    meta, g = load_golden(case)
    args = case_args(meta)
    sd = filler.synthetic_state_dict({k: tuple(v) for k, v in manifest()[case].items()}, meta["weight_seed"])
    model = ConditionalSimpleAutoencoder(args)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    x = filler.normal("x", (meta["B"], args.in_channels, args.Ly, args.Lx), meta["input_seed"])
    param = filler.uniform01("param", meta["B"], meta["input_seed"]).astype(np.float32)
    xd, pd = torch.from_numpy(x).cuda(), torch.from_numpy(param).cuda()
    z = model.encode(xd, pd)
    y = model.decode(z)
    sub = meta["sub"]
    assert rel_l2(z.cpu().numpy(), g["z"]) < STAGE_TOL
    assert rel_l2(y.cpu().numpy()[..., ::sub, ::sub], g["y"]) < STAGE_TOL * 2
    assert rel_l2(z.cpu().numpy(), g["z_f64"]) < STAGE_TOL
    assert torch.equal(model(xd, pd), y)
    orc = lns_oracle.OracleCondAutoencoder(args, sd, "")
    assert rel_l2(z.cpu().numpy(), orc.encode(x, param)) < STAGE_TOL
    assert torch.equal(model.encode(xd[1:2].contiguous(), pd[1:2].contiguous()), z[1:2])      # batch-independent, B = 1
    assert rel_l2(model.encode(xd, pd.flip(0)).cpu().numpy(), g["z"]) > 1e-3                   # conditioning is live
    with pytest.raises((LnsError, TypeError)):
        model._engine(xd).encode(xd)                                                           # param is required


@pytest.mark.gpu
def test_packed_fp32_corun():
    """Regression guard of the co-residency finding: the product is built without packed-fp32 VALU
    instructions (CPU test test_device_code_has_no_packed_fp32_arithmetic); this runs the standalone reproducer
    (tools/pk_hazard: the exact v_pk_fma_f32 ... op_sel pair hipcc emitted in the fp32 1x1 kernel vs scalar v_fma_f32 in
    the SAME wave, alone and beside an MFMA-dense kernel on a second stream).  What must hold: the scalar form -- the one
    the product uses -- never differs from itself.  The packed counts are reported: the minimal pair does NOT reproduce
    the corruption (0 in 1.7e9 words, profiles/r02_pk_hazard_standalone.log), while the engine-level pair stress does as
    soon as the CURRENT kernels are built with packed-fp32 ops (profiles/r02_pair_stress_packed_build.log: 1024..25600
    wrong words on the fp32 1x1 side next to a 16-bit-MFMA kernel, 0 for fp32 | fp32 pairs) and never with the shipped
    build (profiles/r02_pair_stress_shipped_build.log)."""
    _need_gpu()
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pk_hazard"))
    import run as pk
    res = pk.run(rounds=6, launches=8)
    print("pk_hazard", res)
    for mode in ("alone", "corun"):
        assert res[mode]["scalar_mismatches"] == 0, res
    assert res["alone"]["packed_mismatches"] == 0, res        # single-stream runs never failed


@pytest.mark.gpu
def test_engine_pair_stress_is_clean_on_the_shipped_build():
    """The regression test that actually guards the co-residency finding: the in-engine pair stress
    (lns_op_conv_pair_stress: two convolutions on two streams, every output word compared with the kernel's solo
    result) on the pairs that corrupt 1 024 .. 113 663 words as soon as the victim's x*s+t is a v_pk_fma_f32 with
    op_sel broadcast (profiles/r03_pk_experiment.txt: with only that instruction pair made scalar the packed build
    is clean; with the pair written by hand into non-aliasing registers it still fails).  Shipped build: 0 words."""
    _need_gpu()
    import ctypes
    from lns_amd import _lib
    L = _lib.lib()
    L.lns_op_conv_pair_stress.restype = ctypes.c_int
    L.lns_op_conv_pair_stress.argtypes = [ctypes.c_int] * 13 + [ctypes.POINTER(ctypes.c_longlong)] * 2
    for name, B, H, W, a, b, rounds in (("f16 1x1 | fp32 1x1", 32, 64, 64, (64, 64, 1, 7), (64, 64, 1, 1), 6),
                                        ("split 3x3 | fp32 1x1 64->512", 32, 64, 64, (64, 64, 3, 6), (64, 512, 1, -1), 4)):
        ma, mb = ctypes.c_longlong(0), ctypes.c_longlong(0)
        rc = L.lns_op_conv_pair_stress(B, H, W, a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], rounds, 6, ctypes.byref(ma), ctypes.byref(mb))
        assert rc == 0 and ma.value == 0 and mb.value == 0, (name, rc, ma.value, mb.value)


# ---- the latent TRAINING rollout, forward + backward through time on the HIP engine ---------------------
GRAD_CASES = ["ns2d_mini", "twophase", "sw_half_periodic", "twophase_cond"]
GRAD_TOL = 1e-4          # rel-L2 per parameter tensor against the REAL reference's loss.backward()


def _grad_setup(case):
    import torch.nn.functional as F
    import gpu_checks as gc
    from lns_amd import config, filler
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grads_%s.npz" % case))
    meta = json.loads(bytes(g["meta"]).decode())
    args = config.preset(meta["preset"])
    model, _ = gc.build_models(args, meta["weight_seed"])
    B, T = meta["B"], meta["T"]
    c, h, w = meta["latent"]
    z_in = torch.from_numpy(filler.normal("z_in", (B, 1, c, h, w), meta["input_seed"]) * np.float32(meta["z_scale"])).cuda()
    z_out = torch.from_numpy(filler.normal("z_out", (B, T, c, h, w), meta["input_seed"]) * np.float32(meta["z_scale"])).cuda()
    if args.family == "twophase_cond":       # forward(z_in, z_out, param, loss_fn)
        prm = torch.from_numpy(filler.uniform01("param", B, meta["input_seed"]).astype(np.float32)).cuda()
        return g, meta, model, z_in, z_out, (prm, F.smooth_l1_loss)
    return g, meta, model, z_in, z_out, (F.smooth_l1_loss,)


@pytest.mark.parametrize("case", GRAD_CASES)
def test_training_rollout_gradients_match_reference(case):
    """LatentDynamics.forward(z_in, z_out, loss_fn) with autograd enabled + loss.backward() (train_stage2_ns2d.py:126-141,
    213-215) on the HIP engine: loss, z_pred, every propagator parameter's gradient and the gradient of z_in against the
    REAL reference's (tools/make_golden.py grads), circular / zero / half-periodic padding, dilation 2 and 3."""
    _need_gpu()
    g, meta, model, z_in, z_out, tail = _grad_setup(case)
    for p_ in model._ae.parameters():
        p_.requires_grad_(False)
    z_in.requires_grad_(True)
    loss = model(z_in, z_out, *tail)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * abs(float(g["loss"])) + 1e-7, (loss.item(), float(g["loss"]))
    with torch.no_grad():
        zp = model._engine(z_in).rollout_latent(z_in[:, 0].contiguous(), meta["T"], param=tail[0] if len(tail) == 2 else None, to_x=False)[0]
    assert rel_l2(zp.cpu().numpy(), g["z_pred"]) < 2e-5
    params = dict(model.named_parameters())
    sub = meta["sub"]
    worst = []
    for k in meta["keys"]:
        gr = params[k].grad
        assert gr is not None and torch.isfinite(gr).all(), k
        gh = gr.detach().cpu().numpy().astype(np.float64).ravel()
        ref32, ref64 = g["gsub:" + k].astype(np.float64), g["gsub_f64:" + k].astype(np.float64)
        own = rel_l2(ref32, ref64)                               # the reference's own fp32 deviation on this tensor
        e32, e64 = rel_l2(gh[::sub], ref32), rel_l2(gh[::sub], ref64)
        en = abs(np.sqrt((gh ** 2).sum()) / float(g["gnorm_f64:" + k]) - 1.0)
        worst.append((max(e32, e64), k, e32, e64, own, en))
        assert e64 <= max(GRAD_TOL, 3.0 * own), (k, e32, e64, own)
        assert e32 <= max(GRAD_TOL, 3.0 * own), (k, e32, e64, own)
        assert en <= max(GRAD_TOL, 3.0 * own), (k, en)
    print(case, "worst tensors:", sorted(worst, reverse=True)[:3])
    assert rel_l2(z_in.grad.cpu().numpy(), g["grad_z_in_f64"]) <= GRAD_TOL
    assert all(p_.grad is None for p_ in model._ae.parameters())     # frozen autoencoder: untouched


def test_training_rollout_properties():
    """Size-independent properties of the HIP backward: bit-reproducible; linear in the batch (the gradient of the mean
    loss over 4 trajectories is the mean of the two halves' gradients); consistent with a central finite difference of
    the loss along a random direction in parameter space; an optimiser step changes the next loss (parameters are read
    from their device tensors at every call)."""
    _need_gpu()
    g, meta, model, z_in, z_out, (loss_fn,) = _grad_setup("ns2d_mini")
    for p_ in model._ae.parameters():
        p_.requires_grad_(False)
    names = [k for k in meta["keys"]]
    params = dict(model.named_parameters())
    z4_in = torch.cat([z_in, z_in.flip(0) * 0.7 + 0.1], 0)
    z4_out = torch.cat([z_out, z_out.flip(0) * 0.9 - 0.05], 0)

    def grads_of(zi, zo):
        for k in names:
            params[k].grad = None
        loss = model(zi, zo, loss_fn)
        loss.backward()
        return loss.item(), {k: params[k].grad.detach().clone() for k in names}
    l_a, g_a = grads_of(z4_in, z4_out)
    l_b, g_b = grads_of(z4_in, z4_out)
    assert l_a == l_b and all(torch.equal(g_a[k], g_b[k]) for k in names)                 # deterministic
    _, g_1 = grads_of(z4_in[:2], z4_out[:2])
    _, g_2 = grads_of(z4_in[2:], z4_out[2:])
    for k in names:
        assert rel_l2((0.5 * (g_1[k] + g_2[k])).cpu().numpy(), g_a[k].cpu().numpy()) < 2e-5, k
    # directional derivative: (L(theta + eps v) - L(theta - eps v)) / (2 eps)  vs  <grad, v>
    gen = torch.Generator(device="cuda").manual_seed(3)
    v = {k: torch.randn(params[k].shape, device="cuda", generator=gen) * params[k].detach().abs().mean() for k in names}
    dot = sum((g_a[k] * v[k]).sum().item() for k in names)
    eps = 2e-2

    with torch.no_grad():
        for k in names:
            params[k].add_(eps * v[k])
    lp = model(z4_in, z4_out, loss_fn).item()
    with torch.no_grad():
        for k in names:
            params[k].add_(-2 * eps * v[k])
    lm = model(z4_in, z4_out, loss_fn).item()
    with torch.no_grad():
        for k in names:
            params[k].add_(eps * v[k])
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - dot) <= 2e-2 * abs(dot) + 1e-6, (fd, dot)
    # one SGD step on the propagator lowers this batch's loss
    l0, g0 = grads_of(z4_in, z4_out)
    with torch.no_grad():
        for k in names:
            params[k].add_(-0.05 * g0[k])
    assert model(z4_in, z4_out, loss_fn).item() < l0
