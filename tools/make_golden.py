#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs ONLY in the build container (needs /root/reference, imported on CPU through
oracle/ref_shim.py).  The reference can not travel to the GPU box, so what is
committed is DATA: for each case the preset name, seeds, and (sub-sampled)
outputs of the reference's own `LatentDynamics.predict`
(train_stage2_ns2d.py:143-158 and the SW / two-phase variants) in fp32 and in
fp64 (`model.double()`), on deterministic synthetic weights
(lns_amd.filler, a pure function of state_dict key/shape/seed) and inputs.

    python tools/make_golden.py            # all cases
    python tools/make_golden.py ns2d_mini  # one case
    python tools/make_golden.py ens        # reference fp32 ensembles of the long-horizon fixtures
    python tools/make_golden.py grads      # loss.backward() of the latent training rollout
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_models  # noqa: E402
from lns_amd import config, filler  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (preset, overrides, B, T, steps stored, spatial stride of stored fields)
CASES = {
    "ns2d_mini": ("ns2d_mini", {}, 2, 16, [1, 2, 4, 8, 16], 1),
    "ns2d_mini_zeros": ("ns2d_mini", {"is_periodic": False}, 2, 4, [1, 4], 1),
    "ns2d_mini_sa": ("ns2d_mini", {"use_fa": False}, 2, 4, [1, 4], 1),
    "ns2d_mini_nocoarse": ("ns2d_mini", {"disable_coarse_attn": True}, 2, 4, [1, 4], 1),
    "ns2d_mini_fourier": ("ns2d_mini", {"final_smoothing": True, "fourier_resolutions": [16],
                                        "Ly": 64, "Lx": 64, "resolution": 64,
                                        "latent_resolution": 8, "attn_resolutions": [16, 32]},
                          2, 2, [1, 2], 2),
    "ns2d_64": ("ns2d_64", {}, 2, 8, [1, 8], 2),
    "ns2d_128": ("ns2d_128", {}, 2, 64, [1, 2, 4, 8, 16, 32, 64], 4),
    "sw_half_periodic": ("sw_half_periodic", {}, 2, 16, [1, 4, 16], 4),
    "sw_96x192x5": ("sw_96x192x5", {}, 2, 16, [1, 4, 16], 4),
    "twophase": ("twophase", {}, 2, 8, [1, 8], 3),
    "twophase_cond": ("twophase_cond", {}, 2, 16, [1, 4, 16], 3),
    # long horizons of BASELINE configs 3, 4, 5 (T = 64 / 128 / 256): the per-horizon tolerance fixtures
    "sw_96x192x5_T64": ("sw_96x192x5", {}, 2, 64, [16, 32, 64], 4),
    "twophase_cond_T128": ("twophase_cond", {}, 2, 128, [16, 32, 64, 128], 3),
    "ns2d_128_T256": ("ns2d_128", {}, 2, 256, [64, 128, 192, 256], 4),
    # constructor flags the engine implements but no shipped config sets: attention blocks in the
    # ENCODER (modules/autoencoder2d.py:42-47; FABlock2D and, with use_fa=False, SABlock) and two residual blocks per level
    # (:36-40, decoder :105-109)
    "ns2d_mini_attn_enc": ("ns2d_mini", {"use_attn_enc": True}, 2, 4, [1, 4], 1),
    "ns2d_mini_attn_enc_sa": ("ns2d_mini", {"use_attn_enc": True, "use_fa": False}, 2, 4, [1, 4], 1),
    "ns2d_mini_res2": ("ns2d_mini", {"encoder_res_blocks": 2, "decoder_res_blocks": 2}, 2, 4, [1, 4], 1),
    # FULL horizons of BASELINE configs 3 / 4 / 5 on the `stable` filler variant (lns_amd.filler: the last convolution of
    # every residual branch of the propagator x 0.25 -> a non-expansive latent chain, tools/stable_filler_probe.py): the
    # reference's own fp32 runs stay within ~1e-5 of its fp64 run at every step, so the north star's 1e-4 is gated at EVERY
    # stored step, first to last
    "sw_96x192x5_T64_stable": ("sw_96x192x5", {}, 2, 64, [1, 8, 16, 24, 32, 40, 48, 56, 64], 4),
    "twophase_cond_T128_stable": ("twophase_cond", {}, 2, 128, [1, 16, 32, 48, 64, 80, 96, 112, 128], 3),
    "ns2d_128_T256_stable": ("ns2d_128", {}, 2, 256, [1, 32, 64, 96, 128, 160, 192, 224, 256], 4),
}
FILLER_VARIANT = {name: "stable" for name in CASES if name.endswith("_stable")}
WEIGHT_SEED = 1
INPUT_SEED = 7


def make_inputs(args, B):
    x = filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), INPUT_SEED)
    param = None
    if args.family == "twophase_cond":
        param = filler.uniform01("param", B, INPUT_SEED).astype(np.float32)
    return x, param


def run_reference(args, B, T, dtype, variant=None):
    model = ref_models.build_reference_dynamics(args, WEIGHT_SEED, dtype=dtype, variant=variant)
    if dtype == torch.float64 and args.family == "twophase_cond":
        # modules/cond_utils.py:34 casts the embedding to fp32 (`.float()`); the fp64 tie-breaker run re-casts it
        ref_models.patch_cond_embedding_f64()
    x, param = make_inputs(args, B)
    xt = torch.from_numpy(x).to(dtype)
    pt = torch.from_numpy(param).to(dtype) if param is not None else None
    with torch.no_grad():
        z0 = model.x_to_z(xt)
        lat, dec = [], []
        z = z0
        for _ in range(T):
            z = model.propagator(z, pt) if pt is not None else model.propagator(z)
            lat.append(z)
            dec.append(model.z_to_x(z))
        lat = torch.stack(lat, 1)
        dec = torch.stack(dec, 1)
        # cross-check that the loop above IS predict()
        y = model.predict(xt, min(T, 2), pt, to_x=True) if pt is not None else \
            model.predict(xt, min(T, 2), to_x=True)
        assert torch.equal(y, dec[:, : y.shape[1]]), "predict() != explicit loop"
    return z0.numpy(), lat.numpy(), dec.numpy(), model


def main(which):
    os.makedirs(OUT, exist_ok=True)
    for name in which:
        preset, over, B, T, steps, sub = CASES[name]
        args = config.preset(preset, **over)
        t0 = time.time()
        variant = FILLER_VARIANT.get(name)
        z0, lat, dec, model = run_reference(args, B, T, torch.float32, variant)
        z0d, latd, decd, _ = run_reference(args, B, T, torch.float64, variant)
        sidx = [s - 1 for s in steps]
        nparam = sum(v.numel() for v in model.state_dict().values())
        meta = dict(case=name, preset=preset, overrides=over, B=B, T=T, steps=steps, sub=sub,
                    weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED, filler_variant=variant or "default",
                    n_tensors=len(model.state_dict()), n_params=int(nparam),
                    torch=torch.__version__,
                    note="*_f64 arrays: computed by the reference in fp64, stored rounded to fp32; fields stored as y[:, steps-1, :, ::sub, ::sub]; norms over full fields")
        arrays = dict(
            meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
            z0=z0.astype(np.float32),
            z0_f64=z0d.astype(np.float32),
            lat=lat[:, sidx].astype(np.float32),
            lat_f64=latd[:, sidx].astype(np.float32),
            dec=dec[:, sidx][..., ::sub, ::sub].astype(np.float32),
            dec_f64=decd[:, sidx][..., ::sub, ::sub].astype(np.float32),
            # per-(b,t,c) L2 norm of every decoded frame, all T steps
            dec_norm=np.sqrt((dec.astype(np.float64) ** 2).sum((-1, -2))),
            dec_norm_f64=np.sqrt((decd ** 2).sum((-1, -2))),
            # the reference's own fp32 noise: rel-L2 of fp32 vs fp64 per step
            ref_self_err=np.sqrt(((dec - decd) ** 2).sum((0, 2, 3, 4)) / (decd ** 2).sum((0, 2, 3, 4))),
        )
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrays)
        print("%-20s B=%d T=%d  %.1f KB  self-err@T=%.2e  (%.1fs)" % (
            name, B, T, os.path.getsize(path) / 1024, arrays["ref_self_err"][-1], time.time() - t0))

    # state_dict key/shape manifests (data: names and shapes only), one per case
    man_path = os.path.join(OUT, "state_dict_manifest.json")
    man = {}
    if os.path.exists(man_path):
        with open(man_path) as f:
            man = json.load(f)
    for name in which:
        preset, over, _, _, _, _ = CASES[name]
        args = config.preset(preset, **over)
        model = ref_models.build_reference_dynamics(args, WEIGHT_SEED)
        man[name] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(man_path, "w") as f:
        json.dump(man, f, indent=0, sort_keys=False)


# ---------------------------------------------------------------------------------------------------------------
# Ensemble of the REAL reference's fp32 runs at the long horizons.  The random-init
# dynamics amplify rounding noise exponentially, so ONE fp32 run of the reference says little about how far a correct
# fp32 implementation may sit from the fp64 trajectory at t = 128 / 256.  Each member below is the reference's own
# `predict` loop in fp32 with a different but equally valid rounding history:
#   * execution variants: 8 threads / 1 thread, oneDNN convolutions on / off (ATen's im2col + GEMM path);
#   * the input field moved by +-1 ulp per element (sign pattern from the filler hash): a perturbation of 6e-8
#     relative, i.e. of the size of ONE fp32 rounding, applied once at t = 0.
# Stored: rel-L2 of every member against the UNPERTURBED fp64 run, per step over the full fields (`ref_ens_err`
# [K, T]) and over the sub-sampled fields at the stored steps, computed exactly as the parity tests compute the
# engine's distance (`ref_ens_err_sub` [K, len(steps)]).
# ---------------------------------------------------------------------------------------------------------------
ENSEMBLE_CASES = ["sw_96x192x5_T64", "twophase_cond_T128", "ns2d_128_T256", "ns2d_128",
                  "sw_96x192x5_T64_stable", "twophase_cond_T128_stable", "ns2d_128_T256_stable"]
N_ULP_MEMBERS = 6


def _ulp_perturb(x, k):
    u = filler.uniform01("ulp#%d" % k, x.size, INPUT_SEED).reshape(x.shape)
    up = np.nextafter(x, np.float32(np.inf), dtype=np.float32)
    dn = np.nextafter(x, np.float32(-np.inf), dtype=np.float32)
    return np.where(u >= 0.5, up, dn).astype(np.float32)


def _run_member(model, x, param, T):
    xt = torch.from_numpy(x)
    pt = torch.from_numpy(param) if param is not None else None
    with torch.no_grad():
        y = model.predict(xt, T, pt, to_x=True) if pt is not None else model.predict(xt, T, to_x=True)
    return y.numpy()


def add_ensembles(which):
    import contextlib
    for name in which:
        preset, over, B, T, steps, sub = CASES[name]
        args = config.preset(preset, **over)
        path = os.path.join(OUT, name + ".npz")
        old = dict(np.load(path))
        t0 = time.time()
        variant = FILLER_VARIANT.get(name)
        _, _, decd, _ = run_reference(args, B, T, torch.float64, variant)
        den_t = (decd ** 2).sum((0, 2, 3, 4))
        sidx = [s - 1 for s in steps]
        dsub = decd[:, sidx][..., ::sub, ::sub]
        den_s = (dsub ** 2).sum((0, 2, 3, 4))
        model = ref_models.build_reference_dynamics(args, WEIGHT_SEED, dtype=torch.float32, variant=variant)
        x, param = make_inputs(args, B)
        members = [("threads8_onednn", 8, True, None), ("threads1_onednn", 1, True, None),
                   ("threads8_aten", 8, False, None), ("threads1_aten", 1, False, None)]
        members += [("x_ulp_%d" % k, 8, True, k) for k in range(N_ULP_MEMBERS)]
        nthr0 = torch.get_num_threads()
        err, err_sub, desc = [], [], []
        for tag, nthr, onednn, k in members:
            torch.set_num_threads(nthr)
            ctx = contextlib.nullcontext() if onednn else torch.backends.mkldnn.flags(enabled=False)
            with ctx:
                y = _run_member(model, x if k is None else _ulp_perturb(x, k), param, T)
            torch.set_num_threads(nthr0)
            err.append(np.sqrt(((y - decd) ** 2).sum((0, 2, 3, 4)) / den_t))
            err_sub.append(np.sqrt(((y[:, sidx][..., ::sub, ::sub] - dsub) ** 2).sum((0, 2, 3, 4)) / den_s))
            desc.append(tag)
            print("  %-18s %-18s err@T=%.3e  (%.0fs)" % (name, tag, err[-1][-1], time.time() - t0), flush=True)
        err, err_sub = np.array(err), np.array(err_sub)
        # member 0 is the run the fixture already holds
        assert np.allclose(err[0], old["ref_self_err"], rtol=1e-6, atol=0), "member 0 != stored ref_self_err"
        old["ref_ens_err"] = err
        old["ref_ens_err_sub"] = err_sub
        old["ref_ens_desc"] = np.array(desc)
        np.savez_compressed(path, **old)
        print("%-20s ensemble of %d: max/min @T %.3e / %.3e" % (name, len(desc), err[:, -1].max(), err[:, -1].min()))


# ---------------------------------------------------------------------------------------------------------------
# Gradients of the latent training rollout: the REAL reference's `LatentDynamics.forward(z_in, z_out,
# F.smooth_l1_loss)` + `loss.backward()` (train_stage2_ns2d.py:126-141,213-215) on filler weights and filler latents,
# fp32 and fp64.  Stored per propagator parameter: the L2 norm of its gradient and every `sub`-th element (the D = 128
# models have 1.3 M propagator parameters), plus the loss, z_pred and the gradient w.r.t. z_in.
# ---------------------------------------------------------------------------------------------------------------
GRAD_CASES = {"ns2d_mini": ("ns2d_mini", 4, 3), "twophase": ("twophase", 3, 37), "sw_half_periodic": ("sw_half_periodic", 2, 37),
              "twophase_cond": ("twophase_cond", 3, 37)}


def make_grad_goldens(which=None):
    import torch.nn.functional as F
    for name, (preset, T, sub) in GRAD_CASES.items():
        if which and name not in which:
            continue
        args = config.preset(preset)
        B = 2
        out = {}
        t0 = time.time()
        for tag, dt in (("", torch.float32), ("_f64", torch.float64)):
            model = ref_models.build_reference_dynamics(args, WEIGHT_SEED, dtype=dt)
            model.train()
            ae = model.vq_ae if hasattr(model, "vq_ae") else model.ae
            with torch.no_grad():
                x, _ = make_inputs(args, B)
                zs = model.x_to_z(torch.from_numpy(x).to(dt))          # only for the latent shape
            c, h, w = zs.shape[1:]
            z_in = torch.from_numpy(filler.normal("z_in", (B, 1, c, h, w), INPUT_SEED) * np.float32(0.5)).to(dt).requires_grad_(True)
            z_out = torch.from_numpy(filler.normal("z_out", (B, T, c, h, w), INPUT_SEED) * np.float32(0.5)).to(dt)
            for p_ in ae.parameters():
                p_.requires_grad_(False)
            pt = None
            if args.family == "twophase_cond":
                pt = torch.from_numpy(filler.uniform01("param", B, INPUT_SEED).astype(np.float32)).to(dt)
                if dt == torch.float64:        # cond_utils.fourier_embedding casts to fp32 (`.float()`, :34): re-cast for the fp64 run
                    ref_models.patch_cond_embedding_f64()
            loss = model(z_in, z_out, pt, F.smooth_l1_loss) if pt is not None else model(z_in, z_out, F.smooth_l1_loss)
            loss.backward()
            with torch.no_grad():
                zp, z = [], z_in[:, 0]
                for _ in range(T):
                    z = model.propagator(z, pt) if pt is not None else model.propagator(z)
                    zp.append(z)
                zp = torch.stack(zp, 1)
            out["loss" + tag] = np.float64(loss.item())
            out["z_pred" + tag] = zp.numpy().astype(np.float32)
            out["grad_z_in" + tag] = z_in.grad.numpy().astype(np.float32)
            keys = []
            for k, p_ in model.propagator.named_parameters():
                g = p_.grad.detach().numpy().astype(np.float64).ravel()
                keys.append("propagator." + k)
                out["gnorm" + tag + ":" + "propagator." + k] = np.float64(np.sqrt((g ** 2).sum()))
                out["gsub" + tag + ":" + "propagator." + k] = g[::sub].astype(np.float32)
        meta = dict(case=name, preset=preset, B=B, T=T, sub=sub, weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED, keys=keys,
                    latent=[int(c), int(h), int(w)], loss="smooth_l1_loss (mean, beta=1)", z_scale=0.5)
        out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        path = os.path.join(OUT, "grads_" + name + ".npz")
        np.savez_compressed(path, **out)
        print("%-24s T=%d %d tensors  loss %.6f  %.1f KB (%.1fs)" % ("grads_" + name, T, len(keys), out["loss"], os.path.getsize(path) / 1024, time.time() - t0))


def make_op_goldens():
    """Standalone Fourier blocks (not reached by any model config): outputs of the
    reference's FourierBasicBlock / CondFourierBasicBlock on filler weights."""
    import ref_shim
    mods = ref_shim.load_reference()
    out = {}
    B, C, H, W, m1, m2 = 2, 8, 16, 24, 3, 5
    x = filler.normal("xop", (B, C, H, W), INPUT_SEED)
    cond = filler.normal("cop", (B, C), INPUT_SEED)
    for name, blk, extra in (
            ("fourier", mods["modules.basics"].FourierBasicBlock(C, C, modes=[m1, m2]), ()),
            ("cond_fourier", mods["modules.fourier_cond"].CondFourierBasicBlock(C, C, modes=[m1, m2]),
             (torch.from_numpy(cond),))):
        filler.load_into_torch_module(blk, WEIGHT_SEED)
        blk.eval()
        with torch.no_grad():
            y = blk(torch.from_numpy(x), *extra).numpy()
            y64 = blk.double()(torch.from_numpy(x).double(), *[e.double() for e in extra]).numpy()
        out[name + "_y"] = y.astype(np.float32)
        out[name + "_y_f64"] = y64.astype(np.float32)
        out[name + "_keys"] = np.array(sorted("%s:%s" % (k, "x".join(map(str, v.shape)))
                                              for k, v in blk.state_dict().items()))
    # general forms (basics.py:531-583): in_planes != planes with residual=False, the other registry activations
    gen = [("gen_6_10_gelu", 6, 10, "gelu", False, False), ("gen_8_8_silu_nores", 8, 8, "silu", False, False),
           ("gen_8_8_relu", 8, 8, "relu", True, False), ("gen_5_3_tanh", 5, 3, "tanh", False, False),
           ("gen_4_4_sigmoid", 4, 4, "sigmoid", True, False), ("gen_cond_6_10", 6, 10, "gelu", False, True)]
    for name, ci, co, act, res, cnd in gen:
        xg = filler.normal("xop_" + name, (B, ci, H, W), INPUT_SEED)
        cg = filler.normal("cop_" + name, (B, ci), INPUT_SEED)
        if cnd:
            blk = mods["modules.fourier_cond"].CondFourierBasicBlock(ci, co, modes=[m1, m2], residual=res)
            extra = (torch.from_numpy(cg),)
        else:
            blk = mods["modules.basics"].FourierBasicBlock(ci, co, modes=[m1, m2], activation=act, residual=res)
            extra = ()
        filler.load_into_torch_module(blk, WEIGHT_SEED)
        blk.eval()
        with torch.no_grad():
            y = blk(torch.from_numpy(xg), *extra).numpy()
            y64 = blk.double()(torch.from_numpy(xg).double(), *[e.double() for e in extra]).numpy()
        out[name + "_y"] = y.astype(np.float32)
        out[name + "_y_f64"] = y64.astype(np.float32)
        out[name + "_keys"] = np.array(sorted("%s:%s" % (k, "x".join(map(str, v.shape)))
                                              for k, v in blk.state_dict().items()))
    meta = dict(B=B, C=C, H=H, W=W, m1=m1, m2=m2, weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED,
                general=[dict(name=n, cin=ci, cout=co, act=a, residual=r, cond=c) for n, ci, co, a, r, c in gen])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "ops_fourier.npz"), **out)
    print("ops_fourier written")


COND_AE_CASES = {"cond_ae_mini": ("cond_ae_mini", 1), "twophase_cond_ae": ("twophase_cond_ae", 3)}


def make_cond_ae_goldens():
    """ConditionalSimpleAutoencoder (modules/autoencoder2d_nonsquared.py:279-305): encode(x, param) -> z,
    decode(z) -> y of the REAL reference on filler weights (zero_module tensors non-zero), fp32 and fp64."""
    import ref_shim
    mods = ref_shim.load_reference()
    man_path = os.path.join(OUT, "state_dict_manifest.json")
    with open(man_path) as f:
        man = json.load(f)
    for name, (preset, sub) in COND_AE_CASES.items():
        args = config.preset(preset)
        B = 2
        x, _ = make_inputs(args, B)
        param = filler.uniform01("param", B, INPUT_SEED).astype(np.float32)
        out = {}
        for tag, dt in (("", torch.float32), ("_f64", torch.float64)):
            model = mods["modules.autoencoder2d_nonsquared"].ConditionalSimpleAutoencoder(args)
            filler.load_into_torch_module(model, WEIGHT_SEED)
            model = model.to(dt).eval()
            if dt == torch.float64:
                # cond_utils.fourier_embedding casts to fp32 (`.float()`, :34): re-cast for the fp64 tie-breaker
                m = mods["modules.autoencoder2d_nonsquared"]
                orig = m.fourier_embedding
                m.fourier_embedding = lambda t, dim, max_period=10000, _o=orig: _o(t, dim, max_period).to(t.dtype)
            with torch.no_grad():
                z = model.encode(torch.from_numpy(x).to(dt), torch.from_numpy(param).to(dt))
                y = model.decode(z)
                assert torch.equal(model(torch.from_numpy(x).to(dt), torch.from_numpy(param).to(dt)), y)
            if dt == torch.float64:
                m.fourier_embedding = orig
            out["z" + tag] = z.numpy().astype(np.float32)
            out["y" + tag] = y.numpy()[..., ::sub, ::sub].astype(np.float32)
            out["y_norm" + tag] = np.sqrt((y.numpy().astype(np.float64) ** 2).sum((-1, -2)))
        man[name] = {k: list(v.shape) for k, v in model.state_dict().items()}
        meta = dict(case=name, preset=preset, overrides={}, B=B, sub=sub, weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED,
                    n_tensors=len(model.state_dict()), kind="cond_ae")
        out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "written", {k: v.shape for k, v in out.items() if k != "meta"})
    with open(man_path, "w") as f:
        json.dump(man, f, indent=0, sort_keys=False)


if __name__ == "__main__":
    # `ens [case ...]`: only (re)compute the reference ensembles of the long-horizon fixtures (adds arrays to them)
    if sys.argv[1:2] == ["ens"]:
        add_ensembles(sys.argv[2:] or ENSEMBLE_CASES)
        sys.exit(0)
    if sys.argv[1:2] == ["grads"]:
        make_grad_goldens(sys.argv[2:])
        sys.exit(0)
    which = sys.argv[1:] or list(CASES) + ["ops", "cond_ae"]
    if "cond_ae" in which:
        which.remove("cond_ae")
        make_cond_ae_goldens()
    if "ops" in which:
        which.remove("ops")
        make_op_goldens()
    if which:
        main(which)
    ens = [c for c in which if c in ENSEMBLE_CASES]
    if ens:                      # a regenerated long-horizon fixture gets its ensemble back
        add_ensembles(ens)
