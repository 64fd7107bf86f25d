#!/usr/bin/env python3
"""How fast does the REAL reference's latent chain amplify rounding noise under a given weight filler variant?

Build container only (imports /root/reference through oracle/ref_shim.py).  For a preset and a list of filler variants
(lns_amd.filler.VARIANTS) it runs the reference's own propagator loop (train_stage2_ns2d.py:147-156 without the decode) in
fp64, in fp32, and in fp64 from an input moved by 1e-6 relative, and prints the latent rel-L2 distances per horizon: the
growth factor of a perturbation and the reference's own fp32-vs-fp64 distance.  This is the measurement behind the choice
of the `stable` variant (tools/make_golden.py, the `*_stable` fixtures): full-horizon parity can only be
gated at the north star's 1e-4 where the reference itself is reproducible to a few 1e-5 at the final step.

    python tools/stable_filler_probe.py ns2d_128 256 default stable
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_models  # noqa: E402
from lns_amd import config, filler  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).sum().sqrt() / (b ** 2).sum().sqrt())


def chain(model, z, pt, T, every, moves=None):
    out = {}
    with torch.no_grad():
        for t in range(1, T + 1):
            zn = model.propagator(z, pt) if pt is not None else model.propagator(z)
            if moves is not None and (t % every == 0 or t == T):
                moves[t] = rel(zn, z)            # how far one step still moves the state (a fixed point would make late steps a weak test)
            z = zn
            if t % every == 0 or t == T:
                out[t] = z.clone()
    return out


def main():
    preset, T = sys.argv[1], int(sys.argv[2])
    variants = sys.argv[3:] or ["default"]
    args = config.preset(preset)
    x = filler.normal("x", (2, args.in_channels, args.Ly, args.Lx), 7)
    param = filler.uniform01("param", 2, 7).astype(np.float32) if args.family == "twophase_cond" else None
    every = max(1, T // 8)
    if param is not None:
        ref_models.patch_cond_embedding_f64()
    for var in variants:
        m64 = ref_models.build_reference_dynamics(args, 1, dtype=torch.float64, variant=var)
        m32 = ref_models.build_reference_dynamics(args, 1, dtype=torch.float32, variant=var)
        with torch.no_grad():
            z64 = m64.x_to_z(torch.from_numpy(x).double())
            z32 = m32.x_to_z(torch.from_numpy(x))
        p64 = torch.from_numpy(param).double() if param is not None else None
        p32 = torch.from_numpy(param) if param is not None else None
        moves = {}
        a = chain(m64, z64, p64, T, every, moves)
        b = chain(m32, z32, p32, T, every)
        zp = z64 * (1.0 + 1e-6 * torch.from_numpy(filler.normal("pert", tuple(z64.shape), 3)).double())
        c = chain(m64, zp, p64, T, every)
        print("%s / %s: latent rel-L2 to the fp64 chain  (|z| rms at T: %.3f)" % (preset, var, float(a[T].pow(2).mean().sqrt())))
        for t in a:
            print("  t=%4d   fp32 run %.3e   1e-6 perturbation -> %.3e (x%.1f)   |z_t - z_(t-1)| / |z| = %.3f" % (
                t, rel(b[t], a[t]), rel(c[t], a[t]), rel(c[t], a[t]) / 1e-6, moves[t]))


if __name__ == "__main__":
    main()
