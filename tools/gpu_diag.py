#!/usr/bin/env python3
"""GPU-box diagnostic: per-kernel and per-layer parity tables against the oracle.
Writes gpurun_out/diag.txt.  Usage: python tools/gpu_diag.py [ops] [trace:<preset>] ..."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import gpu_checks as gc  # noqa: E402
from lns_amd import config, filler  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "diag.txt"), "a")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def run_ops():
    say("== conv cases")
    for i, c in enumerate(gc.CONV_CASES):
        try:
            e, shp = gc.conv_case(seed=i, **c)
            say("%-4s conv %-90s -> %s err %.2e" % ("OK" if e < 2e-6 else "BAD", c, shp, e))
        except Exception as ex:
            say("FAIL conv", c, repr(ex)[:300])
    say("== groupnorm stats")
    for c in [dict(B=2, C=64, HW=1024, groups=32, eps=1e-6), dict(B=2, C=128, HW=256, groups=1, eps=1e-5),
              dict(B=3, C=64, HW=4097, groups=8, eps=1e-5), dict(B=2, C=128, HW=105, groups=1, eps=1e-5, premul=True),
              dict(B=2, C=64, HW=16384, groups=8, eps=1e-5)]:
        try:
            e = gc.gn_case(**c)
            say("%-4s gn %s err %.2e" % ("OK" if e < 2e-6 else "BAD", c, e))
        except Exception as ex:
            say("FAIL gn", c, repr(ex)[:300])
    say("== attention")
    for c in [dict(B=2, heads=8, D=64, n=256), dict(B=2, heads=8, D=64, n=105), dict(B=2, heads=2, D=32, n=16),
              dict(B=1, heads=8, D=64, n=288)]:
        try:
            e = gc.attention_case(**c)
            say("%-4s attn %s err %.2e" % ("OK" if e < 2e-6 else "BAD", c, e))
        except Exception as ex:
            say("FAIL attn", c, repr(ex)[:300])
    say("== fa sandwich")
    for c in [dict(B=2, heads=8, C=64, H=64, W=64), dict(B=2, heads=8, C=64, H=32, W=32),
              dict(B=1, heads=8, C=64, H=24, W=48), dict(B=1, heads=8, C=64, H=48, W=96),
              dict(B=2, heads=2, C=32, H=16, W=16), dict(B=2, heads=2, C=32, H=8, W=8),
              dict(B=2, heads=8, C=64, H=64, W=64, instnorm=False)]:
        try:
            e = gc.sandwich_case(**c)
            say("%-4s sandwich %s err %.2e" % ("OK" if e < 2e-6 else "BAD", c, e))
        except Exception as ex:
            say("FAIL sandwich", c, repr(ex)[:300])


def run_trace(preset):
    say("== layer trace", preset)
    args = config.preset(preset)
    B = 2
    x = filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 7)
    param = filler.uniform01("param", B, 7).astype(np.float32) if args.family == "twophase_cond" else None
    try:
        rows = gc.layer_trace_compare(args, 1, x, param)
        for st, name, e in rows:
            say("%-4s %-10s %-50s %.2e" % ("OK" if e < 2e-5 else "BAD", st, name, e))
    except Exception:
        say("FAIL trace", preset, traceback.format_exc()[-1500:])




def run_golden(case):
    """rollout error vs the REAL reference's golden fixture (and vs its fp64 run) per stored step"""
    from helpers import load_golden, case_args, case_inputs, rel_l2
    meta, g = load_golden(case)
    args = case_args(meta)
    model, _ = gc.build_models(args, meta["weight_seed"])
    x, param = case_inputs(meta, args)
    xd = torch.from_numpy(x).cuda()
    extra = (torch.from_numpy(param).cuda(),) if param is not None else ()
    dec, lat = model.predict(xd, meta["T"], *extra, to_x=True, return_latents=True)
    dec, lat = dec.cpu().numpy(), lat.cpu().numpy()
    sub = meta["sub"]
    say("== golden", case, "(step: hip-vs-ref32 latent, decoded | hip-vs-ref64 decoded | ref32-vs-ref64 decoded)")
    for i, s in enumerate(meta["steps"]):
        say("   t=%3d  %.2e %.2e | %.2e | %.2e" % (
            s, rel_l2(lat[:, s - 1], g["lat"][:, i]), rel_l2(dec[:, s - 1][..., ::sub, ::sub], g["dec"][:, i]),
            rel_l2(dec[:, s - 1][..., ::sub, ::sub], g["dec_f64"][:, i]), float(g["ref_self_err"][s - 1])))


if __name__ == "__main__":
    what = sys.argv[1:] or ["ops", "trace:ns2d_mini", "trace:ns2d_128"]
    say("# gpu_diag", time.ctime(), torch.cuda.get_device_name(0))
    for w in what:
        if w == "ops":
            run_ops()
        elif w.startswith("trace:"):
            run_trace(w.split(":", 1)[1])
        elif w.startswith("golden:"):
            run_golden(w.split(":", 1)[1])
    say("# done")
