#!/usr/bin/env python3
"""FABlock2D with in_proj inside the sandwich kernel ("fa_fused" option, csrc/fa_fused.inc) against the three-kernel path on
one box: decoded fields of the same latents (relative L2), then the decode timed per kernel form with both settings.

    python tools/fa_fused_check.py [preset] [batch] [reps]        (GPU; prints one JSON line per setting)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

FUSED = int(os.environ.get("FA_FUSED_FORM", "2"))      # 2: double-buffered kernel (default), 1: single-buffered


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "ns2d_128"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda:0")
    args, model, _ = bench.build_model(preset, dev)
    from lns_amd import filler
    x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 3)).to(dev)
    eng = model._engine(x)
    z = eng.encode(x)
    outs = {}
    for fused in (0, FUSED, 0, FUSED):
        eng.set_option("fa_fused", fused)
        y = eng.decode(z)
        torch.cuda.synchronize()
        if fused not in outs:
            outs[fused] = y.clone()
        else:
            assert torch.equal(outs[fused], y), "decode is not reproducible with fa_fused=%d" % fused
        eng.timing_enable(True)
        for _ in range(reps):
            eng.decode(z)
        torch.cuda.synchronize()
        t = eng.timing()
        eng.timing_enable(False)
        rec = {k: round(v["ms"] / reps, 4) for k, v in t.items() if "/" in k and ("FABlock" in k or "1x1" in k)}
        total = sum(v["ms"] for k, v in t.items() if "/" not in k) / reps
        print(json.dumps({"fa_fused": fused, "decode_kernel_ms": round(total, 3), "forms_ms": rec}))
    d = (outs[FUSED] - outs[0]).double()
    rel = float(d.norm() / outs[0].double().norm())
    print(json.dumps({"rel_l2_fused_vs_three_kernel": rel, "max_abs": float(d.abs().max()), "finite": bool(torch.isfinite(outs[FUSED]).all())}))
    if not (rel < 2e-6):
        sys.exit(1)


if __name__ == "__main__":
    main()
