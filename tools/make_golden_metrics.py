#!/usr/bin/env python3
"""Golden vectors for the step right after the path: the datasets' denormalize() followed by
relative_lp_loss, computed by the REAL reference (imported on CPU through oracle/ref_shim.py; runs only in the
build container).  The dataset classes are not constructed (their data files do not exist here): denormalize() only
reads `self.stats` / `self.normstat`, so it is called on a bare namespace carrying those.

Committed: tests/golden/metrics.npz = seeds, statistics and the reference's frame-/sequence-wise errors (fp32 and
fp64) for the NS2d (scalar stats), SW (per-channel stats) and two-phase (wall zeroing + VOF clamp) datasets.

    python tools/make_golden_metrics.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_shim  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import METRIC_SEED as SEED, METRIC_STATS as STATS, metric_inputs as inputs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "metrics.npz")


def main():
    ref_shim.load_reference()
    tu = importlib.import_module("training_utils")
    ns_ds = importlib.import_module("dataset.ns2d_fno_stage2_simpleae")
    sw_ds = importlib.import_module("dataset.Stage2_SW")
    tp_ds = importlib.import_module("dataset.twophase_flow_stage2")
    out = {"seed": np.int64(SEED)}
    for name, st in STATS.items():
        yh, y = inputs(name)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            a, g = torch.from_numpy(yh).to(dt), torch.from_numpy(y).to(dt)
            if name == "ns2d":
                self = types.SimpleNamespace(stats={"mean": torch.tensor(st["mean"], dtype=dt),
                                                    "std": torch.tensor(st["std"], dtype=dt)})
                den = lambda v: ns_ds.NS2DData.denormalize(self, v)
            elif name == "sw":
                ns = {k: {"mean": torch.tensor(m, dtype=dt), "std": torch.tensor(s, dtype=dt)}
                      for k, m, s in zip(("u", "v", "pres"), st["mean"], st["std"])}
                self = types.SimpleNamespace(normstat=ns)
                den = lambda v: sw_ds.SW2DData.denormalize(self, v.clone())
            else:
                stats = {k: torch.tensor(st[k], dtype=dt) for k in ("vel_mean", "vel_std", "prs_mean", "prs_std")}
                self = types.SimpleNamespace(stats=stats)
                den = lambda v: tp_ds.ConditionalTankSloshingData.denormalize(self, v)
            ad, gd = den(a), den(g)
            frame = tu.relative_lp_loss(ad, gd, reduce_dim=(3, 4), p=2, reduce_all=False)
            seq = tu.relative_lp_loss(ad, gd, reduce_dim=(1, 3, 4), p=2, reduce_all=False)
            out["%s_frame_%s" % (name, tag)] = frame.numpy()
            out["%s_seq_%s" % (name, tag)] = seq.numpy()
        print(name, "frame[0,0]", out[name + "_frame_f64"][0, 0], "seq[0]", out[name + "_seq_f64"][0])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
