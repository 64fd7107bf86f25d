"""Shared test helpers: golden loading, synthetic weights, error metrics."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(d["meta"]).decode())
    return meta, d


def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return json.load(f)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def case_args(meta):
    from lns_amd import config
    return config.preset(meta["preset"], **meta["overrides"])


def case_inputs(meta, args):
    from lns_amd import filler
    B = meta["B"]
    x = filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), meta["input_seed"])
    param = None
    if args.family == "twophase_cond":
        param = filler.uniform01("param", B, meta["input_seed"]).astype(np.float32)
    return x, param


def inv_freq(dim):
    from lns_amd import filler
    return filler.inv_freq(dim)


def synthetic_state_dict(shapes, seed, variant=None):
    """{key: ndarray} for a {key: shape} manifest (lns_amd.filler.synthetic_state_dict).  variant: the fixture's
    meta["filler_variant"] ("stable": non-expansive latent chain, `*_stable` fixtures)."""
    from lns_amd import filler
    return filler.synthetic_state_dict(shapes, seed, variant)


def case_variant(meta):
    v = meta.get("filler_variant", "default")
    return None if v == "default" else v


# ---- denormalise + relative-L2 metric fixtures (tests/golden/metrics.npz, tools/make_golden_metrics.py)
METRIC_SEED = 11
METRIC_STATS = {
    "ns2d": dict(shape=(2, 5, 3, 32, 32), mean=0.37, std=1.9),
    "sw": dict(shape=(2, 4, 3, 24, 48), mean=[0.4, -0.2, 9.5], std=[2.1, 1.7, 0.6]),
    "twophase": dict(shape=(2, 4, 4, 31, 61), vel_mean=0.013, vel_std=0.21, prs_mean=310.0, prs_std=180.0),
}


def metric_inputs(name):
    """Deterministic (torch-RNG-independent) synthetic pair: truth y ~ N(0,1), prediction y + 0.05 N(0,1).
    The two-phase VOF channel is spread over about (-0.5, 1.5) so that the clamp acts on both sides; one NS2d plane
    of the truth denormalises to ~0 (the eps clamp of relative_lp_loss)."""
    from lns_amd import filler
    shape = METRIC_STATS[name]["shape"]
    y = filler.normal("metric_y_" + name, shape, METRIC_SEED)
    yh = (y + 0.05 * filler.normal("metric_e_" + name, shape, METRIC_SEED)).astype(np.float32)
    if name == "twophase":
        y[:, :, 3] = 0.5 + 0.35 * y[:, :, 3]
        yh[:, :, 3] = 0.5 + 0.35 * yh[:, :, 3]
    if name == "ns2d":
        y[0, 0, 0] = -METRIC_STATS[name]["mean"] / METRIC_STATS[name]["std"]
    return yh.astype(np.float32), y.astype(np.float32)
