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
    """RotaryEmbedding buffer (modules/embedding.py:166): a constant, not a weight."""
    return (1.0 / (10000 ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)


def synthetic_state_dict(shapes, seed):
    """{key: ndarray} for a {key: shape} manifest: deterministic filler for learned
    tensors, the analytic constant for rotary inv_freq buffers."""
    from lns_amd import filler
    sd = filler.fill_state_dict(shapes, seed)
    for k, shp in shapes.items():
        if k.endswith("inv_freq"):
            sd[k] = inv_freq(2 * int(shp[0]))
    return sd
