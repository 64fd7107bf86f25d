"""Deterministic, torch-RNG-independent synthetic weights and fields.

There are no checkpoints or datasets in the build/bench environment, so every
parity test, golden fixture and bench run uses "random-init" weights.  To make
the SAME weights reproducible in three places (the reference import in the
build container, the CPU oracle, and the HIP engine on the GPU box) the values
are a pure function of (state_dict key, shape, seed): a counter-based
splitmix64 hash mapped to a uniform variate.  Scales follow PyTorch's default
initialisers (U(+-1/sqrt(fan_in)) for conv/linear) so a
64..256-step rollout stays bounded.

`zero_module` tensors of the conditional propagator
(reference: modules/cond_utils.py:12-16, used at
train_stage2_twophase_conditional.py:47-58) get NON-zero values here on
purpose: at the reference's zero init the conditioning path has
no effect and would be untested.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for ch in s.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + _GOLD) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(tag: str, n: int, seed: int) -> np.ndarray:
    """n float32 values in [0,1), a pure function of (tag, seed, index)."""
    h0 = np.uint64((_fnv1a64(tag) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * _GOLD + h0
    z = _splitmix64(idx)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def normal(tag: str, shape, seed: int) -> np.ndarray:
    """Approximately N(0,1) float32 field (sum of 4 uniforms, variance-matched)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float32)
    for k in range(4):
        acc += uniform01("%s#%d" % (tag, k), n, seed)
    return ((acc - np.float32(2.0)) * np.float32(np.sqrt(3.0))).reshape(shape).astype(np.float32)


# Filler variants.  "default": PyTorch-default scales everywhere -- the latent chain of such a random-init model amplifies
# a rounding-sized perturbation by x50 (NS2d, 64 steps) to x1e5 (two-phase, 128 steps), so beyond t ~ 48 two correct fp32
# implementations differ by 1e-3 ... 1 and long-horizon parity can only be judged against an ensemble.  "stable": the last
# convolution of every residual branch of the propagator (DilatedResidualBlock conv.5 / ffn.3, train_stage2_ns2d.py:38-48;
# conditional block cond_conv1.2 / ffn.3, train_stage2_twophase_conditional.py:47-58) is scaled by STABLE_GAIN, which makes
# the chain non-expansive: the REAL reference's own fp32 runs then stay within a few 1e-5 of its fp64 run over the whole
# horizon of BASELINE configs 3 / 4 / 5 (measured: tools/stable_filler_probe.py), and the north star's 1e-4 can be gated at
# EVERY step (`*_stable` fixtures).  Same hash, same keys: a variant only rescales tensors.
import re

STABLE_GAIN = 0.25
_RES_OUT = re.compile(r"propagator\.net\.\d+\.(conv\.5|ffn\.3|cond_conv1\.2)\.(weight|bias)$")


def variant_gain(key: str, variant) -> float:
    """Multiplier the variant applies to state_dict entry `key` ("default" / None: 1; "stable"; "gain=<g>": probing)."""
    if variant in (None, "default"):
        return 1.0
    if variant == "stable":
        g = STABLE_GAIN
    elif isinstance(variant, str) and variant.startswith("gain="):
        g = float(variant[5:])
    else:
        raise ValueError("unknown filler variant %r" % (variant,))
    return g if _RES_OUT.search(key) else 1.0


def fill_tensor(key: str, shape, seed: int, variant=None):
    """Synthetic value for state_dict entry `key`; None => keep module default
    (non-learned buffers such as rotary `inv_freq`)."""
    v = _fill_default(key, shape, seed)
    g = variant_gain(key, variant)
    if v is not None and g != 1.0:
        v = (v * np.float32(g)).astype(np.float32)
    return v


def _fill_default(key: str, shape, seed: int):
    shape = tuple(int(s) for s in shape)
    if key.endswith("inv_freq"):
        return None
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(key, n, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "pe":
        v = (u * 2 - 1) * np.float32(0.035)
    elif leaf in ("weights1", "weights2") and len(shape) == 5:
        # spectral weights: scale * U[0,1)  (reference: modules/basics.py:118-124)
        v = u * np.float32(1.0 / (shape[0] * shape[1]))
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        v = (u * 2 - 1) * np.float32(1.0 / np.sqrt(fan_in))
    elif leaf == "weight":          # norm scales
        v = np.float32(1.0) + (u * 2 - 1) * np.float32(0.1)
    else:                           # biases
        v = (u * 2 - 1) * np.float32(0.05)
    return v.reshape(shape).astype(np.float32)


def fill_state_dict(shapes: dict, seed: int, variant=None) -> dict:
    """shapes: {key: shape}.  Returns {key: float32 ndarray} for every key that
    the filler owns (buffers like inv_freq are omitted)."""
    out = {}
    for k, shp in shapes.items():
        v = fill_tensor(k, shp, seed, variant)
        if v is not None:
            out[k] = v
    return out


def inv_freq(dim):
    """RotaryEmbedding buffer (reference: modules/embedding.py:166): a constant, not a weight."""
    return (1.0 / (10000 ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)


def synthetic_state_dict(shapes: dict, seed: int, variant=None) -> dict:
    """{key: ndarray} for a {key: shape} manifest: the deterministic filler for learned tensors, the analytic constant
    for rotary `inv_freq` buffers."""
    sd = fill_state_dict(shapes, seed, variant)
    for k, shp in shapes.items():
        if k.endswith("inv_freq"):
            sd[k] = inv_freq(2 * int(shp[0]))
    return sd


def load_into_torch_module(module, seed: int, variant=None):
    """Overwrite a torch module's parameters/buffers in place with the
    deterministic filler (used on the reference model and on the drop-in)."""
    import torch
    sd = module.state_dict()
    with torch.no_grad():
        for k, t in sd.items():
            v = fill_tensor(k, tuple(t.shape), seed, variant)
            if v is not None:
                t.copy_(torch.from_numpy(v).to(t.dtype))
    return module
