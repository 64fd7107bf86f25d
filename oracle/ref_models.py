"""TEST INFRASTRUCTURE ONLY -- build the upstream reference models (CPU, PyTorch)
for a named preset, filled with the deterministic synthetic weights.

Build-container only (needs /root/reference, via ref_shim).  Used to generate
the golden fixtures under tests/golden/ and to pin the C oracle.
"""
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for p in (_HERE, _ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import ref_shim  # noqa: E402
from lns_amd import config as lns_config, filler  # noqa: E402

_FAMILY_SCRIPT = {
    "ns2d": "train_stage2_ns2d",
    "sw_half_periodic": "train_stage2_SW",
    "twophase": "train_stage2_twophase",
    "twophase_cond": "train_stage2_twophase_conditional",
}


def build_reference_dynamics(args, weight_seed: int, dtype=torch.float32, variant=None):
    """Reference `LatentDynamics(args)` (random init replaced by the filler),
    eval mode.  For family "sw_nonsquared" (BASELINE config 3) the reference has
    no shipped script: the SW propagator of train_stage2_SW.py:25-87 is paired
    with modules/autoencoder2d_nonsquared.SimpleAutoencoder exactly as
    train_stage2_twophase.py pairs its propagator with that autoencoder."""
    mods = ref_shim.load_reference()
    fam = args.family
    if fam == "ns2d":
        ref_shim.set_square_padding_mode(args.is_periodic)
    if fam == "sw_nonsquared":
        sw = mods["train_stage2_SW"]
        tp = mods["train_stage2_twophase"]

        class LatentDynamics(tp.LatentDynamics):
            def __init__(self, a):
                torch.nn.Module.__init__(self)
                self.vq_ae = mods["modules.autoencoder2d_nonsquared"].SimpleAutoencoder(a)
                self.latent_resolution = a.latent_resolution
                self.latent_dim = a.latent_dim
                self.propagator = sw.SimpleCNN(latent_dim=a.latent_dim, prop_n_block=a.prop_n_block,
                                               prop_n_embd=a.prop_n_embd, dilation=a.dilation)
        model = LatentDynamics(args)
    else:
        model = mods[_FAMILY_SCRIPT[fam]].LatentDynamics(args)
    filler.load_into_torch_module(model, weight_seed, variant)      # variant: lns_amd.filler ("stable": non-expansive chain)
    model = model.to(dtype).eval()
    return model


def patch_cond_embedding_f64():
    """modules/cond_utils.py:34 casts the conditional embedding to fp32 (`.float()`); an fp64 tie-breaker run of the
    conditional propagator re-casts it to the input's dtype (module attribute patched in memory, no source edit)."""
    mod = ref_shim.load_reference()["train_stage2_twophase_conditional"]
    if not getattr(mod.fourier_embedding, "_lns_f64", False):
        orig = mod.fourier_embedding

        def fe64(t, dim, max_period=10000, _o=orig):
            return _o(t, dim, max_period).to(t.dtype)
        fe64._lns_f64 = True
        mod.fourier_embedding = fe64


def reference_predict(model, x, steps, param=None, to_x=True):
    with torch.no_grad():
        if param is not None:
            return model.predict(x, steps, param, to_x=to_x)
        return model.predict(x, steps, to_x=to_x)
