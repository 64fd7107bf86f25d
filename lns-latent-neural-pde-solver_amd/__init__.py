"""MI355X-native LNS rollout engine: encode -> latent-propagate^N -> decode.

Python host side of the hot path of BaratiLab/LNS-Latent-Neural-PDE-Solver
(reference `LatentDynamics.predict`, train_stage2_ns2d.py:143-158).  The
classes keep the reference's module API and state_dict keys; all arithmetic
runs in hand-written HIP kernels (csrc/) reached through the C ABI declared in
include/lns.h.  Import as `lns_amd` (see ../lns_amd/__init__.py).
"""
__version__ = "0.1.0"
