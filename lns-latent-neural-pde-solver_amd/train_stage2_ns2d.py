"""Model half of reference train_stage2_ns2d.py:56-158 (`SimpleCNN`, `LatentDynamics`)."""
from .dropin import LatentDynamics, SimpleCNN  # noqa: F401
