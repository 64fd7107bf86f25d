"""Model half of reference train_stage2_SW.py:56-159 (half-periodic propagator + AE)."""
from .dropin import LatentDynamicsSW as LatentDynamics, SimpleCNNHalfPeriodic as SimpleCNN  # noqa: F401
