"""Model half of reference train_stage2_twophase.py:56-159 (zero-padded propagator, non-square AE)."""
from .dropin import LatentDynamicsTwoPhase as LatentDynamics, SimpleCNNZeros as SimpleCNN  # noqa: F401
