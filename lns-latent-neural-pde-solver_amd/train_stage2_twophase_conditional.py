"""Model half of reference train_stage2_twophase_conditional.py:78-193 (conditional propagator)."""
from .dropin import LatentDynamicsTwoPhaseConditional as LatentDynamics, SimpleCNNConditional as SimpleCNN  # noqa: F401
