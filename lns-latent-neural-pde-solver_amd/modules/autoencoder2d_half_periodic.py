"""Drop-in for reference modules/autoencoder2d_half_periodic.py:233-259 (circular in one axis, zeros in the other)."""
from ..dropin import SimpleAutoencoderHalfPeriodic as SimpleAutoencoder  # noqa: F401
