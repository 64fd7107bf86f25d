"""Drop-in for reference modules/autoencoder2d.py:160-186 (square, fully periodic or zero-padded AE)."""
from ..dropin import SimpleAutoencoder  # noqa: F401
