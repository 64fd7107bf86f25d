"""Drop-in for reference modules/autoencoder2d_nonsquared.py:250-276 (H != W autoencoder)."""
from ..dropin import SimpleAutoencoderNonSquared as SimpleAutoencoder  # noqa: F401
