"""Drop-in for reference modules/autoencoder2d_nonsquared.py: `SimpleAutoencoder` (:250-276, H != W autoencoder) and
`ConditionalSimpleAutoencoder` (:279-305, CondEncoder :71-145)."""
from ..dropin import ConditionalSimpleAutoencoder  # noqa: F401
from ..dropin import SimpleAutoencoderNonSquared as SimpleAutoencoder  # noqa: F401
