"""mt/mvae/models/conv_vae.py:28-79."""
from mvae_amd.models import ConvolutionalVAE  # noqa: F401
