"""mt/mvae/models/ffnn_vae.py:27-60."""
from mvae_amd.models import FeedForwardVAE  # noqa: F401
