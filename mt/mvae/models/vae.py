"""mt/mvae/models/vae.py:29-166."""
from mvae_amd.models import ModelVAE, Outputs, Reparametrized  # noqa: F401
