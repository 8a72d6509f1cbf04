"""mt/data/vae_dataset.py."""
from mvae_amd.data import VaeDataset  # noqa: F401
