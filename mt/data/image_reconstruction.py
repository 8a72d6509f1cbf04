"""mt/data/image_reconstruction.py."""
from mvae_amd.data import CifarVaeDataset, MnistVaeDataset  # noqa: F401
