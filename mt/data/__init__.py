"""mt/data/__init__.py: `mnist` and `cifar` are built (with synthetic stand-ins of the same shape when the files are
absent); `bdp` and `omniglot` are out of scope and raise like an unknown type."""
from .image_reconstruction import CifarVaeDataset, MnistVaeDataset
from .vae_dataset import VaeDataset
from mvae_amd.data import create_dataset  # noqa: F401

__all__ = ["CifarVaeDataset", "MnistVaeDataset", "VaeDataset", "create_dataset"]
