"""mt/mvae/models/__init__.py."""
from .train import Trainer
from .ffnn_vae import FeedForwardVAE
from .conv_vae import ConvolutionalVAE

__all__ = ["ConvolutionalVAE", "FeedForwardVAE", "Trainer"]
