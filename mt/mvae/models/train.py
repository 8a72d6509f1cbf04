"""mt/mvae/models/train.py:34-360."""
from mvae_amd.trainer import Trainer  # noqa: F401
