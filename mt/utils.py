"""mt/utils.py:19-26."""
from mvae_amd.run import str2bool  # noqa: F401
