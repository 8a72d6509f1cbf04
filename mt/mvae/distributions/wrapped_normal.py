"""mt/mvae/distributions/wrapped_normal.py:26-107."""
from mvae_amd.distributions import WrappedNormal  # noqa: F401
