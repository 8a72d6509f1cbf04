"""mt/mvae/distributions/wrapped_distributions.py:39-42."""
from mvae_amd.distributions import EuclideanNormal  # noqa: F401
