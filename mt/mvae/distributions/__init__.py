"""mt/mvae/distributions: the distributions on the hot path (WrappedNormal, EuclideanNormal).  The von Mises-Fisher,
Riemannian-normal, hyperspherical-uniform and EuclideanUniform classes of the reference are never wired by its CLI for
the BASELINE configs and are out of scope."""
from .wrapped_normal import WrappedNormal
from .wrapped_distributions import EuclideanNormal

__all__ = ["WrappedNormal", "EuclideanNormal"]
