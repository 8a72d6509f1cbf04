"""mt/mvae/ops/__init__.py:15-27 (`ive`, the Bessel function of the von Mises-Fisher path, is out of scope)."""
from .manifold import Manifold
from .poincare import PoincareBall
from .hyperbolics import Hyperboloid
from .euclidean import Euclidean
from .spherical_projected import StereographicallyProjectedSphere
from .spherical import Sphere
from .universal import Universal

__all__ = ["Manifold", "StereographicallyProjectedSphere", "Sphere", "Hyperboloid", "PoincareBall", "Euclidean",
           "Universal"]
