"""mt/mvae/ops/manifold.py:22-75."""
from mvae_amd.ops import Manifold, RadiusManifold  # noqa: F401
