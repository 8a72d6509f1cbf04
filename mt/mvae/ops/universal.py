"""mt/mvae/ops/universal.py:28-83."""
from mvae_amd.ops import Universal  # noqa: F401
