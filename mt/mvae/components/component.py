"""mt/mvae/components/component.py:30-242."""
from mvae_amd.components import (Component, ConstantComponent, EuclideanComponent, HyperbolicComponent,  # noqa: F401
                                 PoincareComponent, SphericalComponent, StereographicallyProjectedSphereComponent,
                                 UniversalComponent)
