"""mt/mvae/components/__init__.py."""
from .component import (Component, ConstantComponent, EuclideanComponent, HyperbolicComponent, PoincareComponent,
                        SphericalComponent, StereographicallyProjectedSphereComponent, UniversalComponent)

__all__ = ["Component", "EuclideanComponent", "SphericalComponent", "HyperbolicComponent", "ConstantComponent",
           "PoincareComponent", "UniversalComponent", "StereographicallyProjectedSphereComponent"]
