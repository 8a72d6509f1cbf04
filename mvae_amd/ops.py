"""Manifold interface of the reference (mt/mvae/ops/manifold.py:22-75) backed by the HIP primitives.

Same class names, method names, argument meaning and conventions: tensors are [..., A] with coordinates last and
arbitrary leading dims; RadiusManifold takes a callable returning the live radius parameter
(component.py:125-126 `Hyperboloid(lambda: self._nradius)`); Hyperboloid / PoincareBall negate the curvature.
Every method is differentiable: a tensor argument (or the radius parameter) that requires grad routes the call through
functional._Prim, whose backward kernel evaluates the same device template over dual numbers (the reference's custom
derivative rules included).  The fused train step does not come through here.
"""
from typing import Any, Callable, Tuple

import torch
from torch import Tensor

from . import _lib
from . import functional as Fn


class Manifold:
    KIND = -1

    def _r(self):
        return None

    def exp_map_mu0(self, x: Tensor) -> Tensor:
        return Fn.exp_map_mu0(self.KIND, x, self._r())

    def inverse_exp_map_mu0(self, x: Tensor) -> Tensor:
        return Fn.inverse_exp_map_mu0(self.KIND, x, self._r())

    def parallel_transport_mu0(self, x: Tensor, dst: Tensor) -> Tensor:
        return Fn.parallel_transport_mu0(self.KIND, x, dst, self._r())

    def inverse_parallel_transport_mu0(self, x: Tensor, src: Tensor) -> Tensor:
        return Fn.inverse_parallel_transport_mu0(self.KIND, x, src, self._r())

    def sample_projection_mu0(self, x: Tensor, at_point: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
        return Fn.sample_projection_mu0(self.KIND, x, at_point, self._r())

    def inverse_sample_projection_mu0(self, x_proj: Tensor, at_point: Tensor) -> Tuple[Tensor, Tensor]:
        return Fn.inverse_sample_projection_mu0(self.KIND, x_proj, at_point, self._r())

    # general base point (the reference keeps these as module-level functions only, e.g. hyperbolics.py:106-128; the
    # methods are a convenience over the same kernels)
    def exp_map(self, x: Tensor, at_point: Tensor) -> Tensor:
        return Fn.exp_map(self.KIND, x, at_point, self._r())

    def inverse_exp_map(self, x: Tensor, at_point: Tensor) -> Tensor:
        return Fn.inverse_exp_map(self.KIND, x, at_point, self._r())

    def distance(self, x: Tensor, y: Tensor, keepdim: bool = True) -> Tensor:
        """Geodesic distance (include/mvae_hip.h: mvae_geodesic_distance)."""
        return Fn.geodesic_distance(self.KIND, x, y, self._r(), keepdim=keepdim)

    def logdet(self, mu: Tensor, std: Tensor, z: Tensor, data: Tuple[Tensor, ...]) -> Tensor:
        raise NotImplementedError

    def mu_0(self, shape: torch.Size, **kwargs: Any) -> Tensor:
        raise NotImplementedError

    @property
    def radius(self) -> Tensor:
        raise NotImplementedError

    @property
    def curvature(self) -> Tensor:
        raise NotImplementedError


class RadiusManifold(Manifold):

    def __init__(self, radius: Callable[[], Tensor]):
        super().__init__()
        self._radius = radius

    def _r(self):
        return self._radius()

    @property
    def radius(self) -> Tensor:  # manifold.py:73-75 (differentiable, like the reference's property)
        return torch.clamp(torch.relu(self._radius()), min=1e-8, max=1e8)

    @property
    def curvature(self) -> Tensor:  # manifold.py:69-71
        return 1.0 / self.radius.pow(2)

    def mu_0(self, shape: torch.Size, **kwargs: Any) -> Tensor:  # hyperbolics.py:68-69 | spherical.py:70-71
        e0 = torch.zeros(shape, **kwargs)
        e0[..., 0] = 1
        return e0 * self.radius.detach().to(e0.device)

    def logdet(self, mu: Tensor, std: Tensor, z: Tensor, data: Tuple[Tensor, ...]) -> Tensor:
        return Fn.logdet(self.KIND, data[0], None, None, self._r())


class Hyperboloid(RadiusManifold):
    KIND = _lib.HYPERBOLOID

    @property
    def curvature(self) -> Tensor:  # hyperbolics.py:53-55
        return -super().curvature


class Sphere(RadiusManifold):
    KIND = _lib.SPHERE


class PoincareBall(RadiusManifold):
    KIND = _lib.POINCARE

    @property
    def curvature(self) -> Tensor:  # poincare.py:30-32
        return -super().curvature

    def mu_0(self, shape: torch.Size, **kwargs: Any) -> Tensor:  # poincare.py:112-113
        return torch.zeros(shape, **kwargs)

    def logdet(self, mu: Tensor, std: Tensor, z: Tensor, data: Tuple[Tensor, ...]) -> Tensor:  # poincare.py:55-89
        return Fn.logdet(self.KIND, None, mu, z, self._r())


class StereographicallyProjectedSphere(RadiusManifold):
    """spherical_projected.py:31-88."""
    KIND = _lib.PROJ_SPHERE

    def mu_0(self, shape: torch.Size, **kwargs: Any) -> Tensor:  # spherical_projected.py:120-121
        return torch.zeros(shape, **kwargs)

    def logdet(self, mu: Tensor, std: Tensor, z: Tensor, data: Tuple[Tensor, ...]) -> Tensor:
        return Fn.logdet(self.KIND, None, mu, z, self._r())  # spherical_projected.py:56-88


class Euclidean(Manifold):
    KIND = _lib.EUCLIDEAN

    @property
    def radius(self):  # euclidean.py:26-32
        return 0

    @property
    def curvature(self):
        return 0

    def mu_0(self, shape: torch.Size, **kwargs: Any) -> Tensor:
        return torch.zeros(shape, **kwargs)

    def logdet(self, mu: Tensor, std: Tensor, z: Tensor, data: Tuple[Tensor, ...]) -> Tensor:
        return torch.zeros_like(mu)  # euclidean.py:58-59


class Universal(Manifold):
    """universal.py:28-83.  Takes a callable returning the live CURVATURE parameter.  The kernels dispatch on its sign
    on the device (kind MVAE_UNIVERSAL), so no method here synchronises; `manifold` / `_choice` (host-side views of the
    same decision, as in the reference) do."""
    KIND = _lib.UNIVERSAL

    def __init__(self, curvature: Callable[[], Tensor], eps: float = 1e-6) -> None:
        super().__init__()
        self._curvature = curvature
        self._manifolds = {
            -1: PoincareBall(lambda: self._sub_radius_param()),
            0: Euclidean(),
            1: StereographicallyProjectedSphere(lambda: self._sub_radius_param()),
        }
        self.eps = eps

    def _r(self):
        return self._curvature()

    def _sub_radius_param(self) -> Tensor:
        return self.radius

    @property
    def radius(self) -> Tensor:  # universal.py:30-32 (sqrt = the reference's clamped sqrt, common.py:117-119)
        k = self._curvature().detach()
        return torch.relu(1 / torch.sqrt(torch.clamp(k.abs(), min=1e-9)))

    @property
    def curvature(self) -> Tensor:  # universal.py:34-36
        return self._curvature()

    @property
    def _choice(self) -> int:  # universal.py:67-74
        k = float(self._curvature())
        return -1 if k < -self.eps else (1 if k > self.eps else 0)

    @property
    def manifold(self) -> Manifold:  # universal.py:63-65
        return self._manifolds[self._choice]

    def mu_0(self, shape: torch.Size, **kwargs: Any) -> Tensor:  # zeros in all three sub-manifolds
        return torch.zeros(shape, **kwargs)

    def logdet(self, mu: Tensor, std: Tensor, z: Tensor, data: Tuple[Tensor, ...]) -> Tensor:  # universal.py:82-83
        if self._choice == 0:
            return torch.zeros_like(mu)
        return Fn.logdet(self.KIND, None, mu, z, self._r())


def lorentz_to_poincare(x: Tensor, radius: Tensor) -> Tensor:
    """hyperbolics.py:151-152 (used by the embedding export, train.py:271)."""
    return Fn.manifold_aux(_lib.OP_TO_BALL, _lib.HYPERBOLOID, x, None, radius)
