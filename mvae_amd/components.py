"""Latent components (mt/mvae/components/component.py:30-242): same classes, constructor signatures, parameter names
(`_nradius` for h/p, `_pradius` for s/d, `_curvature` for u -- the optimizer routing, the gradient clip and the radius
warm-up key on these names, train.py:189-194,329-341, vae.py:161) and `dim` convention (ambient: true_dim + 1 for h
and s)."""
from typing import Dict, Optional, Tuple, Type

import torch
from torch import Tensor

from . import functional as Fn
from .distributions import FusedPosterior, FusedPrior
from .ops import (Euclidean, Hyperboloid, Manifold, PoincareBall, Sphere, StereographicallyProjectedSphere,
                  Universal)


class Component(torch.nn.Module):
    LETTER = "?"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type) -> None:
        super().__init__()
        self.dim = dim
        self.fixed_curvature = fixed_curvature
        self._sampling_procedure_type = sampling_procedure
        self.sampling_procedure = None
        self.manifold: Optional[Manifold] = None
        self.fc_mean: Optional[torch.nn.Linear] = None
        self.fc_logvar: Optional[torch.nn.Linear] = None
        self._scalar_parametrization = False
        self._layout = None

    def init_layers(self, in_dim: int, scalar_parametrization: bool) -> None:  # component.py:48-57
        self.manifold = self.create_manifold()
        self.sampling_procedure = self._sampling_procedure_type(self.manifold, scalar_parametrization)
        self._scalar_parametrization = scalar_parametrization
        self.fc_mean = torch.nn.Linear(in_dim, self.mean_dim)
        self.fc_logvar = torch.nn.Linear(in_dim, 1 if scalar_parametrization else self.true_dim)

    @property
    def device(self) -> torch.device:
        return self.fc_mean.weight.device

    def _single_layout(self) -> Fn.ComponentLayout:
        if self._layout is None:
            self._layout = Fn.ComponentLayout([(self.LETTER, self.true_dim)], self._scalar_parametrization)
        return self._layout

    def _radius_param(self) -> Optional[Tensor]:
        return getattr(self, "_nradius", getattr(self, "_pradius", getattr(self, "_curvature", None)))

    def _radii_tensor(self) -> Tensor:
        """The raw radius / curvature parameter as a [1] tensor (keeps its autograd history)."""
        r = self._radius_param()
        if r is None:
            return torch.zeros(1, device=self.device)
        return r.reshape(1)

    def _heads(self, x: Tensor) -> Tensor:
        """fc_mean(x) and fc_logvar(x) as ONE contraction over the stacked weights; differentiable (Fn.linear)."""
        W = torch.cat((self.fc_mean.weight, self.fc_logvar.weight), dim=0)
        b = torch.cat((self.fc_mean.bias, self.fc_logvar.bias), dim=0)
        return Fn.linear(x, W, b)

    def forward(self, x: Tensor):  # component.py:32-35
        q_z = FusedPosterior(self, self._heads(x))
        p_z = FusedPrior(self, x.shape[0], x.device)
        return q_z, p_z, (q_z.loc, q_z.scale)

    def encode(self, x: Tensor) -> Tuple[Tensor, Tensor]:  # component.py:63-75
        q_z = FusedPosterior(self, self._heads(x))
        return q_z.loc, q_z.scale

    def reparametrize(self, z_mean: Tensor, std: Tensor):  # component.py:77-78
        """Free-standing (q_z, p_z) from already-encoded parameters, as the reference's method; Component.forward
        returns the fused posterior instead (same protocol, one launch)."""
        return self.sampling_procedure.reparametrize(z_mean, std)

    def kl_loss(self, q_z, p_z, z: Tensor, data: Tuple) -> Tensor:
        return self.sampling_procedure.kl_loss(q_z, p_z, z, data)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(R^{self.dim})"

    def _shortcut(self) -> str:
        return f"{self.__class__.__name__.lower()[0]}{self.true_dim}"

    def summary_name(self, comp_idx: int) -> str:
        return f"comp_{comp_idx:03d}_{self._shortcut()}"

    def summaries(self, comp_idx: int, q_z, prefix: str = "train") -> Dict[str, Tensor]:
        name = prefix + "/" + self.summary_name(comp_idx)
        return {name + "/mean/norm": torch.norm(q_z.mean, p=2, dim=-1),
                name + "/stddev/norm": torch.norm(q_z.stddev, p=2, dim=-1)}

    def create_manifold(self) -> Manifold:
        raise NotImplementedError

    @property
    def true_dim(self) -> int:
        raise NotImplementedError

    @property
    def mean_dim(self) -> int:
        return self.true_dim


class HyperbolicComponent(Component):
    LETTER = "h"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type, radius: float = 1.0) -> None:
        super().__init__(dim + 1, fixed_curvature, sampling_procedure)
        self._nradius = torch.nn.Parameter(torch.tensor(radius), requires_grad=not fixed_curvature)

    def create_manifold(self) -> Manifold:
        return Hyperboloid(lambda: self._nradius)

    @property
    def true_dim(self) -> int:
        return self.dim - 1


class PoincareComponent(Component):
    LETTER = "p"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type, radius: float = 1.0) -> None:
        super().__init__(dim, fixed_curvature, sampling_procedure)
        self._nradius = torch.nn.Parameter(torch.tensor(radius), requires_grad=not fixed_curvature)

    def create_manifold(self) -> Manifold:
        return PoincareBall(lambda: self._nradius)

    @property
    def true_dim(self) -> int:
        return self.dim


class SphericalComponent(Component):
    LETTER = "s"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type, radius: float = 1.0) -> None:
        super().__init__(dim + 1, fixed_curvature, sampling_procedure)
        self._pradius = torch.nn.Parameter(torch.tensor(radius), requires_grad=not fixed_curvature)

    def create_manifold(self) -> Manifold:
        return Sphere(lambda: self._pradius)

    @property
    def true_dim(self) -> int:
        return self.dim - 1


class StereographicallyProjectedSphereComponent(Component):  # component.py:170-189
    LETTER = "d"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type, radius: float = 1.0) -> None:
        super().__init__(dim, fixed_curvature, sampling_procedure)
        self._pradius = torch.nn.Parameter(torch.tensor(radius), requires_grad=not fixed_curvature)

    def create_manifold(self) -> Manifold:
        return StereographicallyProjectedSphere(lambda: self._pradius)

    @property
    def true_dim(self) -> int:
        return self.dim

    def _shortcut(self) -> str:
        return f"d{self.true_dim}"


class ConstantComponent(Component):  # component.py:206-222
    """Parses and constructs like the reference's, and -- like the reference's -- cannot be wired into a model:
    `init_layers` instantiates EuclideanConstantProcedure(manifold, scalar_parametrization) without the `dim` argument
    its constructor requires (sampling_procedures.py:119-131), a TypeError in both code bases."""
    LETTER = "c"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type, const: Optional[Tensor] = None,
                 eps: Optional[Tensor] = None) -> None:
        super().__init__(dim, fixed_curvature=False, sampling_procedure=sampling_procedure)

    def create_manifold(self) -> Manifold:
        return Euclidean()

    @property
    def true_dim(self) -> int:
        return self.dim


class UniversalComponent(Component):  # component.py:225-242
    LETTER = "u"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type, curvature: float = 0.0,
                 eps: float = 1e-6) -> None:
        super().__init__(dim, fixed_curvature, sampling_procedure)
        self._curvature = torch.nn.Parameter(torch.tensor(curvature), requires_grad=not fixed_curvature)
        self._eps = eps

    def create_manifold(self) -> Manifold:
        return Universal(lambda: self._curvature, eps=self._eps)

    @property
    def true_dim(self) -> int:
        return self.dim


class EuclideanComponent(Component):
    LETTER = "e"

    def __init__(self, dim: int, fixed_curvature: bool, sampling_procedure: Type) -> None:
        super().__init__(dim, fixed_curvature=True, sampling_procedure=sampling_procedure)

    def create_manifold(self) -> Manifold:
        return Euclidean()

    @property
    def true_dim(self) -> int:
        return self.dim
