"""Distributions of the latent components.

Two families with the reference's distribution protocol (wrapped_distributions.py:23-36, wrapped_normal.py:62-107:
`rsample_with_parts(shape) -> (z, data)`, `log_prob_from_parts(z, data)`, `log_prob(z)`, `rsample_log_prob(shape)`,
`.mean/.loc`, `.stddev/.scale`):

* `FusedPosterior` / `FusedPrior`: what Component.forward hands out.  One launch of the fused component operator
  computes the sample AND the single-sample KL / log-probabilities; `data[-1]` is a FusedParts record carrying them, so
  the later `kl_loss(q, p, z, data)` call costs nothing.
* `WrappedNormal(loc, scale, manifold)` / `EuclideanNormal(loc, scale)`: the reference's free-standing classes with its
  constructor signatures (wrapped_normal.py:26-60, wrapped_distributions.py:39-42), every method one HIP primitive
  (manifold sample projection / logdet, diagonal-normal log-density) and differentiable.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import functional as Fn


@dataclass
class FusedParts:
    kl: Optional[Tensor] = None
    log_q: Optional[Tensor] = None
    log_p: Optional[Tensor] = None
    eps: Optional[Tensor] = None


class FusedPosterior:
    """q(z|x) of ONE component: WrappedNormal(mu, std) on h/s/p, Normal(mu, std) on e."""

    def __init__(self, component, heads: Tensor, generator: Optional[torch.Generator] = None):
        self.component = component
        self.heads = heads  # [B, 2d] or [B, d+1]: fc_mean output then fc_logvar output
        self.manifold = component.manifold
        self._generator = generator
        self._params = None

    def _layout(self):
        return self.component._single_layout()

    def _loc_scale(self):
        if self._params is None:
            lay = self._layout()
            d = self.component.true_dim
            lvd = lay.descs[0].logvar_dim
            radii = self.component._radii_tensor()
            if torch.is_grad_enabled() and (self.heads.requires_grad or radii.requires_grad):
                # differentiable: Component.encode as the reference writes it (component.py:63-75), one primitive each
                loc = Fn.exp_map_mu0(lay.descs[0].kind, self.heads[:, :d], radii)
                scale = Fn.guarded("std", self.heads[:, d:d + lvd].contiguous())
                self._params = (loc, scale)
            else:
                eps = torch.zeros(self.heads.shape[0], lay.eps_dim, device=self.heads.device)
                out = Fn.component_forward(lay, self.heads.detach(), eps, radii.detach(), want_kl=False,
                                           want_params=True)
                self._params = (out["mu"], out["std"][:, :lvd] if lvd == 1 else out["std"][:, :d])
        return self._params

    @property
    def loc(self) -> Tensor:
        return self._loc_scale()[0]

    mean = loc

    @property
    def scale(self) -> Tensor:
        return self._loc_scale()[1]

    stddev = scale

    def rsample_with_parts(self, shape: torch.Size = torch.Size(), want_log_probs: bool = False,
                           eps: Optional[Tensor] = None) -> Tuple[Tensor, Tuple]:
        lay = self._layout()
        B = self.heads.shape[0]
        if eps is None:
            eps = torch.randn(tuple(shape) + (B, lay.eps_dim), device=self.heads.device, generator=self._generator)
            eps.masked_fill_(eps == 0, 1e-10)  # (an all-zero eps is 0 / 0 on the sphere: see ModelVAE._eps)
        radii = self.component._radii_tensor()
        if len(shape) == 0 and not want_log_probs and torch.is_grad_enabled() and \
                (self.heads.requires_grad or radii.requires_grad):
            z, kl = Fn.component_rsample_kl(lay, self.heads, radii, eps)  # differentiable (training path)
            return z, (FusedParts(kl=kl[0], eps=eps),)
        out = Fn.component_forward(lay, self.heads.detach(), eps, radii.detach(), want_kl=len(shape) == 0,
                                   want_log_probs=want_log_probs)
        pick = lambda t: None if t is None else t[0]  # noqa: E731
        return out["z"], (FusedParts(kl=pick(out["kl"]), log_q=pick(out["log_q"]), log_p=pick(out["log_p"]), eps=eps),)

    def rsample(self, shape: torch.Size = torch.Size()) -> Tensor:
        return self.rsample_with_parts(shape)[0]

    def rsample_log_prob(self, shape: torch.Size = torch.Size()):
        z, data = self.rsample_with_parts(shape, want_log_probs=True)
        return z, data[-1].log_q


class FusedPrior:
    """p(z) of one component: WrappedNormal(mu_0, 1) / Normal(0, 1) (sampling_procedures.py:96-98,149-150)."""

    def __init__(self, component, batch: int, device):
        self.component = component
        self.manifold = component.manifold
        self._batch, self._device = batch, device

    @property
    def loc(self) -> Tensor:
        return self.manifold.mu_0((self._batch, self.component.dim), device=self._device)

    mean = loc

    @property
    def scale(self) -> Tensor:
        return torch.ones(self._batch, self.component.true_dim, device=self._device)

    stddev = scale


class EuclideanNormal:
    """wrapped_distributions.py:39-42: a diagonal Normal whose log_prob is summed over the last dim."""

    def __init__(self, loc: Tensor, scale: Tensor) -> None:
        self.loc, self.scale = torch.broadcast_tensors(loc, scale)

    @property
    def mean(self) -> Tensor:
        return self.loc

    @property
    def stddev(self) -> Tensor:
        return self.scale

    def rsample(self, sample_shape: torch.Size = torch.Size(), eps: Optional[Tensor] = None) -> Tensor:
        if eps is None:
            eps = torch.randn(tuple(sample_shape) + tuple(self.loc.shape), device=self.loc.device)
        return Fn.normal_rsample(eps, self.loc, self.scale)

    def rsample_with_parts(self, shape: torch.Size = torch.Size(), eps: Optional[Tensor] = None):
        return self.rsample(shape, eps=eps), None

    def log_prob(self, value: Tensor) -> Tensor:
        return Fn.normal_log_prob(value, self.loc, self.scale)

    def log_prob_from_parts(self, z: Tensor, data) -> Tensor:
        return self.log_prob(z)

    def rsample_log_prob(self, shape: torch.Size = torch.Size()):
        z, data = self.rsample_with_parts(shape)
        return z, self.log_prob_from_parts(z, data)


class WrappedNormal:
    """wrapped_normal.py:26-107: v ~ N(0, diag(scale^2)) in the tangent space at mu_0, parallel-transported to `loc` and
    pushed through the exponential map; density = normal density minus the log-determinant of that projection."""

    def __init__(self, loc: Tensor, scale: Tensor, manifold) -> None:
        from .ops import PoincareBall, StereographicallyProjectedSphere, Universal
        self.dim = loc.shape[-1]
        projected = isinstance(manifold, (PoincareBall, StereographicallyProjectedSphere, Universal))
        tangent_dim = self.dim if projected else self.dim - 1
        if scale.shape[-1] > 1 and scale.shape[-1] != tangent_dim:
            raise ValueError("Invalid scale dimension: neither isotropic nor elliptical.")
        if scale.shape[-1] == 1:  # wrapped_normal.py:46-49
            rep = [1] * scale.dim()
            rep[-1] = tangent_dim
            scale = scale.repeat(rep)
        assert loc.shape[:-1] == scale.shape[:-1] and tangent_dim == scale.shape[-1]
        self.loc, self.scale, self.manifold = loc, scale, manifold
        self.device = loc.device
        self.normal = EuclideanNormal(torch.zeros_like(scale), scale)

    @property
    def mean(self) -> Tensor:
        return self.loc

    @property
    def stddev(self) -> Tensor:
        return self.scale

    def rsample_with_parts(self, shape: torch.Size = torch.Size(), eps: Optional[Tensor] = None):
        v_tilde = self.normal.rsample(shape, eps=eps)
        return self.manifold.sample_projection_mu0(v_tilde, at_point=self.loc)

    def rsample(self, sample_shape: torch.Size = torch.Size()) -> Tensor:
        return self.rsample_with_parts(sample_shape)[0]

    def log_prob_from_parts(self, z: Tensor, data) -> Tensor:
        if data is None:
            raise ValueError("Additional data cannot be empty for WrappedNormal.")
        n_logprob = self.normal.log_prob(data[1])
        logdet = self.manifold.logdet(self.loc, self.scale, z, (*data, n_logprob))
        return n_logprob - logdet

    def log_prob(self, z: Tensor) -> Tensor:
        data = self.manifold.inverse_sample_projection_mu0(z, at_point=self.loc)
        return self.log_prob_from_parts(z, data)

    def rsample_log_prob(self, shape: torch.Size = torch.Size()):
        z, data = self.rsample_with_parts(shape)
        return z, self.log_prob_from_parts(z, data)
