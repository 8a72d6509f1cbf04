"""Posterior / prior objects handed out by Component.forward, backed by the fused component operator.

Distribution protocol of the reference (wrapped_distributions.py:23-36, wrapped_normal.py:62-107):
`rsample_with_parts(shape) -> (z, data)`, `rsample(shape)`, `.mean/.loc`, `.stddev/.scale`.  `data[-1]` is a
FusedParts record carrying what the same kernel launch already computed (single-sample KL, log q, log p), so the
later `kl_loss(q, p, z, data)` call costs nothing.
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
            eps = torch.zeros(self.heads.shape[0], lay.eps_dim, device=self.heads.device)
            out = Fn.component_forward(lay, self.heads, eps, self.component._radii_tensor(), want_kl=False,
                                       want_params=True)
            d = self.component.true_dim
            lvd = lay.descs[0].logvar_dim
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
        out = Fn.component_forward(lay, self.heads, eps, self.component._radii_tensor(), want_kl=len(shape) == 0,
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
