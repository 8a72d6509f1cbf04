"""SamplingProcedure strategy objects (mt/mvae/sampling/sampling_procedures.py:31-50,91-116,145-155).

The distributions are thin views over the fused component operator (distributions.py): `reparametrize` builds (q, p),
`q.rsample_with_parts` runs the fused kernel once and carries the KL / log-probabilities in its `data`, and `kl_loss`
/ `rsample_log_probs` read them back -- the call sequence the reference uses (vae.py:73-76,137; :95-99) is unchanged.
"""
from typing import Tuple

import torch
from torch import Tensor

from . import functional as Fn
from .distributions import EuclideanNormal, FusedParts, FusedPosterior, FusedPrior, WrappedNormal


class SamplingProcedure:

    def __init__(self, manifold, scalar_parametrization: bool) -> None:
        self._manifold = manifold
        self._scalar_parametrization = scalar_parametrization

    @property
    def scalar_parametrization(self) -> bool:
        return self._scalar_parametrization

    def reparametrize(self, z_mean: Tensor, std: Tensor):
        raise NotImplementedError

    @staticmethod
    def _fused(data):
        parts = data[-1] if isinstance(data, tuple) and len(data) else data
        return parts if isinstance(parts, FusedParts) else None

    def _log_prob(self, q_z, p_z, z: Tensor, data):  # sampling_procedures.py:112-116
        return q_z.log_prob_from_parts(z, data), p_z.log_prob(z)

    def kl_loss(self, q_z, p_z, z: Tensor, data: Tuple) -> Tensor:
        """Single-sample Monte-Carlo KL (sampling_procedures.py:101-104).  With the fused posterior the value was
        computed by the launch that drew the sample; with free-standing distributions it is log q - log p."""
        parts = self._fused(data)
        if parts is not None:
            if parts.kl is None:
                raise ValueError("kl_loss needs the `data` of a single-sample q_z.rsample_with_parts()")
            return parts.kl
        log_q, log_p = self._log_prob(q_z, p_z, z, data)
        return log_q - log_p

    def rsample_log_probs(self, sample_shape: torch.Size, q_z, p_z):
        if isinstance(q_z, FusedPosterior):
            z, data = q_z.rsample_with_parts(sample_shape, want_log_probs=True)
            parts = data[-1]
            return z, parts.log_q, parts.log_p
        z, data = q_z.rsample_with_parts(sample_shape)  # sampling_procedures.py:106-110
        log_q, log_p = self._log_prob(q_z, p_z, z, data)
        return z, log_q, log_p


class WrappedNormalProcedure(SamplingProcedure):

    def reparametrize(self, z_mean: Tensor, std: Tensor):  # sampling_procedures.py:93-99
        q_z = WrappedNormal(z_mean, std, manifold=self._manifold)
        mu_0 = self._manifold.mu_0(z_mean.shape, device=z_mean.device)
        p_z = WrappedNormal(mu_0, torch.ones_like(std), manifold=self._manifold)
        return q_z, p_z


class EuclideanNormalProcedure(SamplingProcedure):

    def reparametrize(self, z_mean: Tensor, std: Tensor):  # sampling_procedures.py:147-151
        return EuclideanNormal(z_mean, std), EuclideanNormal(torch.zeros_like(z_mean), torch.ones_like(std))

    def kl_loss(self, q_z, p_z, z: Tensor, data: Tuple) -> Tensor:  # sampling_procedures.py:153-155
        parts = self._fused(data)
        if parts is not None:
            return super().kl_loss(q_z, p_z, z, data)
        return Fn.normal_kl_standard(q_z.loc, q_z.scale)  # p_z is the standard normal by construction


class EuclideanConstantProcedure(SamplingProcedure):
    """sampling_procedures.py:117-143.  Same constructor signature as the reference's -- including the required `dim`
    that Component.init_layers does not pass, which is why the `c` component cannot be instantiated there either.  The
    uniform-box distributions behind it are not built."""

    def __init__(self, manifold, scalar_parametrization: bool, dim: int, const=None, eps=None) -> None:
        super().__init__(manifold, scalar_parametrization)
        raise NotImplementedError("EuclideanConstantProcedure (an ablation stub of the reference) is not built")


class UniversalSamplingProcedure(SamplingProcedure):
    """sampling_procedures.py:184-206: wrapped normal on the Poincare ball / projected sphere or the Euclidean normal
    procedure, by the sign of the component's curvature.  The fused component operator takes that decision on the
    device (kind MVAE_UNIVERSAL), so this class has nothing left to dispatch; `sampling_procedure` mirrors the
    reference's property for callers that inspect it."""

    def __init__(self, manifold, scalar_parametrization: bool) -> None:
        super().__init__(manifold, scalar_parametrization)
        self._sampling_procedures = {
            -1: WrappedNormalProcedure(manifold._manifolds[-1], scalar_parametrization),
            0: EuclideanNormalProcedure(manifold._manifolds[0], scalar_parametrization),
            1: WrappedNormalProcedure(manifold._manifolds[1], scalar_parametrization),
        }

    @property
    def sampling_procedure(self) -> SamplingProcedure:
        return self._sampling_procedures[self._manifold._choice]

    def reparametrize(self, z_mean: Tensor, std: Tensor):  # sampling_procedures.py:197-198
        return self.sampling_procedure.reparametrize(z_mean, std)

    def kl_loss(self, q_z, p_z, z: Tensor, data: Tuple) -> Tensor:
        if self._fused(data) is not None:
            return super().kl_loss(q_z, p_z, z, data)
        return self.sampling_procedure.kl_loss(q_z, p_z, z, data)
