"""SamplingProcedure strategy objects (mt/mvae/sampling/sampling_procedures.py:31-50,91-116,145-155).

The distributions are thin views over the fused component operator (distributions.py): `reparametrize` builds (q, p),
`q.rsample_with_parts` runs the fused kernel once and carries the KL / log-probabilities in its `data`, and `kl_loss`
/ `rsample_log_probs` read them back -- the call sequence the reference uses (vae.py:73-76,137; :95-99) is unchanged.
"""
from typing import Tuple

import torch
from torch import Tensor

from .distributions import FusedParts, FusedPosterior, FusedPrior


class SamplingProcedure:

    def __init__(self, manifold, scalar_parametrization: bool) -> None:
        self._manifold = manifold
        self._scalar_parametrization = scalar_parametrization

    @property
    def scalar_parametrization(self) -> bool:
        return self._scalar_parametrization

    def reparametrize(self, z_mean: Tensor, std: Tensor):
        raise NotImplementedError("built by Component.reparametrize from the fused posterior")

    def kl_loss(self, q_z: FusedPosterior, p_z: FusedPrior, z: Tensor, data: Tuple) -> Tensor:
        parts = data[-1] if isinstance(data, tuple) else data
        if not isinstance(parts, FusedParts) or parts.kl is None:
            raise ValueError("kl_loss needs the `data` returned by q_z.rsample_with_parts()")
        return parts.kl

    def rsample_log_probs(self, sample_shape: torch.Size, q_z: FusedPosterior, p_z: FusedPrior):
        z, data = q_z.rsample_with_parts(sample_shape, want_log_probs=True)
        parts = data[-1]
        return z, parts.log_q, parts.log_p


class WrappedNormalProcedure(SamplingProcedure):
    pass


class EuclideanNormalProcedure(SamplingProcedure):
    pass


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
