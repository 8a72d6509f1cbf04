"""ModelVAE / FeedForwardVAE of the reference (mt/mvae/models/vae.py:29-166, ffnn_vae.py:27-60) on the fused HIP step.

Same constructor signatures, attribute names (`components`, `fc_e0`, `fc_d0`, `fc_logits`, `total_z_dim`, `device`),
state-dict keys and method names.  After `.to(device)` every nn.Parameter is a view into the StepEngine's flat HBM
buffer, so `state_dict()` / `load_state_dict()` / checkpoints keep working while the kernels see one contiguous
parameter / gradient / optimizer-state layout.  float32 only (`--doubles=False`); there is no CPU execution path.
"""
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import functional as Fn
from ._lib import MvaeHipError
from .components import Component
from .distributions import FusedParts, FusedPosterior, FusedPrior
from .engine import StepEngine  # noqa: F401  (ConvEngine has the same surface)
from .stats import BatchStatsFloat

_CHECK_FINITE_EVERY_STEP = os.environ.get("MVAE_CHECK_FINITE", "0") not in ("", "0")


class Reparametrized:  # vae.py:29-35

    def __init__(self, q_z, p_z, z: Tensor, data: Tuple) -> None:
        self.q_z = q_z
        self.p_z = p_z
        self.z = z
        self.data = data


class _SlicedPosterior:
    """q_z of one component inside a fused forward: loc/scale are slices of the tensors the kernel produced."""

    def __init__(self, loc: Tensor, scale: Tensor, manifold):
        self.loc = self.mean = loc
        self.scale = self.stddev = scale
        self.manifold = manifold


class BatchStats:
    """stats.py:144-212 over tensors the fused kernels produced ([B] bce, [ncomp, B] kl)."""

    def __init__(self, bce: Tensor, component_kl: Tensor, beta: float, log_likelihood: Optional[Tensor] = None,
                 mutual_info: Optional[Tensor] = None, cov_norm: Optional[Tensor] = None) -> None:
        self._bce, self._component_kl, self._beta = bce, component_kl, beta
        self._log_likelihood, self._mutual_info, self._cov_norm = log_likelihood, mutual_info, cov_norm

    @property
    def bce(self) -> Tensor:
        return self._bce.sum(dim=0)

    @property
    def component_kl(self) -> List[Tensor]:
        return [k.sum(dim=0) for k in self._component_kl]

    @property
    def kl(self) -> Tensor:
        return self._component_kl.sum(dim=0).sum(dim=-1)

    @property
    def elbo(self) -> Tensor:
        return (-self._bce - self._beta * self._component_kl.sum(dim=0)).sum(dim=0)

    @property
    def beta(self) -> float:
        return self._beta

    @property
    def log_likelihood(self) -> Optional[Tensor]:
        return None if self._log_likelihood is None else self._log_likelihood.sum(dim=0)

    @property
    def mutual_info(self) -> Optional[Tensor]:
        return None if self._mutual_info is None else self._mutual_info.sum(dim=0)

    @property
    def cov_norm(self) -> Optional[Tensor]:
        return None if self._cov_norm is None else self._cov_norm.sum(dim=0)

    def convert_to_float(self) -> "EagerBatchStatsFloat":
        return EagerBatchStatsFloat(self)


class EagerBatchStatsFloat:  # stats.py:115-127 (eval path: syncs, like the reference)

    def __init__(self, bs: BatchStats) -> None:
        self.bce, self.kl, self.elbo = bs.bce.item(), bs.kl.item(), bs.elbo.item()
        self.log_likelihood = None if bs.log_likelihood is None else bs.log_likelihood.item()
        self.mutual_info = None if bs.mutual_info is None else bs.mutual_info.item()
        self.cov_norm = None if bs.cov_norm is None else bs.cov_norm.item()
        self.component_kl = [x.item() for x in bs.component_kl]
        self.beta = bs.beta


Outputs = Tuple[List[Reparametrized], Tensor, Tensor]


class ModelVAE(nn.Module):

    def __init__(self, h_dim: int, components: List[Component], dataset, scalar_parametrization: bool) -> None:
        super().__init__()
        self.device = torch.device("cpu")
        self.components = nn.ModuleList(components)
        self.reconstruction_loss = dataset.reconstruction_loss
        self.total_z_dim = sum(component.dim for component in components)
        for component in components:
            component.init_layers(h_dim, scalar_parametrization=scalar_parametrization)
        self._h_dim = h_dim
        self._scalar_parametrization = scalar_parametrization
        self.engine: Optional[StepEngine] = None
        self._generator: Optional[torch.Generator] = None
        self._dp = None  # distributed.DataParallelStep once enable_data_parallel() was called

    # ---- device placement: build the StepEngine and alias every parameter into its flat buffer
    def to(self, device) -> "ModelVAE":
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.engine is not None and self.engine.device == self.device:
            return self  # already bound: a second .to(same device) must not orphan the optimizer / graph state
        super().to(self.device)
        if self.device.type == "cuda":
            self._bind_engine()
            self._dp = None  # bound to the previous engine (if any): enable_data_parallel() again
        return self

    def enable_data_parallel(self, group=None):
        """Data-parallel training over torch.distributed (new functionality; the reference is single-device): every
        rank feeds its own rows, the flat gradient buffer is all-reduced (SUM) before the replicated optimizer step.
        Rank 0's parameters / optimizer state become everybody's starting point."""
        from .distributed import DataParallelStep
        self._dp = DataParallelStep(self._need_engine(), group=group)
        self._dp.broadcast_state(0)
        return self._dp

    def _comps_desc(self):
        return [(c.LETTER, c.true_dim) for c in self.components]

    def _bind_engine(self, lr: float = 1e-3) -> None:
        raise NotImplementedError

    def _alias_parameters(self, eng: StepEngine) -> None:
        views, gviews = eng.param_views(), eng.grad_views()
        for name, p in self.named_parameters():
            views[name].copy_(p.data)
            p.data = views[name]
            if p.requires_grad:
                p.grad = gviews[name]
        self.engine = eng

    def _need_engine(self) -> StepEngine:
        if self.engine is None:
            raise MvaeHipError("the model is not on a HIP device: call model.to('cuda') (there is no CPU path)")
        return self.engine

    def seed_sampler(self, seed: int) -> None:
        self._generator = torch.Generator(device=self.device).manual_seed(seed)

    def _eps(self, *lead: int) -> Tensor:
        """The N(0, 1) draw behind every rsample of the step.  A float32 Box-Muller draw is EXACTLY zero with probability
        ~2^-25 per pair, and a sphere component whose eps is all zeros is 0 / 0 in the reference's formula (spherical.py:87-88,
        |u| unclamped; float64 draws, the reference's default, never get there): the draw comes from `Fn.randn_nonzero`
        (one launch, both Box-Muller uniforms on the open interval, seeded from the torch generator)."""
        return Fn.randn_nonzero((*lead, self._need_engine().layout.eps_dim), self.device, self._generator)

    # ---- reference API
    def encode(self, x: Tensor) -> Tensor:
        raise NotImplementedError

    def decode(self, concat_z: Tensor) -> Tensor:
        raise NotImplementedError

    def _decode_bce_rows(self, concat_z: Tensor, x: Tensor) -> Tensor:
        """sum_j BCE(decode(z)[..., j], x[..., j]) of log_likelihood (vae.py:98-109)."""
        return Fn.bce_rows(self.decode(concat_z), x)

    def _wrap_outputs(self, out, heads: Optional[Tensor] = None) -> Outputs:
        eng = self._need_engine()
        reps = []
        B = out["concat_z"].shape[0]
        for i, c in enumerate(self.components):
            d = eng.layout.descs[i]
            A = c.dim
            z = out["concat_z"][:, d.z_col:d.z_col + A]
            q = _SlicedPosterior(out["mu"][:, d.z_col:d.z_col + A] if "mu" in out else None,
                                 out["std"][:, d.eps_col:d.eps_col + d.logvar_dim] if "std" in out else None,
                                 c.manifold)
            reps.append(Reparametrized(q, FusedPrior(c, B, self.device), z, (FusedParts(kl=out["kl"][i]),)))
        return reps, out["concat_z"], out["logits"]

    def _wants_grad(self) -> bool:
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _stacked_heads_params(self):
        """The fused head matrix / bias / radii as autograd-visible tensors: concatenations of the nn.Parameters in
        the layout's order (all fc_mean rows, then all fc_logvar rows)."""
        W = torch.cat([c.fc_mean.weight for c in self.components] + [c.fc_logvar.weight for c in self.components], dim=0)
        b = torch.cat([c.fc_mean.bias for c in self.components] + [c.fc_logvar.bias for c in self.components], dim=0)
        radii = torch.cat([c._radii_tensor() for c in self.components], dim=0)
        return W, b, radii

    float64_chain = False  # run.py --doubles True: the components' latent chain in float64 (Fn.float64_chain), autograd step

    def forward(self, x: Tensor, eps: Optional[Tensor] = None) -> Outputs:  # vae.py:69-80
        if self.float64_chain and not Fn._FLOAT64_CHAIN:
            with Fn.float64_chain(True):
                return self._forward(x, eps)
        return self._forward(x, eps)

    def _forward(self, x: Tensor, eps: Optional[Tensor] = None) -> Outputs:
        """With gradients enabled every dense layer / the component operator runs as a torch.autograd.Function over the
        HIP kernels, so the reference's own sequence works: `stats = compute_batch_stats(...); (-stats.elbo).backward();
        optimizer.step()` (vae.py:154-164).  ModelVAE.train_step does all of that as one fused launch sequence instead."""
        eng = self._need_engine()
        x = x.to(self.device, torch.float32)
        eps = self._eps(x.shape[0]) if eps is None else eps
        h = self.encode(x)
        if self._wants_grad():
            # an autograd graph is being built: whatever the flat gradient buffer held from an engine-side
            # forward_backward() is no longer what the next optimizer.step() should apply (it reads p.grad then)
            eng.grads_from_engine = False
            W, b, radii = self._stacked_heads_params()
            heads = Fn.linear(h, W, b)
            z, kl = Fn.component_rsample_kl(eng.layout, heads, radii, eps)
            with torch.no_grad():  # q_z.loc / q_z.scale for summaries: values only
                co = Fn.component_forward(eng.layout, heads.detach(), eps, radii.detach(), want_kl=False,
                                          want_params=True)
            x_ = self.decode(z)
            out = {"concat_z": z, "kl": kl, "mu": co["mu"], "std": co["std"], "logits": x_}
        else:
            P = eng.param_views_raw()
            heads = Fn.linear_forward(h, P["w_heads"], P["b_heads"])
            co = Fn.component_forward(eng.layout, heads, eps, eng.params[:eng.layout.n], want_kl=True,
                                      want_params=True)
            x_ = self.decode(co["z"])
            out = {"concat_z": co["z"], "kl": co["kl"], "mu": co["mu"], "std": co["std"], "logits": x_}
        self._last_forward = (x, out)
        return self._wrap_outputs(out)

    def log_likelihood(self, x: Tensor, n: int = 500, eps: Optional[Tensor] = None):  # vae.py:82-123
        if self.float64_chain and not Fn._FLOAT64_CHAIN:
            with Fn.float64_chain(True):
                return self.log_likelihood(x, n, eps)
        eng = self._need_engine()
        x = x.to(self.device, torch.float32)
        B = x.shape[0]
        eps = self._eps(n, B) if eps is None else eps
        h = self.encode(x)
        P = eng.param_views_raw()
        heads = Fn.linear_forward(h, P["w_heads"], P["b_heads"])
        co = Fn.component_forward(eng.layout, heads, eps, eng.params[:eng.layout.n], want_kl=False, want_log_probs=True)
        concat_z = co["z"]  # [n, B, Z]
        bce = self._decode_bce_rows(concat_z, x)  # [n, B] without materialising x.repeat
        tail = None if Fn._FLOAT64_CHAIN else Fn.loglik_tail(bce, co["log_p"], co["log_q"], concat_z, x)
        if tail is not None:  # logsumexp rows, mutual information and the covariance norm in two launches
            return tail
        log_p_z, log_q_z_x = co["log_p"].sum(dim=0), co["log_q"].sum(dim=0)
        log_p_x, mi = Fn.loglik_reduce(bce, log_p_z, log_q_z_x)
        # cov_norm (vae.py:119-121): mean_n[(x - mean_x)^T (z_n - mean_z_n)] = (x - mean_x)^T mean_n(z_n - mean_z_n)
        zn = concat_z.mean(dim=0)  # mean_n(z_n - mean_b z_n) = mean_n z_n - mean_b mean_n z_n: one pass over the samples
        zc = zn - zn.mean(dim=0, keepdim=True)
        xc = x - x.mean(dim=0, keepdim=True)
        cov, _, _ = Fn.linear_backward(xc, torch.zeros(zc.shape[1], xc.shape[1], device=self.device), zc,
                                       need_dx=False)
        return log_p_x, mi, cov.norm()

    def compute_batch_stats(self, x_mb: Tensor, x_mb_: Tensor, reparametrized: List[Reparametrized], beta: float,
                            likelihood_n: int = 0) -> BatchStats:  # vae.py:125-147
        bce = Fn.bce_with_logits_rows(x_mb_, x_mb.to(self.device, torch.float32))
        kl = torch.stack([c.kl_loss(r.q_z, r.p_z, r.z, r.data) for c, r in zip(self.components, reparametrized)])
        ll = mi = cn = None
        if likelihood_n:
            ll, mi, cn = self.log_likelihood(x_mb, n=likelihood_n)
        return BatchStats(bce, kl, beta, ll, mi, cn)

    def _train_step_float64_chain(self, optimizer, x_mb: Tensor, beta: float, eps: Optional[Tensor]):
        """vae.py:149-166 as the reference writes it -- zero_grad, forward, ELBO, backward, optimizer.step through the autograd
        operators -- with the latent chain of every component in float64 (`--doubles True`: Fn.float64_chain).  The fused
        step's kernels are float32 throughout; this path trades their speed for the reference's default numerics."""
        self._need_engine()
        x = x_mb.to(self.device, torch.float32)
        eps = self._eps(x.shape[0]) if eps is None else eps
        optimizer.bind(self)
        self._sync_trainable()
        optimizer.zero_grad()
        with Fn.float64_chain(True), torch.enable_grad():
            reparametrized, concat_z, x_mb_ = self(x, eps=eps)
            stats = self.compute_batch_stats(x, x_mb_, reparametrized, beta)
            (-stats.elbo).backward()
        optimizer.step()
        # the engine's device-side running sums are what Trainer._train_epoch reads: add this step like the fused step does
        from ._lib import check, load, ptr, stream_ptr
        eng = self.engine
        bce_rows, kl_rows = stats._bce.detach().contiguous(), stats._component_kl.detach().contiguous()
        check(load().mvae_batch_stats(ptr(bce_rows), ptr(kl_rows), ptr(eng.stats), float(beta), x.shape[0], eng.layout.n,
                                      stream_ptr(self.device)))
        return stats.convert_to_float(), (reparametrized, concat_z, x_mb_)

    def train_step(self, optimizer, x_mb: Tensor, beta: float, eps: Optional[Tensor] = None):  # vae.py:149-166
        """zero_grad -> forward -> ELBO -> backward -> optimizer.step as ONE fused launch sequence.  Returns a lazy
        BatchStatsFloat (no device sync until a field is read) and an empty outputs tuple.  With `self.float64_chain` set
        (run.py --doubles True) the step runs through the autograd operators with the latent chain in float64 instead."""
        if self.float64_chain:
            return self._train_step_float64_chain(optimizer, x_mb, beta, eps)
        eng = self._need_engine()
        x = x_mb.to(self.device, torch.float32)
        eps = self._eps(x.shape[0]) if eps is None else eps
        optimizer.bind(self)
        self._sync_trainable()
        if self._dp is None:
            eng.train_step(x, eps, float(beta), optimizer.curv_condition())
        else:  # x / eps are this rank's rows of the global batch: gradients -> all-reduce -> optimizer
            self._dp.train_step(x, eps, float(beta), optimizer.curv_condition())
        stats = BatchStatsFloat(eng, beta)
        if _CHECK_FINITE_EVERY_STEP:  # debug mode: the reference's `assert torch.isfinite(loss).all()` (vae.py:158)
            assert np.isfinite(stats.elbo), "non-finite ELBO"
        return stats, (None, None, None)

    def _sync_trainable(self) -> None:
        """Parameter.requires_grad of the radii / curvatures is the source of truth (the --universal schedule flips it,
        run.py:153-165); the engine learns about a change before the next step."""
        eng = self.engine
        flags = [bool(getattr(c._radius_param(), "requires_grad", False)) and c.LETTER != "e" for c in self.components]
        if flags != list(eng.radius_trainable):
            eng.set_radius_trainable(flags)
            gviews = eng.grad_views()
            for name, p in self.named_parameters():
                if p.requires_grad and p.grad is None and name in gviews:
                    p.grad = gviews[name]


class FeedForwardVAE(ModelVAE):

    def __init__(self, h_dim: int, components: List[Component], dataset, scalar_parametrization: bool) -> None:
        super().__init__(h_dim, components, dataset, scalar_parametrization)
        self.in_dim = dataset.in_dim
        self.fc_e0 = nn.Linear(dataset.in_dim, h_dim)  # ffnn_vae.py:36
        self.fc_d0 = nn.Linear(self.total_z_dim, h_dim)  # :39
        self.fc_logits = nn.Linear(h_dim, dataset.in_dim)  # :40

    def _bind_engine(self, lr: float = 1e-3) -> None:
        trainable = [bool(getattr(c._radius_param(), "requires_grad", False)) for c in self.components]
        eng = StepEngine(self._comps_desc(), self.in_dim, self._h_dim, self.device,
                         scalar_parametrization=self._scalar_parametrization, radius_trainable=trainable, lr=lr)
        self._alias_parameters(eng)

    def encode(self, x: Tensor) -> Tensor:  # ffnn_vae.py:42-50
        assert x.dim() == 2 and x.shape[1] == self.in_dim
        return Fn.linear(x, self.fc_e0.weight, self.fc_e0.bias, relu=True)

    def decode(self, concat_z: Tensor) -> Tensor:  # ffnn_vae.py:52-60
        assert concat_z.dim() >= 2
        h = Fn.linear(concat_z, self.fc_d0.weight, self.fc_d0.bias, relu=True)
        return Fn.linear(h, self.fc_logits.weight, self.fc_logits.bias)

    def _decode_bce_rows(self, concat_z: Tensor, x: Tensor) -> Tensor:
        # the fused launch (hidden layer and logits stay on chip) for the shapes it covers, else the three operators
        out = None
        if not Fn._FLOAT64_CHAIN and concat_z.shape[-1] <= 64:
            out = Fn.decode_bce_rows(concat_z, self.fc_d0.weight, self.fc_d0.bias, self.fc_logits.weight, self.fc_logits.bias, x)
        return super()._decode_bce_rows(concat_z, x) if out is None else out


class ConvolutionalVAE(ModelVAE):
    """conv_vae.py:28-79 (BASELINE config [4], CIFAR shapes): 3 x Conv(k4,s2,p1) encoder, Linear + 3 x ConvTranspose
    decoder, `h_dim` must be 8192 as in the reference.  Runs on mvae_amd.conv.ConvEngine."""

    def __init__(self, h_dim: int, components: List[Component], dataset, scalar_parametrization: bool,
                 img_dims: Tuple[int, int, int] = (3, 32, 32)) -> None:
        if h_dim != 8192 or tuple(img_dims) != (3, 32, 32):
            raise ValueError("'conv' architecture only works with --h_dim=8192 and 3x32x32 images (as in the reference)")
        super().__init__(h_dim, components, dataset, scalar_parametrization)
        self.img_dims = img_dims
        self.img_dims_flat = 3072
        self.in_dim = 3072
        self.e0 = nn.Conv2d(3, 64, kernel_size=4, stride=2, padding=1)
        self.e1 = nn.Conv2d(64, 128, kernel_size=4, stride=2, padding=1)
        self.e2 = nn.Conv2d(128, 512, kernel_size=4, stride=2, padding=1)
        self.d0 = nn.Linear(self.total_z_dim, 2048)
        self.d1 = nn.ConvTranspose2d(128, 256, kernel_size=4, stride=2, padding=1)
        self.d2 = nn.ConvTranspose2d(256, 64, kernel_size=4, stride=2, padding=1)
        self.d3 = nn.ConvTranspose2d(64, 3, kernel_size=4, stride=2, padding=1)

    def _bind_engine(self, lr: float = 1e-3) -> None:
        from .conv import ConvEngine
        trainable = [bool(getattr(c._radius_param(), "requires_grad", False)) for c in self.components]
        eng = ConvEngine(self._comps_desc(), self.device, scalar_parametrization=self._scalar_parametrization,
                         radius_trainable=trainable, lr=lr)
        self._alias_parameters(eng)

    def forward(self, x: Tensor, eps: Optional[Tensor] = None) -> Outputs:
        eng = self._need_engine()
        x = x.to(self.device, torch.float32).contiguous()
        eps = self._eps(x.shape[0]) if eps is None else eps
        c = eng._forward(x, eps)
        out = {"concat_z": c["z"], "kl": c["kl"], "logits": c["logits"]}
        return self._wrap_outputs(out)

    def decode(self, concat_z: Tensor) -> Tensor:
        return self._need_engine().decode(concat_z)

    def log_likelihood(self, x: Tensor, n: int = 500, eps: Optional[Tensor] = None, max_rows: int = 4096):
        """vae.py:82-123 on the conv architecture.  The decoder's patch matrices are 4096 floats per 8x8 position, so
        the n samples are decoded in chunks of at most `max_rows` (sample, image) rows; only the per-row BCE survives
        a chunk (x.repeat((n,1,1)) is never materialised)."""
        eng = self._need_engine()
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        eps = self._eps(n, B) if eps is None else eps
        heads = eng.encode_heads(x)
        co = Fn.component_forward(eng.layout, heads, eps, eng.params[:eng.layout.n], want_kl=False, want_log_probs=True)
        concat_z = co["z"]  # [n, B, Z]
        step = max(1, max_rows // B)
        bce = torch.cat([Fn.bce_rows(eng.decode(concat_z[i:i + step]), x) for i in range(0, n, step)], dim=0)  # [n, B]
        tail = Fn.loglik_tail(bce, co["log_p"], co["log_q"], concat_z, x.view(B, -1))
        if tail is not None:
            return tail
        log_p_x, mi = Fn.loglik_reduce(bce, co["log_p"].sum(dim=0), co["log_q"].sum(dim=0))
        zn = concat_z.mean(dim=0)  # mean_n(z_n - mean_b z_n) = mean_n z_n - mean_b mean_n z_n: one pass over the samples
        zc = zn - zn.mean(dim=0, keepdim=True)
        xc = x - x.mean(dim=0, keepdim=True)
        cov, _, _ = Fn.linear_backward(xc, torch.zeros(zc.shape[1], xc.shape[1], device=self.device), zc,
                                       need_dx=False)
        return log_p_x, mi, cov.norm()
