"""ORACLE (test infrastructure) -- the per-batch ELBO step of oskopek/mvae, CPU PyTorch with autograd.

Functional restatement: a model is a `Spec` (layout) plus a dict of tensors keyed by the reference's state-dict names
(`components.{i}._nradius`, `components.{i}.fc_mean.weight`, `fc_e0.weight`, ...).  `eps` (the N(0,1) draw behind every
rsample) and the binarised `x` are explicit inputs, so results are reproducible across devices.

Follows (all under /root/reference):
  mt/mvae/utils.py:78-140           model-string grammar
  mt/mvae/components/component.py   encode (63-75), per-type radius / curvature parameter names (114-242)
  mt/mvae/sampling/sampling_procedures.py:91-116,145-155   q/p construction, MC-KL / analytic KL
  mt/mvae/distributions/wrapped_normal.py:33-107            wrapped normal sample + log-prob
  mt/mvae/models/{vae,ffnn_vae,conv_vae}.py                 forward, ELBO, train_step, log_likelihood
  mt/mvae/stats.py:144-212          BatchStats reductions
  mt/mvae/models/train.py:189-194,327-360 + mt/mvae/utils.py:148-180   warm-up override, optimizer routing
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import ops

LETTERS = ("h", "u", "s", "d", "p", "c", "e")  # utils.py:30-38
SUPPORTED = ("h", "s", "e", "p", "d", "u")  # SURVEY.md section 8 incl. row f-3; `c` (constant component) is not built


# --------------------------------------------------------------------------------------------- grammar
def parse_component_str(s: str) -> Tuple[int, str, int]:
    """utils.py:78-100: `[mult]<letters><dim>[-suffix]`.  As in the reference, the multiplier is only recognised when a
    non-digit follows it and the space type only when a non-letter follows it (so "h" or "12" raise ValueError)."""
    s = s.split("-")[0]
    i = 0
    while i < len(s) and "0" <= s[i] <= "9":
        i += 1
    mult = s[:i] if i < len(s) else ""
    j = i
    while j < len(s) and "a" <= s[j] <= "z":
        j += 1
    letter = s[len(mult):j] if j < len(s) else ""
    return int(mult or "1"), letter, int(s[j:])


@dataclass
class ComponentSpec:
    letter: str
    true_dim: int

    @property
    def kind(self) -> int:
        return ops.KIND_OF_LETTER[self.letter]

    @property
    def dim(self) -> int:  # ambient dim: +1 for h and s (component.py:121-122,159)
        return self.true_dim + 1 if self.letter in ("h", "s") else self.true_dim

    @property
    def radius_name(self) -> Optional[str]:  # component.py:123,142,160
        return {"h": "_nradius", "p": "_nradius", "s": "_pradius", "d": "_pradius", "u": "_curvature"}.get(self.letter)


def parse_components(arg: str) -> List[ComponentSpec]:
    """utils.py:103-140."""
    arg = arg.lower().strip()
    if not arg:
        return []
    out: List[ComponentSpec] = []
    for token in (t.strip() for t in arg.split(",")):
        mult, letter, dim = parse_component_str(token)
        if mult < 1:
            raise ValueError(f"Space multiplier has to be at least 1, was: '{mult}'.")
        if dim < 1:
            raise ValueError(f"Dimension has to be at least 1, was: '{dim}'.")
        if letter not in LETTERS:
            raise NotImplementedError(f"Unknown latent space type '{letter}'.")
        if letter not in SUPPORTED:
            raise NotImplementedError(f"Latent space type '{letter}' is outside the oracle's scope.")
        out.extend(ComponentSpec(letter, dim) for _ in range(mult))
    return out


@dataclass
class Spec:
    model: str
    in_dim: int = 784
    h_dim: int = 400
    arch: str = "ff"  # "ff" | "conv"
    scalar_parametrization: bool = False
    fixed_curvature: bool = True
    components: List[ComponentSpec] = field(default_factory=list)

    def __post_init__(self):
        if not self.components:
            self.components = parse_components(self.model)

    @property
    def total_z_dim(self) -> int:
        return sum(c.dim for c in self.components)

    @property
    def total_true_dim(self) -> int:
        return sum(c.true_dim for c in self.components)

    def named_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """Parameter names/shapes in the reference's registration order (vae.py:49-57, component.py:48-57,
        ffnn_vae.py:36-40, conv_vae.py:47-55)."""
        out: List[Tuple[str, Tuple[int, ...]]] = []
        for i, c in enumerate(self.components):
            if c.radius_name:
                out.append((f"components.{i}.{c.radius_name}", ()))
            lv = 1 if self.scalar_parametrization else c.true_dim
            out += [(f"components.{i}.fc_mean.weight", (c.true_dim, self.h_dim)),
                    (f"components.{i}.fc_mean.bias", (c.true_dim,)),
                    (f"components.{i}.fc_logvar.weight", (lv, self.h_dim)),
                    (f"components.{i}.fc_logvar.bias", (lv,))]
        if self.arch == "ff":
            out += [("fc_e0.weight", (self.h_dim, self.in_dim)), ("fc_e0.bias", (self.h_dim,)),
                    ("fc_d0.weight", (self.h_dim, self.total_z_dim)), ("fc_d0.bias", (self.h_dim,)),
                    ("fc_logits.weight", (self.in_dim, self.h_dim)), ("fc_logits.bias", (self.in_dim,))]
        else:
            out += [("e0.weight", (64, 3, 4, 4)), ("e0.bias", (64,)), ("e1.weight", (128, 64, 4, 4)),
                    ("e1.bias", (128,)), ("e2.weight", (512, 128, 4, 4)), ("e2.bias", (512,)),
                    ("d0.weight", (2048, self.total_z_dim)), ("d0.bias", (2048,)),
                    ("d1.weight", (128, 256, 4, 4)), ("d1.bias", (256,)), ("d2.weight", (256, 64, 4, 4)),
                    ("d2.bias", (64,)), ("d3.weight", (64, 3, 4, 4)), ("d3.bias", (3,))]
        return out


# --------------------------------------------------------------------------------------------- network
def encode(spec: Spec, P: Dict[str, Tensor], x: Tensor) -> Tensor:
    if spec.arch == "ff":  # ffnn_vae.py:42-50
        return torch.relu(F.linear(x, P["fc_e0.weight"], P["fc_e0.bias"]))
    h = x.view(x.shape[0], 3, 32, 32)  # conv_vae.py:57-66
    for name in ("e0", "e1", "e2"):
        h = torch.relu(F.conv2d(h, P[name + ".weight"], P[name + ".bias"], stride=2, padding=1))
    return h.reshape(x.shape[0], -1)


def decode(spec: Spec, P: Dict[str, Tensor], z: Tensor) -> Tensor:
    bs = z.shape[-2]
    if spec.arch == "ff":  # ffnn_vae.py:52-60
        h = torch.relu(F.linear(z, P["fc_d0.weight"], P["fc_d0.bias"]))
        out = F.linear(h, P["fc_logits.weight"], P["fc_logits.bias"])
    else:  # conv_vae.py:68-79
        h = torch.relu(F.linear(z, P["d0.weight"], P["d0.bias"])).view(-1, 128, 4, 4)
        h = torch.relu(F.conv_transpose2d(h, P["d1.weight"], P["d1.bias"], stride=2, padding=1))
        h = torch.relu(F.conv_transpose2d(h, P["d2.weight"], P["d2.bias"], stride=2, padding=1))
        out = F.conv_transpose2d(h, P["d3.weight"], P["d3.bias"], stride=2, padding=1)
    return out.reshape(-1, bs, spec.in_dim).squeeze(0)


# --------------------------------------------------------------------------------------------- latent component
def _normal_log_prob_sum(v: Tensor, scale: Tensor) -> Tensor:
    """sum_d log N(v; 0, scale) as torch.distributions.Normal.log_prob computes it (wrapped_distributions.py:39-42)."""
    var = scale**2
    return (-(v**2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))).sum(dim=-1)


@dataclass
class ComponentOut:
    z: Tensor
    kl: Optional[Tensor]  # single-sample MC KL (wrapped) or analytic KL (euclidean); None when only log-probs asked
    mu: Tensor  # posterior location on the manifold (q_z.loc)
    std: Tensor
    u: Optional[Tensor] = None
    v: Optional[Tensor] = None
    log_q: Optional[Tensor] = None
    log_p: Optional[Tensor] = None


def component_forward(c: ComponentSpec, mean_raw: Tensor, logvar_raw: Tensor, eps: Tensor,
                      radius_param: Optional[Tensor], want_log_probs: bool = False) -> ComponentOut:
    """One latent component, from the two Linear-head outputs to (z, KL).

    `eps` may carry leading sample dims ([n, B, d], log-likelihood path) while the head outputs are [B, d].
    component.py:63-75 -> sampling_procedures.py:93-99|147-151 -> wrapped_normal.py:70-78 -> kl_loss :101-116|153-155
    """
    std = F.softplus(logvar_raw) + 1e-5  # component.py:72
    letter = c.letter
    if letter == "u":  # universal.py:63-74 + sampling_procedures.py:184-206: dispatch on the sign of the curvature
        choice = ops.u_choice(radius_param)
        letter = {-1: "p", 0: "e", 1: "d"}[choice]
        if choice != 0:  # the sub-manifold sees relu(1/sqrt|K|) through RadiusManifold.radius (universal.py:30-32,57-61)
            radius_param = ops.u_radius(radius_param)
    if letter == "e":
        mu = ops.e_exp_map_mu0(mean_raw)
        z = mu + eps * std  # Normal.rsample (wrapped_distributions.py:25-27)
        out = ComponentOut(z=z, kl=None, mu=mu, std=std)
        if want_log_probs:  # sampling_procedures.py:46-50 with EuclideanNormal.log_prob
            out.log_q = _normal_log_prob_sum(z - mu, std.expand_as(mu))
            out.log_p = _normal_log_prob_sum(z, torch.ones_like(mu))
        else:  # torch.distributions kl_divergence(Normal, Normal(0,1)) summed over dims (:153-155)
            var_ratio = std.pow(2)
            t1 = mu.pow(2)
            out.kl = (0.5 * (var_ratio + t1 - 1 - var_ratio.log())).sum(dim=-1)
        return out

    R = ops.radius_from_param(radius_param)
    if std.shape[-1] == 1 and c.true_dim > 1:  # wrapped_normal.py:46-49
        std = std.repeat(*([1] * (std.dim() - 1)), c.true_dim)
    v = eps * std  # Normal(0, std).rsample
    if letter == "h":
        mu = ops.h_exp_map_mu0(mean_raw, R)
        z, (u, _) = ops.h_sample_projection_mu0(v, mu, R)
        logdet_q = ops.h_logdet(u, R)
        mu0 = ops.h_mu0(mu.shape, R, dtype=mu.dtype)
        u0, v0 = ops.h_inverse_sample_projection_mu0(z, mu0, R)
        logdet_p = ops.h_logdet(u0, R)
    elif letter == "s":
        mu = ops.s_exp_map_mu0(mean_raw, R)
        z, (u, _) = ops.s_sample_projection_mu0(v, mu, R)
        logdet_q = ops.s_logdet(u, R)
        mu0 = ops.h_mu0(mu.shape, R, dtype=mu.dtype)  # spherical.py:70-71: same R*e_0
        u0, v0 = ops.s_inverse_sample_projection_mu0(z, mu0, R)
        logdet_p = ops.s_logdet(u0, R)
    elif letter == "p":
        mu = ops.p_exp_map_mu0(mean_raw, R)
        z, (u, _) = ops.p_sample_projection_mu0(v, mu, R)
        logdet_q = ops.p_logdet(mu, z, R)
        mu0 = torch.zeros_like(mu)
        u0, v0 = ops.p_inverse_sample_projection_mu0(z, mu0, R)
        logdet_p = ops.p_logdet(mu0, z, R)
    elif letter == "d":
        mu = ops.d_exp_map_mu0(mean_raw, R)
        z, (u, _) = ops.d_sample_projection_mu0(v, mu, R)
        logdet_q = ops.d_logdet(mu, z, R)
        mu0 = torch.zeros_like(mu)  # spherical_projected.py:120-121
        u0, v0 = ops.d_inverse_sample_projection_mu0(z, mu0, R)
        logdet_p = ops.d_logdet(mu0, z, R)
    else:
        raise NotImplementedError(c.letter)
    log_q = _normal_log_prob_sum(v, std.expand_as(v)) - logdet_q  # wrapped_normal.py:84-97
    log_p = _normal_log_prob_sum(v0, torch.ones_like(v0)) - logdet_p  # :99-103 on p_z = WN(mu0, 1)
    return ComponentOut(z=z, kl=log_q - log_p, mu=mu, std=std, u=u, v=v, log_q=log_q, log_p=log_p)


def _heads(spec: Spec, P: Dict[str, Tensor], h: Tensor, i: int) -> Tuple[Tensor, Tensor]:
    pre = f"components.{i}."
    return (F.linear(h, P[pre + "fc_mean.weight"], P[pre + "fc_mean.bias"]),
            F.linear(h, P[pre + "fc_logvar.weight"], P[pre + "fc_logvar.bias"]))


def _radius_param(P: Dict[str, Tensor], i: int, c: ComponentSpec) -> Optional[Tensor]:
    return P[f"components.{i}.{c.radius_name}"] if c.radius_name else None


@dataclass
class ForwardOut:
    logits: Tensor
    concat_z: Tensor
    bce: Tensor  # [B]
    kl: Tensor  # [n_comp, B]
    elbo: Tensor  # scalar: sum_batch(-bce - beta * sum_i kl_i)   (stats.py:200-202)
    comps: List[ComponentOut]


def forward(spec: Spec, P: Dict[str, Tensor], x: Tensor, eps: Tensor, beta: float = 1.0) -> ForwardOut:
    """vae.py:69-80 + compute_batch_stats :125-147 + BatchStats (stats.py:144-202). eps: [B, sum true_dim]."""
    h = encode(spec, P, x)
    comps, off = [], 0
    for i, c in enumerate(spec.components):
        m, lv = _heads(spec, P, h, i)
        comps.append(component_forward(c, m, lv, eps[..., off:off + c.true_dim], _radius_param(P, i, c)))
        off += c.true_dim
    concat_z = torch.cat([o.z for o in comps], dim=-1)
    logits = decode(spec, P, concat_z)
    bce = F.binary_cross_entropy_with_logits(logits, x, reduction="none").sum(dim=-1)
    kl = torch.stack([o.kl for o in comps], dim=0)
    elbo = (-bce - beta * kl.sum(dim=0)).sum(dim=0)
    return ForwardOut(logits, concat_z, bce, kl, elbo, comps)


def log_likelihood(spec: Spec, P: Dict[str, Tensor], x: Tensor, eps: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """vae.py:82-123 with eps: [n, B, sum true_dim]. Returns (log p(x) [B], mi [B], cov_norm [])."""
    n = eps.shape[0]
    h = encode(spec, P, x)
    log_p_z = torch.zeros(n, x.shape[0], dtype=x.dtype)
    log_q_z_x = torch.zeros(n, x.shape[0], dtype=x.dtype)
    zs, off = [], 0
    for i, c in enumerate(spec.components):
        m, lv = _heads(spec, P, h, i)
        o = component_forward(c, m, lv, eps[..., off:off + c.true_dim], _radius_param(P, i, c), want_log_probs=True)
        off += c.true_dim
        zs.append(o.z)
        log_p_z = log_p_z + o.log_p
        log_q_z_x = log_q_z_x + o.log_q
    concat_z = torch.cat(zs, dim=-1)
    logits = decode(spec, P, concat_z)
    x_orig = x.repeat((n, 1, 1))
    log_p_x_z = -F.binary_cross_entropy_with_logits(logits, x_orig, reduction="none").sum(dim=-1)
    log_p_x = (log_p_x_z + log_p_z - log_q_z_x).logsumexp(dim=0) - math.log(n)
    mi = (log_q_z_x - log_p_z).logsumexp(dim=0) - math.log(n)
    mean_z = concat_z.mean(dim=1, keepdim=True)
    mean_x = x_orig.mean(dim=1, keepdim=True)
    cov_norm = torch.bmm((x - mean_x).transpose(1, 2), concat_z - mean_z).mean(dim=0).norm()
    return log_p_x, mi, cov_norm


# --------------------------------------------------------------------------------------------- training step
class StepOracle:
    """Holds parameters + optimizer state and advances them the way Trainer._train_epoch / ModelVAE.train_step do."""

    def __init__(self, spec: Spec, state: Dict[str, Tensor], lr: float = 1e-3, dtype=torch.float32):
        self.spec = spec
        self.P: Dict[str, Tensor] = {}
        for name, shape in spec.named_shapes():
            t = state[name].detach().clone().to(dtype)
            assert tuple(t.shape) == tuple(shape), (name, t.shape, shape)
            is_radius = name.endswith("radius") or name.endswith("_curvature")
            t.requires_grad_(not (is_radius and spec.fixed_curvature))  # component.py:123,160,232
            self.P[name] = t
        # train.py:327-360: routing by name substring
        net = [p for n, p in self.P.items() if "radius" not in n and "curvature" not in n]
        neg = [p for n, p in self.P.items() if "nradius" in n or "curvature" in n]
        pos = [p for n, p in self.P.items() if "pradius" in n]
        self.adam = torch.optim.Adam(net, lr=lr)
        self.sgd_neg = torch.optim.SGD(neg, lr=1e-4) if neg else None
        self.sgd_pos = torch.optim.SGD(pos, lr=1e-4) if pos else None

    def begin_epoch(self, epoch: int) -> None:
        """train.py:189-194: for epoch < 10 every h/p/s radius is overwritten with 11 - epoch."""
        if epoch < 10:
            for n, p in self.P.items():
                if n.endswith("radius"):
                    p.data = torch.ones_like(p.data) * (11 - epoch)

    def train_step(self, x: Tensor, eps: Tensor, beta: float, epoch: int) -> ForwardOut:
        """vae.py:149-166 + utils.py:174-180 (curvature step iff not fixed and epoch >= 10, train.py:357-358)."""
        for p in self.P.values():
            p.grad = None
        out = forward(self.spec, self.P, x, eps, beta)
        (-out.elbo).backward()
        c_params = [p for n, p in self.P.items() if "curvature" in n]  # vae.py:161-163 (universal components only)
        if c_params:
            torch.nn.utils.clip_grad_norm_(c_params, max_norm=1.0, norm_type=2)
        self.adam.step()
        if (not self.spec.fixed_curvature) and epoch >= 10:
            if self.sgd_pos is not None:
                self.sgd_pos.step()
            if self.sgd_neg is not None:
                self.sgd_neg.step()
        return out

    def state_dict(self) -> Dict[str, Tensor]:
        return {k: v.detach().clone() for k, v in self.P.items()}


def curvature_of(c: ComponentSpec, radius_param: Optional[Tensor]) -> float:
    """manifold.py:69-71, hyperbolics.py:53-55, spherical.py:53-55, poincare.py:30-32, euclidean.py:30-32."""
    if c.letter == "e":
        return 0.0
    if c.letter == "u":  # universal.py:34-36
        return float(radius_param)
    R = float(ops.radius_from_param(radius_param))
    return (-1.0 if c.letter in ("h", "p") else 1.0) / (R * R)
