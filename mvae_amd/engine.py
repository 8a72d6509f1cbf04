"""StepEngine: owns the flat device buffers of one model and drives the fused HIP step (include/mvae_hip.h,
`mvae_step_forward_backward` / `mvae_step_optimizer`).

HBM layout (float32; every segment starts on a 64-float boundary so matrix rows are 16-byte aligned):

    [0, 64)          raw radius parameters, entry i = component i   (components.{i}._nradius / _pradius / _curvature)
    W_heads [NH, H]  fc_mean rows of every component, then fc_logvar rows     (component.py:52-57)
    b_heads [NH]
    W_e0 [H, D], b_e0 [H]            fc_e0        (ffnn_vae.py:36)
    W_d0 [H, Z], b_d0 [H]            fc_d0        (ffnn_vae.py:39)
    W_logits [D, H], b_logits [D]    fc_logits    (ffnn_vae.py:40)

`params`, `grads`, `adam_m`, `adam_v` share this layout, so the optimizer is one streaming pass and a data-parallel
run all-reduces ONE contiguous gradient buffer.  nn.Parameters of the host-side model are views into `params`
(state-dict keys and shapes are the reference's).
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import ModelDesc, check, load, ptr, stream_ptr
from .functional import ComponentLayout


def _up64(n: int) -> int:
    return (n + 63) // 64 * 64


class FlatLayout:
    """Offsets of every named tensor of a FeedForwardVAE inside the flat buffers."""

    def __init__(self, layout: ComponentLayout, in_dim: int, h_dim: int, scalar_parametrization: bool):
        self.comp_layout = layout
        self.in_dim, self.h_dim = in_dim, h_dim
        NH, Z, H, D = layout.heads_dim, layout.z_dim, h_dim, in_dim
        o = _lib.RADII_REGION
        self.off_w_heads = o; o += _up64(NH * H)
        self.off_b_heads = o; o += _up64(NH)
        self.off_w_e0 = o; o += _up64(H * D)
        self.off_b_e0 = o; o += _up64(H)
        self.off_w_d0 = o; o += _up64(H * Z)
        self.off_b_d0 = o; o += _up64(H)
        self.off_w_logits = o; o += _up64(D * H)
        self.off_b_logits = o; o += _up64(D)
        self.n_params = o
        # name -> (offset, shape), reference state-dict names and registration order
        self.entries: List[Tuple[str, int, Tuple[int, ...]]] = []
        radius_name = {"h": "_nradius", "p": "_nradius", "s": "_pradius", "d": "_pradius", "u": "_curvature"}
        for i, (letter, d) in enumerate(layout.comps):
            desc = layout.descs[i]
            pre = f"components.{i}."
            if letter in radius_name:
                self.entries.append((pre + radius_name[letter], i, ()))
            self.entries.append((pre + "fc_mean.weight", self.off_w_heads + desc.mean_col * H, (d, H)))
            self.entries.append((pre + "fc_mean.bias", self.off_b_heads + desc.mean_col, (d,)))
            self.entries.append((pre + "fc_logvar.weight", self.off_w_heads + desc.logvar_col * H,
                                 (desc.logvar_dim, H)))
            self.entries.append((pre + "fc_logvar.bias", self.off_b_heads + desc.logvar_col, (desc.logvar_dim,)))
        self.entries += [("fc_e0.weight", self.off_w_e0, (H, D)), ("fc_e0.bias", self.off_b_e0, (H,)),
                         ("fc_d0.weight", self.off_w_d0, (H, Z)), ("fc_d0.bias", self.off_b_d0, (H,)),
                         ("fc_logits.weight", self.off_w_logits, (D, H)), ("fc_logits.bias", self.off_b_logits, (D,))]

    def views(self, flat: Tensor) -> Dict[str, Tensor]:
        out = {}
        for name, off, shape in self.entries:
            n = 1
            for s in shape:
                n *= s
            out[name] = flat[off:off + n].view(shape)
        return out

    def n_logical_params(self) -> int:
        t = 0
        for _, _, shape in self.entries:
            n = 1
            for s in shape:
                n *= s
            t += n
        return t


class StepEngine:

    def __init__(self, comps: Sequence[Tuple[str, int]], in_dim: int, h_dim: int, device,
                 scalar_parametrization: bool = False, radius_trainable: Optional[Sequence[bool]] = None,
                 lr: float = 1e-3, curvature_lr: float = 1e-4):
        load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MvaeHipError("StepEngine needs a HIP device: the product path has no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.layout = ComponentLayout(comps, scalar_parametrization)
        self.flat = FlatLayout(self.layout, in_dim, h_dim, scalar_parametrization)
        self.in_dim, self.h_dim = in_dim, h_dim
        self.lr, self.curvature_lr = float(lr), float(curvature_lr)
        n = self.layout.n
        tr = [False] * n if radius_trainable is None else [bool(t) for t in radius_trainable]
        self.radius_trainable = [t and letter != "e" for t, (letter, _) in zip(tr, self.layout.comps)]
        P = self.flat.n_params
        z = lambda k, dt=torch.float32: torch.zeros(k, dtype=dt, device=self.device)  # noqa: E731
        self.params, self.grads, self.adam_m, self.adam_v = z(P), z(P), z(P), z(P)
        self.counters = z(32, torch.int32)
        self.stats = z(3 * (4 + n))  # sums | last step | Kahan compensation of the sums
        self._ctx: Dict[int, Tuple[int, Tensor]] = {}
        self._valid_rows: Dict[int, int] = {}  # physical batch -> valid rows (set_valid_rows); re-applied to rebuilt contexts
        self._last_batch: Optional[int] = None
        self.grads_from_engine = False  # the flat gradient buffer was filled by forward_backward(), not through p.grad
        # bumped whenever the contexts (workspace pointers, lr baked into kernel arguments) are rebuilt: anything that
        # captured launches of this engine into a HIP graph must re-capture when it changes
        self.generation = 0
        self._trainable_arr = (C.c_uint8 * n)(*[1 if t else 0 for t in self.radius_trainable])

    def set_radius_trainable(self, radius_trainable: Sequence[bool]) -> None:
        """Parameter.requires_grad toggles on radii / curvatures (the --universal schedule, run.py:153-165)."""
        if len(radius_trainable) != self.layout.n:
            raise ValueError("one flag per component")
        self.radius_trainable = [bool(t) and letter != "e" for t, (letter, _) in zip(radius_trainable, self.layout.comps)]
        for i, t in enumerate(self.radius_trainable):
            self._trainable_arr[i] = 1 if t else 0
        for ctx, _ in self._ctx.values():
            check(load().mvae_set_radius_trainable(ctx, self._trainable_arr))

    # ---- state
    def param_views(self) -> Dict[str, Tensor]:
        return self.flat.views(self.params)

    def param_views_raw(self) -> Dict[str, Tensor]:
        """The fused matrices as the kernels see them (all heads of all components stacked)."""
        f, NH, H = self.flat, self.layout.heads_dim, self.h_dim
        return {"w_heads": self.params[f.off_w_heads:f.off_w_heads + NH * H].view(NH, H),
                "b_heads": self.params[f.off_b_heads:f.off_b_heads + NH]}

    def grad_views(self) -> Dict[str, Tensor]:
        return self.flat.views(self.grads)

    def load_state(self, state: Dict[str, Tensor]) -> None:
        views = self.param_views()
        for name, v in views.items():
            v.copy_(state[name].to(device=self.device, dtype=torch.float32).reshape(v.shape))

    def state_dict(self) -> Dict[str, Tensor]:
        return {k: v.detach().clone() for k, v in self.param_views().items()}

    def set_radii(self, value: float) -> None:
        """Trainer._train_epoch warm-up override (train.py:189-194): every h/p/s/d radius <- value."""
        for i, (letter, _) in enumerate(self.layout.comps):
            if letter in ("h", "p", "s", "d"):  # not the universal curvature, train.py:189-194
                self.params[i] = value

    def set_lr(self, lr: float, curvature_lr: Optional[float] = None) -> None:
        self.lr = float(lr)
        if curvature_lr is not None:
            self.curvature_lr = float(curvature_lr)
        self._drop_contexts()

    def reset_optimizer(self) -> None:
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.counters.zero_()

    def _drop_contexts(self) -> None:
        for h, _ in self._ctx.values():
            load().mvae_destroy(h)
        self._ctx.clear()
        self._last_batch = None
        self.generation += 1

    def __del__(self):
        try:
            self._drop_contexts()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # ---- contexts (one per batch size; they share params / grads / optimizer state / stats)
    def _context(self, batch: int) -> int:
        hit = self._ctx.get(batch)
        if hit is not None:
            return hit[0]
        d = ModelDesc()
        d.abi_version = _lib.ABI_VERSION
        d.arch = 0
        d.batch, d.in_dim, d.h_dim = batch, self.in_dim, self.h_dim
        d.ncomp = self.layout.n
        d.heads_dim, d.z_dim, d.eps_dim = self.layout.heads_dim, self.layout.z_dim, self.layout.eps_dim
        d.n_params = self.flat.n_params
        d.comps = self.layout.descs
        d.off_radii = 0
        for f in ("off_w_heads", "off_b_heads", "off_w_e0", "off_b_e0", "off_w_d0", "off_b_d0", "off_w_logits",
                  "off_b_logits"):
            setattr(d, f, getattr(self.flat, f))
        d.params, d.grads = self.params.data_ptr(), self.grads.data_ptr()
        d.adam_m, d.adam_v = self.adam_m.data_ptr(), self.adam_v.data_ptr()
        d.step_count = self.counters.data_ptr()
        d.stats = self.stats.data_ptr()
        d.radius_trainable = self._trainable_arr
        d.lr, d.curvature_lr = self.lr, self.curvature_lr
        nws = load().mvae_workspace_floats(C.byref(d))
        ws = torch.zeros(int(nws), dtype=torch.float32, device=self.device)
        d.workspace = ws.data_ptr()
        handle = C.c_void_p()
        check(load().mvae_create(C.byref(d), C.byref(handle)))
        self._ctx[batch] = (handle.value, ws)
        if batch in self._valid_rows:
            check(load().mvae_set_valid_rows(handle.value, self._valid_rows[batch]))
        return handle.value

    # ---- padding rows (batch sizes that are not a multiple of 16, e.g. the reference CLI's default 100)
    def set_valid_rows(self, batch: int, valid_rows: int) -> bool:
        """Declare rows [valid_rows, batch) of every (x, eps) this engine is stepped on at physical batch size `batch` to be
        PADDING (mvae_set_valid_rows): no loss, no KL, no gradient, no statistics from them; the in-step input pipeline
        prepares `valid_rows` rows.  The caller keeps the padding rows finite (zeros).  False -- and nothing changed -- when
        this model / shape does not take the four-launch step, the only one that masks."""
        rc = int(load().mvae_set_valid_rows(self._context(int(batch)), int(valid_rows)))
        if rc == _lib.MVAE_E_UNSUPPORTED:
            return False
        check(rc)
        if int(valid_rows) == int(batch):
            self._valid_rows.pop(int(batch), None)
        else:
            self._valid_rows[int(batch)] = int(valid_rows)
        return True

    def padded_rows(self, batch: int) -> int:
        """The physical batch size to allocate (x, eps) buffers with so that a batch of `batch` rows runs on the fused
        kernels: `batch` itself if it is a multiple of 16 or if padding is not available for this model (then the
        one-row-per-workgroup kernels take the exact batch), else the next multiple of 16 with the rest declared padding."""
        batch = int(batch)
        if batch % 16 == 0 or os.environ.get("MVAE_NO_PAD_ROWS", "") not in ("", "0"):
            return batch
        padded = (batch + 15) // 16 * 16
        cur = self._valid_rows.get(padded)
        if padded > 256 or (cur is None and padded in self._ctx) or (cur is not None and cur != batch):
            return batch  # (too large for the four-launch step, or that physical size is in use with another row count)
        return padded if self.set_valid_rows(padded, batch) else batch

    # ---- the step
    def _check_inputs(self, x: Tensor, eps: Tensor) -> int:
        if x.dim() != 2 or x.shape[1] != self.in_dim:
            raise ValueError(f"x must be [B, {self.in_dim}], got {tuple(x.shape)}")
        if tuple(eps.shape) != (x.shape[0], self.layout.eps_dim):
            raise ValueError(f"eps must be [{x.shape[0]}, {self.layout.eps_dim}], got {tuple(eps.shape)}")
        return x.shape[0]

    def forward_backward(self, x: Tensor, eps: Tensor, beta: float = 1.0, want_outputs: bool = False):
        """forward -> ELBO -> backward.  Fills self.grads with d(-ELBO)/d(theta); adds to self.stats."""
        B = self._check_inputs(x, eps)
        ctx = self._context(B)
        self._last_batch = B
        out = None
        lo = cz = bce = kl = None
        if want_outputs:
            lo = x.new_empty(B, self.in_dim)
            cz = x.new_empty(B, self.layout.z_dim)
            bce = x.new_empty(B)
            kl = x.new_empty(self.layout.n, B)
            out = {"logits": lo, "concat_z": cz, "bce": bce, "kl": kl}
        check(load().mvae_step_forward_backward(ctx, ptr(x), ptr(eps), float(beta), 1 if want_outputs else 0, ptr(lo),
                                                ptr(cz), ptr(bce), ptr(kl), stream_ptr(self.device)))
        self.grads_from_engine = True  # CurvatureOptimizer.step uses the flat gradient buffer as it is
        return out

    def optimizer_step(self, do_curvature_step: bool, batch: Optional[int] = None) -> None:
        """The optimizer kernel is independent of the batch size; `batch` only selects which context's component table
        travels with the launch (default: the batch size of the last forward_backward, else any existing context, else
        a context for batch 1 is created)."""
        if batch is None:
            batch = self._last_batch if self._last_batch is not None else (next(iter(self._ctx)) if self._ctx else 1)
        ctx = self._context(batch)
        check(load().mvae_step_optimizer(ctx, 1 if do_curvature_step else 0, stream_ptr(self.device)))
        self.grads_from_engine = False

    def owned_range(self, rank: int, world: int):
        """[lo, hi) in floats of the flat buffers that the sharded optimizer forms give to `rank` of `world`
        (peer_slice4, csrc/mvae_common.hpp: whole float4, at least the radii region)."""
        n = int(self.params.numel())
        s4 = max(16, (n // 4 + world - 1) // world)
        return min(n, 4 * s4 * rank), min(n, 4 * s4 * (rank + 1))

    def optimizer_step_slice(self, rank: int, world: int, do_curvature_step: bool, batch: Optional[int] = None) -> None:
        """optimizer_step on `owned_range(rank, world)` only (radii: rank 0), for a data-parallel exchange that left the
        SUMMED gradient of that range in `grads`; the caller all-gathers `params` afterwards.  Adam's moments move on the
        owned range only."""
        if batch is None:
            batch = self._last_batch if self._last_batch is not None else (next(iter(self._ctx)) if self._ctx else 1)
        check(load().mvae_step_optimizer_slice(self._context(batch), int(rank), int(world), 1 if do_curvature_step else 0,
                                               stream_ptr(self.device)))
        self.grads_from_engine = False

    def train_step(self, x: Tensor, eps: Tensor, beta: float = 1.0, do_curvature_step: bool = False) -> None:
        B = self._check_inputs(x, eps)
        self._last_batch = B
        check(load().mvae_train_step(self._context(B), ptr(x), ptr(eps), float(beta),
                                     1 if do_curvature_step else 0, stream_ptr(self.device)))

    def set_next_batch_feed(self, batch: int, images: Optional[Tensor], perm: Optional[Tensor] = None, seed: int = 0,
                            batches_per_epoch: int = 1, mode: int = 1, x_next: Optional[Tensor] = None,
                            eps_next: Optional[Tensor] = None) -> None:
        """Arms the context of batch size `batch`: its next step also prepares the batch AFTER it into x_next / eps_next on
        spare workgroups of launch 4 (mvae_set_next_batch_feed; images None disarms).  One-shot."""
        import ctypes
        if images is None:
            check(load().mvae_set_next_batch_feed(self._context(batch), None, None, 0, ctypes.c_uint64(0), 1, 0, None, None))
            return
        if images.dtype != torch.uint8 or images.dim() != 2 or images.shape[1] != self.in_dim or not images.is_contiguous():
            raise ValueError(f"images must be contiguous uint8 [N, {self.in_dim}]")
        if tuple(x_next.shape) != (batch, self.in_dim) or tuple(eps_next.shape) != (batch, self.layout.eps_dim):
            raise ValueError("x_next / eps_next do not match the batch")
        check(load().mvae_set_next_batch_feed(self._context(batch), ptr(images), ptr(perm), int(images.shape[0]),
                                              ctypes.c_uint64(int(seed)), int(batches_per_epoch), int(mode), ptr(x_next),
                                              ptr(eps_next)))

    def kernel_path(self, batch: int = None) -> str:
        """"row" | "fused" | "block": which latent kernels the step takes at this batch size (mvae_step_kernel_path)."""
        B = int(batch) if batch is not None else (self._last_batch or 128)
        rc = int(load().mvae_step_kernel_path(self._context(B)))
        if rc < 0:
            check(rc)
        return ("row", "fused", "block")[rc]

    STEP_KERNELS = ("enc_fwd", "latent_fwd", "dec1_fwd", "dec1_bwd", "latent_bwd", "enc_bwd")

    def profile_step(self, x: Tensor, eps: Tensor, beta: float = 1.0, do_curvature_step: bool = False,
                     iters: int = 50) -> Dict[str, float]:
        """Average duration (ms) of each launch of the step, HIP events on the launch stream (mvae_step_profile)."""
        B = self._check_inputs(x, eps)
        ms = (C.c_float * len(self.STEP_KERNELS))()
        check(load().mvae_step_profile(self._context(B), ptr(x), ptr(eps), float(beta), 1 if do_curvature_step else 0,
                                       int(iters), ms, stream_ptr(self.device)))
        return dict(zip(self.STEP_KERNELS, [float(v) for v in ms]))

    # ---- statistics (stats.py:120-127 semantics, read when the host wants them)
    def read_stats(self, reset: bool = False) -> Dict[str, object]:
        s = self.stats.cpu()  # one sync
        n = 4 + self.layout.n
        rec = lambda v: {"bce": float(v[0]), "kl": float(v[1]), "elbo": float(v[2]), "steps": int(v[3]),  # noqa: E731
                         "component_kl": [float(t) for t in v[4:n]]}
        out = {"sum": rec(s[:n]), "last": rec(s[n:2 * n])}
        if reset:
            self.stats.zero_()
        return out
