"""Trainer shell of the reference (mt/mvae/models/train.py:34-360) around the fused HIP step: epoch loop, radius
warm-up, beta schedule, early stopping, rolling checkpoints (reference-compatible state-dict files), stdout format."""
import os
import warnings
from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch

from .components import (HyperbolicComponent, PoincareComponent, SphericalComponent,
                         StereographicallyProjectedSphereComponent)
from .models import ModelVAE
from .stats import EpochStats


class CurvatureOptimizer:
    """mt/mvae/utils.py:148-180 + Trainer.build_optimizer (train.py:327-360): Adam(lr) on the network parameters,
    SGD(1e-4) on `_nradius` / `_pradius` when `should_do_curvature_step()`.  The arithmetic runs inside the fused
    step (gradient epilogues) or in `mvae_step_optimizer`; this object carries the hyper-parameters and the gate."""

    def __init__(self, learning_rate: float, curvature_lr: float, should_do_curvature_step) -> None:
        self.learning_rate, self.curvature_lr = float(learning_rate), float(curvature_lr)
        self.curv_condition = should_do_curvature_step
        self._model: Optional[ModelVAE] = None
        self._engine = None

    def bind(self, model: ModelVAE) -> None:
        """Hands the hyper-parameters to the model's engine.  Re-binds when the model OR its engine changed (a second
        `model.to(device)` builds a new engine) and refuses to share one engine between optimizers with different
        learning rates silently: the later bind wins and the engine's contexts / captured graphs are rebuilt
        (StepEngine.set_lr bumps its generation, which the graph caches are keyed on)."""
        eng = model._need_engine()
        if self._model is not model or self._engine is not eng or eng.lr != self.learning_rate or \
                eng.curvature_lr != self.curvature_lr:
            if eng.lr != self.learning_rate or eng.curvature_lr != self.curvature_lr:
                eng.set_lr(self.learning_rate, self.curvature_lr)
            self._model, self._engine = model, eng

    def zero_grad(self) -> None:
        """The fused train_step overwrites the gradients, so it never needs this.  In the reference's eager sequence
        (`optimizer.zero_grad(); loss.backward(); optimizer.step()`, vae.py:150-164) autograd ACCUMULATES into p.grad:
        zero the flat gradient buffer and point every p.grad at its slice of it, so the accumulation lands where the
        optimizer kernel reads."""
        if self._model is None or self._model.engine is None:
            return
        eng = self._model.engine
        eng.grads.zero_()
        eng.grads_from_engine = False
        gviews = eng.grad_views()
        for name, p in self._model.named_parameters():
            if p.requires_grad:
                p.grad = gviews[name]

    def step(self, closure: Optional[Any] = None) -> None:
        """Adam on the network parameters + SGD on the radii iff `curv_condition()` (utils.py:174-180), one kernel over
        the flat buffers.  Gradients that autograd left outside the flat buffer (p.grad re-assigned, or None after a
        `zero_grad(set_to_none=True)`) are gathered first."""
        model = self._model
        eng = model._need_engine()
        if getattr(eng, "grads_from_engine", False):
            # the engine's own forward_backward() filled the flat gradient buffer (p.grad is not involved): use it as it is
            eng.optimizer_step(self.curv_condition())
            return
        gviews, pviews = eng.grad_views(), eng.param_views()
        mviews, vviews = eng.flat.views(eng.adam_m), eng.flat.views(eng.adam_v)
        skipped = []
        for name, p in model.named_parameters():
            v = gviews[name]
            if p.grad is None:
                # torch.optim.Adam / SGD SKIP a parameter without a gradient: neither the parameter nor its moments move.
                # The optimizer here is one kernel over the flat buffers, so such a parameter is put back afterwards.
                v.zero_()
                skipped.append((name, pviews[name].clone(), mviews[name].clone(), vviews[name].clone()))
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad.reshape(v.shape))
                p.grad = v
        eng.optimizer_step(self.curv_condition())
        for name, p0, m0, v0 in skipped:
            pviews[name].copy_(p0)
            mviews[name].copy_(m0)
            vviews[name].copy_(v0)


class Trainer:

    def __init__(self, model: ModelVAE, img_dims=None, chkpt_dir: str = "./chkpt", train_statistics: bool = False,
                 show_embeddings: int = 0, export_embeddings: int = 0, test_every: int = 0) -> None:
        self.model = model
        self.chkpt_dir = chkpt_dir
        self.epoch = 0
        self.global_step = 0
        self.test_every = test_every
        # stats.py:35-52: tensorboard-side options.  The projector / figure output (`show_embeddings`) is outside the
        # hot-path scope; the representation export (`export_embeddings`: every N-th test epoch) is built.
        self.show_embeddings = show_embeddings
        self.export_embeddings = export_embeddings
        self.test_epochs = 0
        os.makedirs(chkpt_dir, exist_ok=True)

    # ---- checkpoints (train.py:57-77): model weights only, `{epoch}.chkpt`
    def _path(self, epoch: int) -> str:
        return os.path.join(self.chkpt_dir, f"{epoch}.chkpt")

    def _load_epoch(self, epoch: int) -> None:
        self.model.load_state_dict(torch.load(self._path(epoch), map_location=self.model.device))

    def _save_epoch(self, epoch: int) -> None:
        torch.save({k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}, self._path(epoch))

    def _delete_epoch(self, epoch: int) -> None:
        if os.path.isfile(self._path(epoch)):
            os.remove(self._path(epoch))

    def _update_checkpoints(self, lookahead: int) -> None:
        if self.epoch - lookahead - 1 >= 0:
            self._delete_epoch(self.epoch - lookahead - 1)
        self._save_epoch(self.epoch)

    @staticmethod
    def _should_stop(results: Dict[int, EpochStats], epoch: int, lookahead: int, max_epoch: int) -> Optional[int]:
        """train.py:79-95."""
        stop = epoch - lookahead
        elbos = np.asarray([float(results[i].elbo) for i in range(stop + 1, epoch + 1)])
        best = int(np.argmax(elbos))
        if elbos[best] < float(results[stop].elbo):
            return stop
        if epoch == max_epoch:
            return stop + 1 + best
        return None

    def get_beta(self, betas: Optional[Sequence[float]]) -> float:
        if betas is None:
            return 1.0
        return float(betas[-1] if self.epoch >= len(betas) else betas[self.epoch])

    def build_optimizer(self, learning_rate: float, fixed_curvature: bool) -> CurvatureOptimizer:
        def condition() -> bool:  # train.py:357-358
            return (not fixed_curvature) and (self.epoch >= 10)
        has_radii = any(c._radius_param() is not None for c in self.model.components)
        if not fixed_curvature and not has_radii:
            warnings.warn("Fixed curvature disabled, but found no curvature parameters. Did you mean to set "
                          "fixed=True, or not?")
        opt = CurvatureOptimizer(learning_rate, 1e-4, condition)
        if self.model.engine is not None:  # the eager sequence calls zero_grad() / step() without a train_step first
            opt.bind(self.model)
        return opt

    # ---- epochs
    def _train_epoch(self, optimizer: CurvatureOptimizer, train_data, beta: float) -> EpochStats:
        print(f"\tTrainEpoch {self.epoch}:\t", end="")
        self.model.train()
        eng = self.model._need_engine()
        if self.epoch < 10:  # train.py:189-194 (applies to fixed-curvature models too)
            if any(isinstance(c, (SphericalComponent, PoincareComponent, HyperbolicComponent,
                                  StereographicallyProjectedSphereComponent)) for c in self.model.components):
                eng.set_radii(11 - self.epoch)
        eng.read_stats(reset=True)
        if not self._device_pipeline_epoch(optimizer, train_data, beta):
            for x_mb, _ in train_data:
                self.model.train_step(optimizer, x_mb, beta=beta)
                self.global_step += 1
        dp = getattr(self.model, "_dp", None)
        world = dp.world if dp is not None else 1
        if world > 1:  # global sums on every rank: the early-stopping decisions below stay identical across ranks
            dp.check_exchange()  # a timed-out peer wait means the ranks have diverged: an error, at the epoch boundary
            eng.stats.copy_(dp.reduce_stats())
        sums = eng.read_stats(reset=True)["sum"]  # the only device sync of the epoch
        # The reference asserts isfinite after nearly every op (e.g. vae.py:158 on the loss), a host sync each.  Here a
        # non-finite value in ANY step poisons the running sums, so one check per epoch reports the same condition
        # with the same exception type (MVAE_CHECK_FINITE=1 moves the check to every step, see ModelVAE.train_step).
        if not all(np.isfinite(v) for v in (sums["bce"], sums["kl"], sums["elbo"])):
            raise AssertionError(f"non-finite training statistics in epoch {self.epoch}: {sums}")
        epoch_stats = EpochStats(sums, length=len(train_data.dataset) * world, beta=beta)  # equal shards per rank
        print(self._epoch_dict(epoch_stats), flush=True)
        return epoch_stats

    def _device_pipeline_epoch(self, optimizer: CurvatureOptimizer, train_data, beta: float) -> bool:
        """Scope row f-2: when the training set is a device-resident uint8 image matrix with dynamic binarisation, the
        epoch runs as HIP-graph replays of [mvae_prepare_batch, fused step] (runner.EpochRunner) -- no per-step host
        work.  A ragged last batch is fed through the ordinary train_step.  Returns False if the loader does not
        qualify (the caller then iterates it batch by batch)."""
        from .conv import ConvEngine
        from .data import DeviceLoader
        from .engine import StepEngine
        from .runner import EpochRunner
        eng = self.model.engine
        if getattr(self.model, "float64_chain", False):  # --doubles True: the autograd step, batch by batch
            return False
        # MNIST on the MLP engine (dynamic binarisation) or CIFAR on the conv engine (pixel / 255, no binarisation)
        if not (isinstance(train_data, DeviceLoader) and train_data.train and train_data.images.dtype == torch.uint8 and
                ((isinstance(eng, StepEngine) and train_data.binarize) or
                 (isinstance(eng, ConvEngine) and not train_data.binarize)) and
                train_data.images.shape[0] >= train_data.batch_size):
            return False
        er = getattr(self, "_epoch_runner", None)
        optimizer.bind(self.model)  # may change the engine's lr (-> new generation) before the runner is looked up
        eng = self.model.engine
        if er is None or er.images is not train_data.images or er.B != train_data.batch_size or er.eng is not eng:
            seed = int(torch.randint(0, 2**31 - 1, (1,)).item()) if train_data._gen is None else \
                int(train_data._gen.initial_seed())
            dp = getattr(self.model, "_dp", None)
            seed += 0 if dp is None else dp.rank  # every rank binarises / draws eps from its own Philox stream
            er = self._epoch_runner = EpochRunner(eng, train_data.images, train_data.batch_size, seed=seed, dp=dp,
                                                  binarize=train_data.binarize)
        self.model._sync_trainable()
        self.global_step += er.run_epoch(beta, optimizer.curv_condition())
        tail = er.N - er.nb * er.B
        if tail:
            idx = er.perm[er.nb * er.B:].long()
            x = train_data.images[idx].to(torch.float32) / 255.0
            if train_data.binarize:
                x = (x > torch.rand(x.shape, device=x.device, generator=train_data._gen)).to(torch.float32)
            self.model.train_step(optimizer, x, beta=beta)
            self.global_step += 1
        return True

    def _epoch_dict(self, epoch_stats: EpochStats) -> Dict[str, float]:
        d = epoch_stats.to_print()
        for i, component in enumerate(self.model.components):
            d[f"{component.summary_name(i)}/curvature"] = float(component.manifold.curvature)
        return d

    def _test_epoch(self, test_data, likelihood_n: int, beta: float) -> EpochStats:
        print(f"\tEpoch {self.epoch}:\t", end="")
        self.model.eval()
        sums = {"bce": 0.0, "kl": 0.0, "elbo": 0.0, "component_kl": [0.0] * len(self.model.components)}
        ll = mi = cn = 0.0
        with torch.no_grad():
            for x_mb, _ in test_data:
                reps, _, x_ = self.model(x_mb)
                st = self.model.compute_batch_stats(x_mb, x_, reps, likelihood_n=likelihood_n, beta=beta)
                f = st.convert_to_float()
                sums["bce"] += f.bce
                sums["kl"] += f.kl
                sums["elbo"] += f.elbo
                sums["component_kl"] = [a + b for a, b in zip(sums["component_kl"], f.component_kl)]
                ll += f.log_likelihood or 0.0
                mi += f.mutual_info or 0.0
                cn += f.cov_norm or 0.0
        epoch_stats = EpochStats(sums, length=len(test_data.dataset), beta=beta, log_likelihood=ll, mutual_info=mi,
                                 cov_norm=cn)
        print(self._epoch_dict(epoch_stats), flush=True)
        if self.export_embeddings > 0 and self.test_epochs % self.export_embeddings == 0:  # train.py:288-291
            self._export_representations(test_data)
        self.model.train()
        self.test_epochs += 1
        return epoch_stats

    def _export_representations(self, data, mode: str = "eval") -> None:
        """train.py:297-325: posterior locations per component, the concatenated sample and the labels of every batch,
        as `<chkpt_dir>/repr/<mode>_<component|total|labels>_<epoch>.pt` (what mt/data/representation_dataset.py reads)."""
        print(f"\tExporting {mode} representations...")
        self.model.eval()
        repr_folder = os.path.join(self.chkpt_dir, "repr")
        os.makedirs(repr_folder, exist_ok=True)

        def _filename(component: str) -> str:
            return os.path.join(repr_folder, f"{mode}_{component}_{self.epoch}.pt")

        per_comp = [[] for _ in self.model.components]
        totals, labels = [], []
        with torch.no_grad():
            for x_mb, y_mb in data:
                reps, concat_z, _ = self.model(x_mb)
                for i, r in enumerate(reps):
                    per_comp[i].append(r.q_z.loc.to("cpu"))
                totals.append(concat_z.to("cpu"))
                labels.append(y_mb.to("cpu"))
        torch.save(torch.cat(totals, dim=0), _filename("total"))
        torch.save(torch.cat(labels, dim=0), _filename("labels"))
        for i, component in enumerate(self.model.components):
            torch.save(torch.cat(per_comp[i], dim=0), _filename(component.summary_name(i)))

    def _try_test_during_train(self, test_results, eval_data, likelihood_n, betas) -> None:
        if self.test_every > 0 and self.epoch % self.test_every == 0:
            test_results[self.epoch - 1] = self._test_epoch(eval_data, likelihood_n, self.get_beta(betas))

    def train_epochs(self, optimizer, train_data, eval_data, betas, epochs: int = 300, likelihood_n: int = 500):
        test_results: Dict[int, EpochStats] = {}
        for _ in range(epochs):
            self._train_epoch(optimizer, train_data, beta=self.get_beta(betas))
            self.epoch += 1
            self._try_test_during_train(test_results, eval_data, likelihood_n, betas)
        test_results[self.epoch - 1] = self._test_epoch(eval_data, likelihood_n, self.get_beta(betas))
        self._save_epoch(self.epoch)
        return test_results

    def train_stopping(self, optimizer, train_data, eval_data, betas, warmup: int = 5, lookahead: int = 2,
                       likelihood_n: int = 500, max_epochs: int = 1000):
        """train.py:105-154."""
        assert warmup >= lookahead
        train_results: Dict[int, EpochStats] = {}
        test_results: Dict[int, EpochStats] = {}
        for _ in range(warmup):
            train_results[self.epoch] = self._train_epoch(optimizer, train_data, beta=self.get_beta(betas))
            self._update_checkpoints(lookahead)
            self.epoch += 1
            self._try_test_during_train(test_results, eval_data, likelihood_n, betas)
        stop_epoch = None
        for _ in range(warmup, max_epochs):
            train_results[self.epoch] = self._train_epoch(optimizer, train_data, beta=self.get_beta(betas))
            stop_epoch = Trainer._should_stop(train_results, self.epoch, lookahead, max_epoch=max_epochs - 1)
            self._update_checkpoints(lookahead)
            if stop_epoch:
                break
            self.epoch += 1
            self._try_test_during_train(test_results, eval_data, likelihood_n, betas)
        if not stop_epoch:
            warnings.warn("Did not stop using early stopping.")
            stop_epoch = self.epoch - 1 if not os.path.isfile(self._path(self.epoch)) else self.epoch
        self._load_epoch(stop_epoch)
        last_epoch = self.epoch
        self.epoch = stop_epoch
        print(f"Stopped at epoch: {stop_epoch}. Deleting epochs [{stop_epoch + 1}, {last_epoch}] and "
              f"[{last_epoch - lookahead},{stop_epoch - 1}].")
        for e in range(stop_epoch + 1, last_epoch + 1):
            self._delete_epoch(e)
        for e in range(last_epoch - lookahead, stop_epoch):
            self._delete_epoch(e)
        test_results[stop_epoch] = self._test_epoch(eval_data, likelihood_n, self.get_beta(betas))
        return test_results
