"""Import shim for the *reference* (oskopek/mvae) -- used ONLY in the build container.

Fixture-generation infrastructure: this module makes `/root/reference` importable under the
container's torch 2.10 so that `make_golden.py` can record input/output vectors of the
reference's own code.  It never runs on the GPU box (there is no /root/reference there) and
nothing under `mvae_amd/`, `bench.py` or the `-m gpu` tests imports it.

The shim does not touch reference arithmetic:
  1. `sys.modules` stubs for import-time-only dependencies that are absent here:
     geoopt (mt/mvae/ops/poincare.py:18), tensorboardX + torchvision (mt/mvae/stats.py:22-23,
     mt/data/image_reconstruction.py:21), bokeh (mt/visualization/utils.py:18-21).
  2. `Distribution.set_default_validate_args(False)`: torch>=1.8 validates arg_constraints in
     `Distribution.__init__`, before WrappedNormal sets loc/scale (wrapped_normal.py:26-34).
  3. `torch.backends.cudnn.flags` wrapper swallowing the removed `verbose=` kwarg
     (mt/mvae/utils.py:51-59; the call was always a no-op, an un-entered context manager).
  4. `eps` capture: `torch.distributions.normal._standard_normal` is replaced by a recorder that
     draws from a dedicated generator, so the exact N(0,1) draw behind every `rsample` is stored.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


class _Anything:
    """Attribute sink: any attribute access / call returns another sink."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything


def _stub(name):
    if name in sys.modules:
        return sys.modules[name]
    m = _StubModule(name)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(_stub(parent), child, m)
    return m


class _Lambda:

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)


_installed = False
eps_log = []  # every N(0,1) draw made through torch.distributions.Normal.rsample, in call order
_eps_gen = torch.Generator().manual_seed(1000)


def reseed_eps(seed):
    _eps_gen.manual_seed(seed)
    eps_log.clear()


_eps_queue = []  # pre-loaded draws (make_golden feeds the mvae_amd.synthetic recipe through here)


def preload_eps(tensors):
    eps_log.clear()
    _eps_queue.clear()
    _eps_queue.extend(tensors)


def _recording_standard_normal(shape, dtype, device):
    if _eps_queue:
        e = _eps_queue.pop(0)
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        e = e.to(dtype=dtype, device=device)
        eps_log.append(e.clone())
        return e
    e = torch.randn(tuple(shape), generator=_eps_gen, dtype=torch.float64).to(dtype=dtype, device=device)
    eps_log.append(e.clone())
    return e


def install():
    global _installed
    if _installed:
        return
    for name in [
            "geoopt", "geoopt.manifolds", "geoopt.manifolds.poincare", "geoopt.manifolds.poincare.math",
            "tensorboardX", "torchvision", "torchvision.utils", "torchvision.transforms", "torchvision.datasets",
            "bokeh", "bokeh.io", "bokeh.plotting", "bokeh.resources", "bokeh.models", "bokeh.layouts",
            "bokeh.palettes", "bokeh.transform",
    ]:
        _stub(name)
    sys.modules["torchvision.transforms"].Lambda = _Lambda  # subclassed at import time

    torch.distributions.Distribution.set_default_validate_args(False)

    _orig_flags = torch.backends.cudnn.flags

    def _flags(*args, **kwargs):
        kwargs.pop("verbose", None)
        return _orig_flags(*args, **kwargs)

    torch.backends.cudnn.flags = _flags

    import torch.distributions.normal as _normal
    _normal._standard_normal = _recording_standard_normal

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
