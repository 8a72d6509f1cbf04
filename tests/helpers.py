"""Shared test helpers: golden loading and the parity metrics used throughout tests/."""
import functools
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as f:
        return {k: f[k] for k in f.files}


@functools.lru_cache(maxsize=None)
def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def rel_err(a, b):
    """Tensor-level relative error: max|a-b| / max(|b|_inf, tiny). a, b: array-likes."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close(a, b, rtol, what="", atol_frac=None):
    """|a-b| <= rtol*|b| + atol elementwise, atol = atol_frac * max|b| (default atol_frac = rtol/10): the element-wise
    relative bar, with an absolute floor tied to the tensor's own scale so entries that cancel to ~0 are not compared
    at a precision the arithmetic never had."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    if a.size == 0:
        return
    atol = (rtol / 10 if atol_frac is None else atol_frac) * max(np.abs(b).max(), 1e-30)
    bad = np.abs(a - b) > rtol * np.abs(b) + atol
    if bad.any():
        i = np.unravel_index(np.argmax(np.abs(a - b) - rtol * np.abs(b)), a.shape)
        raise AssertionError(f"{what}: {bad.sum()}/{a.size} entries off; worst at {i}: got {a[i]!r} want {b[i]!r} "
                             f"(rtol={rtol}, atol={atol:.3e}, tensor rel err {rel_err(a, b):.3e})")


def summary_of(a, summary):
    """Recompute the (sum, L2, max, sampled entries) summary stored by make_golden._summary for tensor `a`."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    k = (len(summary) - 3) // 2
    idx = summary[3:3 + k].astype(np.int64)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()], idx.astype(np.float64), a[idx]])


def assert_close_after_adam(a, b, lr, steps, what="", rtol=2e-4, max_steps_apart=1.0, bad_frac=1e-4):
    """Parameters after `steps` Adam steps.  Adam divides by sqrt(v): where a gradient entry is pure rounding noise
    (columns of x that are almost always 0) the update m/sqrt(v) is O(1) in BOTH implementations and its sign follows
    the noise, so a handful of entries may differ by up to ~lr per step although both runs are correct float32
    evaluations.  Bar: element-wise 1e-4-class agreement for all but 1e-4 of the entries, and NO entry further apart
    than lr * steps (max_steps_apart = 2: where the two runs' gradients themselves differ by a flipped ReLU output, an
    entry whose gradient is ~0 can take the step in opposite directions)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape and np.isfinite(a).all(), what
    if a.size == 0:
        return
    scale = max(np.abs(b).max(), 1e-30)
    bad = np.abs(a - b) > rtol * np.abs(b) + rtol * scale
    assert bad.sum() <= max(1, int(bad_frac * a.size)), f"{what}: {bad.sum()}/{a.size} entries off"
    assert np.abs(a - b).max() <= lr * steps * 1.01 * max_steps_apart + rtol * scale, f"{what}: max diff {np.abs(a - b).max():.3e}"


def relu_flips(ca, cb):
    """Entries of the six ReLU outputs of the conv step that are > 0 in one run and 0 in the other.  A rounding-level
    difference in the forward pass (another summation order) can flip an activation whose exact value is ~0; its
    derivative then switches between 0 and 1, and every gradient upstream of it changes by one term of a long sum."""
    return sum(int(((ca[k] > 0) != (cb[k] > 0)).sum()) for k in ("a0", "a1", "a2", "t0", "b1", "b2"))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
