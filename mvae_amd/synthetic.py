"""Deterministic synthetic inputs and initial states (no dataset / checkpoint exists on either box).

The recipes follow SURVEY.md section 8(d) / BASELINE.md section 3:
  * MNIST-like input: x_ij ~ Bernoulli(p_j), p = Generator(1234).rand(D)**3  (values {0,1}; mirrors the
    reference's dynamic binarisation, mt/data/image_reconstruction.py:44-53)
  * CIFAR-like input: x ~ U[0,1]^D (mirrors ToTensor, image_reconstruction.py:123-127)
  * eps ~ N(0,1)^{steps x B x sum(d)} from Generator(1000 + rank)
  * initial state: every parameter drawn U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear / nn.Conv2d default
    bound) from a generator keyed by crc32(parameter name) -- independent of module construction order, so the
    golden generator can `load_state_dict` the very same state into the reference model.

Everything is generated on the CPU with torch generators (bit-reproducible across boxes with the same torch build)
and moved to the device by the caller.
"""
import zlib
from typing import Dict, Iterable, Tuple

import torch


def pixel_probs(in_dim: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(1234)
    return torch.rand(in_dim, generator=g, dtype=torch.float64)**3


def binary_batches(steps: int, batch: int, in_dim: int, seed: int = 4321, dtype=torch.float32) -> torch.Tensor:
    """[steps, batch, in_dim] tensor of {0,1} -- a stand-in for dynamically binarised MNIST."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(steps, batch, in_dim, generator=g, dtype=torch.float64)
    return (u < pixel_probs(in_dim)).to(dtype)


def digits_like_batches(steps: int, batch: int, side: int = 28, n_classes: int = 10, seed: int = 4321,
                        dtype=torch.float32) -> torch.Tensor:
    """[steps, batch, side*side] {0,1} images with MNIST-like latent structure: every sample picks one of `n_classes`
    stroke-like prototypes (a few anisotropic Gaussian blobs on the pixel grid), a random sub-pixel shift and a random
    stroke thickness, and is then binarised dynamically, x = (p > U(0,1)) as in image_reconstruction.py:44-53.
    Unlike i.i.d. pixels this gives the posterior something to encode, so long training runs behave like the
    reference's MNIST runs (no posterior collapse)."""
    gp = torch.Generator().manual_seed(777)  # the class prototypes are the same for every seed (train / test splits)
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(side, dtype=torch.float64), torch.arange(side, dtype=torch.float64),
                            indexing="ij")
    n_blobs = 4
    cx = 6 + 16 * torch.rand(n_classes, n_blobs, generator=gp, dtype=torch.float64)
    cy = 6 + 16 * torch.rand(n_classes, n_blobs, generator=gp, dtype=torch.float64)
    sx = 1.5 + 3.0 * torch.rand(n_classes, n_blobs, generator=gp, dtype=torch.float64)
    sy = 1.5 + 3.0 * torch.rand(n_classes, n_blobs, generator=gp, dtype=torch.float64)
    n = steps * batch
    cls = torch.randint(0, n_classes, (n,), generator=g)
    shift = (torch.rand(n, 2, generator=g, dtype=torch.float64) - 0.5) * 3.0
    thick = 0.8 + 0.5 * torch.rand(n, 1, 1, generator=g, dtype=torch.float64)
    out = torch.empty(n, side * side, dtype=dtype)
    chunk = 4096
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        c = cls[lo:hi]
        dx = xx[None, None] - (cx[c] + shift[lo:hi, 0:1])[:, :, None, None]
        dy = yy[None, None] - (cy[c] + shift[lo:hi, 1:2])[:, :, None, None]
        e = torch.exp(-0.5 * ((dx / sx[c][:, :, None, None])**2 + (dy / sy[c][:, :, None, None])**2)).sum(dim=1)
        prob = (0.02 + 0.95 * torch.clamp(e * thick[lo:hi], max=1.0)).reshape(hi - lo, -1)
        u = torch.rand(hi - lo, side * side, generator=g, dtype=torch.float64)
        out[lo:hi] = (u < prob).to(dtype)
    return out.reshape(steps, batch, side * side)


def uniform_batches(steps: int, batch: int, in_dim: int, seed: int = 4321, dtype=torch.float32) -> torch.Tensor:
    """[steps, batch, in_dim] tensor in [0,1) -- a stand-in for CIFAR pixels (BCE with soft targets)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(steps, batch, in_dim, generator=g, dtype=torch.float64).to(dtype)


def eps_batches(steps: int, batch: int, total_true_dim: int, rank: int = 0, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + rank)
    return torch.randn(steps, batch, total_true_dim, generator=g, dtype=torch.float64).to(dtype)


def _fan_in(name: str, shape: Tuple[int, ...], fan_in_of_weight: Dict[str, int]) -> int:
    if len(shape) >= 2:
        fi = 1
        for s in shape[1:]:
            fi *= s
        return fi
    return fan_in_of_weight.get(name.rsplit(".", 1)[0], 1)


def synthetic_state(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], radius: float = 2.0,
                    dtype=torch.float32, transposed_conv: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """Name-keyed deterministic state. Radii (`*_nradius`, `*_pradius`) are set to `radius`."""
    named_shapes = list(named_shapes)
    tconv = set(transposed_conv)
    fan = {}
    for name, shape in named_shapes:
        if len(shape) >= 2:
            mod = name.rsplit(".", 1)[0]
            if mod in tconv:  # ConvTranspose2d weight is [in, out, kh, kw]; torch's fan_in uses dim 1
                fi = shape[1]
                for s in shape[2:]:
                    fi *= s
            else:
                fi = 1
                for s in shape[1:]:
                    fi *= s
            fan[mod] = fi
    state = {}
    for name, shape in named_shapes:
        if name.endswith("radius") or name.endswith("_curvature"):
            state[name] = torch.tensor(radius, dtype=dtype)
            continue
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        bound = 1.0 / (fan[name.rsplit(".", 1)[0]]**0.5)
        u = torch.rand(tuple(shape), generator=g, dtype=torch.float64)
        state[name] = ((2.0 * u - 1.0) * bound).to(dtype)
    return state
