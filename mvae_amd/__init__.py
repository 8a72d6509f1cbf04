"""mvae_amd -- MI355X (gfx950) implementation of the per-batch hot path of mixed-curvature VAEs.

Layers, bottom up:
  csrc/ + libmvae_hip.so   HIP kernels behind the C ABI of include/mvae_hip.h (built by `python -m mvae_amd.build`)
  _lib, functional         ctypes binding and tensor-level wrappers
  engine, runner, conv     flat HBM layout, fused train step, HIP-graph replay, conv architecture
  distributed              data-parallel step over torch.distributed (RCCL)
  ops, components, sampling, distributions, models, trainer, data, utils, run
                           host mirror of the reference's Python interface (same names and argument meaning)
There is no CPU execution path: importing is free, running needs a HIP device and the built library.
"""
__version__ = "0.1.0"
ABI_VERSION = 12  # == MVAE_ABI_VERSION of include/mvae_hip.h, checked against the library at load time
