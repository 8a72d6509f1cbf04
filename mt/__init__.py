"""`mt` -- the reference's import paths (oskopek/mvae: `mt.mvae.ops`, `mt.mvae.components`, `mt.mvae.models`,
`mt.data`, `mt.examples.run`, ...) served by the MI355X implementation in `mvae_amd`.

A user of the reference switches by putting this repository ahead of the reference on PYTHONPATH: the same
`from mt.mvae.ops import hyperbolics as H`, `from mt.mvae.models import FeedForwardVAE, Trainer`,
`python -m mt.examples.run --dataset=mnist --model=h2,s2,e2 ...` keep working, now on libmvae_hip.so.
The modules here are thin: classes come from `mvae_amd`, the ops modules' free functions call `mvae_amd.functional`.
float32 on a HIP device only; there is no CPU execution path.
"""
