"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-PyTorch *CPU* restatement of the per-batch hot path of oskopek/mvae (ModelVAE.train_step and everything it
calls), written from the reference's behaviour with each function citing the reference file:line it follows.

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` -- as the
checker / the timed CPU baseline only.  Nothing under `mvae_amd/` imports it; the product path raises if the HIP
library is missing instead of falling back to this code.

Pinning: `tests/test_oracle_golden.py` checks this restatement against vectors recorded from the reference itself
(`tests/golden/make_golden.py`, run in the build container where /root/reference is importable under a shim that
does not touch reference arithmetic).  Pinned: Hyperboloid (h), Sphere (s), Euclidean (e) primitives, guarded scalar
functions and their custom backward rules, the component forward/KL with gradients, the whole train step (fwd, ELBO,
bwd, Adam + curvature SGD, radius warm-up) for MLP and conv architectures, log_likelihood, the model-string parser.

PARITY UNPINNED: the Poincare ball (p).  Its arithmetic lives in the third-party dependency geoopt==0.1.0
(reference pin: Makefile:18), which is not vendored in /root/reference and not installed here, so no reference
output exists for it.  `oracle/ops.py` restates the published gyrovector formulas (Ganea et al. 2018) with
geoopt-0.1.0's guard constants as best known; it is validated only by the properties the reference's own tests
state (round trips, dist(mu, exp_mu(u)) = lambda_mu |u|, agreement with the hyperboloid model through
poincare_to_lorentz / lorentz_to_poincare).
"""
