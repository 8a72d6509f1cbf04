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

Also pinned (g6_projected.npz): the reference-owned part of the projected sphere (d) -- exp_map_mu0,
inverse_exp_map_mu0, both parallel transports, lambda_x, projected_to_spherical, spherical_projected_distance, logdet
-- and Universal.radius / _choice (u).

NO DIRECT REFERENCE VECTORS -- PINNED THROUGH THE PINNED MODELS: the Poincare ball (p), and the one function through
which the projected sphere (d) and therefore the universal manifold (u) cross into the same library (mob_add ->
mobius_add(c=-K), spherical_projected.py:113).  That arithmetic lives in the third-party dependency geoopt==0.1.0
(reference pin: Makefile:18), which is not vendored in /root/reference and not installed here, so no reference output
exists for it ("parity unpinned" in the strict sense of a recorded vector).  `oracle/ops.py` restates the published
gyrovector formulas (Ganea et al. 2018) with geoopt-0.1.0's guard constants.  It is pinned twice over:
  * through the models that ARE pinned (tests/test_oracle_crossmodel.py, float64, 1e-9, values and gradients): the
    reference-owned isometries poincare_to_lorentz (poincare.py:167-170) and projected_to_spherical
    (spherical_projected.py:191-196) carry every p / d operator (exp0, exp, log, parallel transport, sample projection,
    distance, the whole component forward with d/d heads and d/d radius) onto its h / s counterpart, which is checked
    against vectors recorded from the reference (g1, g2);
  * on the properties AND TOLERANCES the reference's own tests state (tests/mvae/ops/test_poincare.py:140-211,
    test_spherical_projected.py:86-289: round trips to 5e-6, dist(mu, exp_mu(u)) = lambda_mu |u|, known answers) --
    restated in tests/test_oracle_projected.py.  Those tolerances decide the one guard that was in doubt: the mobius_add
    denominator is clamp_min(1e-15), not "+ 1e-5" (the latter misses the reference's round-trip tolerances by
    1e-4 ... 0.45).
What stays unverifiable is geoopt's behaviour strictly inside its guards (|x| < 1e-15, |artanh argument| > 1 - 1e-5).
"""
