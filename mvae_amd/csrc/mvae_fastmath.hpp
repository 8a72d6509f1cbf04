// mvae_fastmath.hpp -- short, branch-free float32 elementary functions for the per-row manifold arithmetic.
//
// Why: a latent component is ONE dependent chain of ~25 transcendental evaluations executed by a single lane, once per
// launch; at batch 128 its latency (and the instruction fetch of its straight-line code) is on the critical path of
// the step.  ocml's coshf/sinhf/sinf/cosf/log1pf are 120-150 instructions each; the versions below are 15-35, share
// work between the pairs the manifolds always need together (cosh & sinh, sin & cos) and stay within ~2 ulp on the
// ranges the path uses (checked against float64 in tests/test_fastmath.py through the host build of this header).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MVF __host__ __device__ __forceinline__
#else
#define MVF inline
#endif

namespace mvf {

// exp / log: on the device the hardware base-2 pair (v_exp_f32 / v_log_f32, ~1 ulp) with one multiply, instead of
// ocml's ~12-instruction range-extended versions; |x| <= 85 on this path (cosh/sinh arguments are clamped there), so
// the rounding of x*log2(e) costs < 6e-6 relative at the very end of the range and < 1e-6 where the model operates.
#if defined(__HIP_DEVICE_COMPILE__)
MVF float fexp(float x) { return __expf(x); }
MVF float flog(float x) { return __logf(x); }
#else
MVF float fexp(float x) { return expf(x); }
MVF float flog(float x) { return logf(x); }
#endif

// cosh and sinh of the same argument from ONE exp:  e = exp(|x|), cosh = (e + 1/e)/2, sinh = sign(x)(e - 1/e)/2;
// for |x| < 0.35 sinh uses its odd Taylor polynomial (the difference would cancel).
MVF void sinhcosh(float x, float* sh, float* ch) {
  const float ax = fabsf(x);
  const float e = fexp(ax);
  const float ei = 1.0f / e;
  *ch = 0.5f * e + 0.5f * ei;
  const float x2 = ax * ax;
  // x + x^3/6 + x^5/120 + x^7/5040 + x^9/362880
  const float poly = ax + ax * x2 * (1.6666667163e-01f + x2 * (8.3333337680e-03f + x2 * (1.9841270114e-04f +
                                                                                        x2 * 2.7557314297e-06f)));
  const float big = 0.5f * e - 0.5f * ei;
  const float s = (ax < 0.35f) ? poly : big;
  *sh = copysignf(s, x);
}

// sin and cos of the same argument: Cody-Waite reduction by pi/2 (two fused steps, exact enough for |x| < 8192) and
// the classic minimax kernels on [-pi/4, pi/4].  Returns false for arguments outside the fast range.
MVF bool sincos_fast(float x, float* s, float* c) {
  const float k = rintf(x * 0.6366197723675814f);
  float r = fmaf(k, -1.5707963705062866f, x);   // (float)(pi/2)
  r = fmaf(k, 4.371139000186241e-08f, r);       // pi/2 - (float)(pi/2) = -4.37e-8  ->  subtract k*lo
  const float r2 = r * r;
  const float sp = r + r * r2 * (-1.6666667163e-01f + r2 * (8.3333337680e-03f + r2 * (-1.9841270114e-04f +
                                                                                     r2 * 2.7557314297e-06f)));
  const float cp = 1.0f + r2 * (-0.5f + r2 * (4.1666667908e-02f + r2 * (-1.3888889225e-03f +
                                                                       r2 * (2.4801587642e-05f + r2 * -2.7557314297e-07f))));
  const int q = ((int)k) & 3;
  const float s0 = (q & 1) ? cp : sp;
  const float c0 = (q & 1) ? sp : cp;
  *s = (q & 2) ? -s0 : s0;
  *c = ((q + 1) & 2) ? -c0 : c0;
  return fabsf(x) < 8192.0f;
}

// log(1 + e) for e >= 0 without cancellation: with u = fl(1 + e), log1p(e) = log(u) * e / (u - 1) (u != 1), else e.
MVF float log1p_pos(float e) {
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  const float l = flog(u) * (e / d);
  return (d == 0.0f) ? e : ((u > 3.0e38f) ? flog(e) : l);
}

}  // namespace mvf
