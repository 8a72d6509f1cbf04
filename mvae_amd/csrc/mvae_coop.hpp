// mvae_coop.hpp -- wave-cooperative latent components for LARGE true dimensions (d >= 9: the reference's own `h40`,
// `s40-wn` models, tests/mvae/models/test_vae.py:212-249).
//
// mvae_math.hpp evaluates a component with one LANE per (row, component[, direction]) over small vectors that live in
// VGPRs -- for d <= 8.  Beyond that its arrays are indexed at run time and land in scratch memory; `h40` at B = 128 spent
// 391 us of a 434 us step there.  Here one WAVE evaluates one (row, component[, input direction]): lane i holds entry i of
// every ambient vector (z, mu, u, ...: A = d + 1 <= 64 entries for the hyperboloid and the sphere) and entry i - 1 of every
// true-dimension vector (head outputs, eps, sigma, v) -- the reference's expand_proj_dims (common.py:156-158: x = [0, v]) is
// then the identity.  Norms, dot and Lorentz products are wave reductions (DPP row operations + two row broadcasts,
// wave_sum), the scalar chain in between (cosh / sinh, acosh, logsinh, ... with the reference's guarded functions and
// derivative rules, the templates of mvae_math.hpp over T = float | Dual) is evaluated redundantly by every lane.
// Same formulas, operator by operator, as component_forward<KIND, ...>; only the order of the additions inside a
// reduction differs (a tree instead of index order -- like ATen's own vectorised sums).
// Kinds: hyperboloid, sphere, Euclidean, and (coop_projected) the two projected models -- Poincare ball, stereographically
// projected sphere -- with the universal component resolved to one of them or to Euclidean by the sign of its curvature.
#pragma once

namespace mv {

__device__ __forceinline__ float co_sum(float x) { return wave_sum(x); }
__device__ __forceinline__ Dual co_sum(Dual x) { return Dual{wave_sum(x.v), wave_sum(x.d)}; }
__device__ __forceinline__ float co_lane0(float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)); }
__device__ __forceinline__ Dual co_lane0(Dual x) { return Dual{co_lane0(x.v), co_lane0(x.d)}; }
template <typename T> __device__ __forceinline__ T co_sel(bool c, T a, T b) { return c ? a : b; }

// torch.norm(p=2) from the sum of squares, derivative 0 at the origin (norm2 of mvae_math.hpp)
__device__ __forceinline__ float co_norm(float ss) { return sqrtf(ss); }
__device__ __forceinline__ Dual co_norm(Dual ss) {
  const float nv = sqrtf(ss.v);
  return {nv, nv == 0.0f ? 0.0f : 0.5f * ss.d / nv};
}

// Optional further outputs of a cooperative evaluation (the stand-alone operator's log_q / log_p / mu / sigma): log-probabilities
// on every lane, mu in z's lane layout, sigma = entry lane - 1.
template <typename T> struct CoopExtra {
  T lq, lp, mu, sg;
};

// One (row, component[, direction]) per wave.  Lane i: m, l, e = entry i - 1 of the mean head, the logvar head (its ONE
// entry on every lane under the scalar parametrisation) and eps, for 1 <= i <= d; anything on the other lanes is ignored.
// Returns the component's KL term (every lane) and leaves in *z_lane: hyperboloid / sphere: entry `lane` of z (0 past A);
// Euclidean: entry lane - 1 (lanes 1 .. d).
template <int KIND, typename T>
__device__ __forceinline__ T coop_component(T m, T l, float e, T rp, int d, int lane, T* z_lane, CoopExtra<T>* ex = nullptr) {
  const bool act = lane >= 1 && lane <= d;
  const T zero = cst<T>(0.0f);
  m = co_sel(act, m, zero);
  const T sigma = co_sel(act, t_softplus(l) + 1e-5f, cst<T>(1.0f));  // component.py:72
  if constexpr (KIND == kEuclidean) {
    const T mu = m / 2.0f;                                           // euclidean.py:78-79
    *z_lane = co_sel(act, mu + e * sigma, zero);                     // Normal.rsample
    const T var_ratio = (sigma / 1.0f) * (sigma / 1.0f);
    const T t1 = ((mu - 0.0f) / 1.0f) * ((mu - 0.0f) / 1.0f);
    const T term = 0.5f * (var_ratio + t1 - 1.0f - t_log(var_ratio));  // kl_divergence(N(mu, sigma), N(0, 1))
    if (ex) {  // EuclideanNormal.log_prob (wrapped_distributions.py:39-42)
      ex->lq = co_sum(co_sel(act, normal_logprob_term(*z_lane - mu, sigma), zero));
      ex->lp = co_sum(co_sel(act, normal_logprob_term(*z_lane, cst<T>(1.0f)), zero));
      ex->mu = co_sel(act, mu, zero);
      ex->sg = sigma;
    }
    return co_sum(co_sel(act, term, zero));
  } else {
    constexpr bool HYP = KIND == kHyperboloid;
    const int A = d + 1;
    const T R = radius_of(rp);
    const T e0R = co_sel(lane == 0, R, 0.0f * R);  // mu_0 = R e_0 (hyperbolics.py:68-69 | spherical.py:70-71)
    auto prod = [&](T x, T y) {                    // <x, y>_L = sum(all) - 2 x0 y0 (hyperbolics.py:72-78) | <x, y>
      const T s = co_sum(x * y);
      if constexpr (HYP) return s - 2.0f * (co_lane0(x) * co_lane0(y));
      else return s;
    };
    // exp_map_mu0 (hyperbolics.py:114-121 | spherical.py:94-101)
    const T n = co_norm(co_sum(m * m));
    const T xn = n / R;
    const T nc = hard_clamp(n, 1e-12f, INFINITY);  // F.normalize(eps=1e-12)
    T c, s;
    if constexpr (HYP) g_cosh_sinh(xn, &c, &s);
    else t_cos_sin(xn, &c, &s);
    const T mu = co_sel(lane == 0, c * R, co_sel(act, s * ((m / nc) * R), zero));
    const T v = co_sel(act, e * sigma, zero);      // Normal(0, sigma).rsample; x = [0, v] is v itself in this layout
    // parallel_transport_mu0 (hyperbolics.py:87-93 | spherical.py:74-77)
    const T mu00 = co_lane0(mu);
    const T coef = prod(mu, v) / (R * (R + mu00));
    const T shifted = mu + e0R;                    // (dst[0] + R, dst[1], ...)
    const T u = HYP ? v + coef * shifted : v - coef * shifted;
    // exp_map(u, at = mu) (hyperbolics.py:106-111 | spherical.py:86-91)
    T un;
    if constexpr (HYP) un = g_sqrt(prod(u, u)) / R;
    else un = co_norm(co_sum(u * u)) / R;
    T c2, s2;
    if constexpr (HYP) g_cosh_sinh(un, &c2, &s2);
    else t_cos_sin(un, &c2, &s2);
    const T z = co_sel(lane < A, c2 * mu + s2 * (u / un), zero);
    *z_lane = z;
    // logdet of the projection Jacobian from a tangent vector (hyperbolics.py:58-65 | spherical.py:58-67)
    const float nm1 = (float)(A - 1 - 1);
    auto logdet = [&](T uu) {
      if constexpr (HYP) {
        const T r = g_sqrt(prod(uu, uu)) / R;
        return nm1 * (t_log(R) + g_logsinh(r) - t_log(r));
      } else {
        const T r = co_norm(co_sum(uu * uu)) / R;
        T cr, sr;
        t_cos_sin(r, &cr, &sr);
        return nm1 * (t_log(R) + t_log(hard_clamp(t_abs(sr), 1e-5f, INFINITY)) - t_log(hard_clamp(r, 1e-5f, INFINITY)));
      }
    };
    const T logdet_q = logdet(u);
    // prior WrappedNormal(mu_0, 1): inverse_sample_projection_mu0 (wrapped_normal.py:99-103)
    T alpha, coef2;
    if constexpr (HYP) {                            // hyperbolics.py:124-128
      alpha = -prod(e0R, z) / (R * R);
      coef2 = g_acosh(alpha) / g_sqrt(alpha * alpha - 1.0f);
    } else {                                        // spherical.py:104-109
      alpha = prod(e0R, z) / (R * R);
      coef2 = t_acos(hard_clamp(alpha, -1.0f, 1.0f)) / g_sqrt(1.0f - alpha * alpha);
    }
    const T u0 = co_sel(lane < A, coef2 * (z - alpha * e0R), zero);
    // inverse_parallel_transport_mu0 (hyperbolics.py:96-103 | spherical.py:80-83) with src = mu_0
    const T u00 = co_lane0(u0), src0 = co_lane0(e0R);
    const T sh0 = e0R + e0R;                        // (src[0] + R, 0, ...)
    T w;
    if constexpr (HYP) w = u0 + ((-u00) / (R + src0)) * sh0;
    else w = u0 - (u00 / (R + src0)) * sh0;
    const T v0 = co_sel(act, w, zero);
    const T logdet_p = logdet(u0);
    const T nq = co_sum(co_sel(act, normal_logprob_term(v, sigma), zero));
    const T np = co_sum(co_sel(act, normal_logprob_term(v0, cst<T>(1.0f)), zero));
    if (ex) {
      ex->lq = nq - logdet_q;
      ex->lp = np - logdet_p;
      ex->mu = mu;
      ex->sg = sigma;
    }
    return (nq - logdet_q) - (np - logdet_p);       // wrapped_normal.py:84-97, sampling_procedures.py:101-104
  }
}

// The projected models (poincare.py / spherical_projected.py; `p40`, `d40` of tests/mvae/models/test_vae.py:212-249): tangent
// dimension = ambient dimension = d.  Lane i holds entry i - 1 of EVERY vector (1 <= i <= d), so the Lorentz / spherical image of
// a point -- poincare_to_lorentz (poincare.py:167-170), projected_to_spherical (spherical_projected.py:191-196): [scalar, 2 R^2 y /
// den] -- puts its scalar on lane 0 and needs no shift; the log-det of the projection Jacobian is taken there
// (poincare.py:55-89 | spherical_projected.py:56-88), as in p_logdet / d_logdet.  Same operators in the same order as
// component_forward<kPoincare | kProjSphere>: exp_map_mu0, the conformal factor lambda, exp_map = mobius_add(mu, tanh | tan
// ...), log_map at the origin, geoopt 0.1.0's guards (MIN_NORM 1e-15 on norms and the mobius_add denominator, tanh clamp 15,
// artanh clamp 1 - 1e-5).  *z_lane = entry lane - 1 of z (lanes 1 .. d).
template <int KIND, typename T>
__device__ __forceinline__ T coop_projected(T m, T l, float e, T rp, int d, int lane, T* z_lane, CoopExtra<T>* ex = nullptr) {
  constexpr bool BALL = KIND == kPoincare;
  const bool act = lane >= 1 && lane <= d;
  const T zero = cst<T>(0.0f);
  m = co_sel(act, m, zero);
  const T sigma = co_sel(act, t_softplus(l) + 1e-5f, cst<T>(1.0f));  // component.py:72
  const T R = radius_of(rp);
  const T c = 1.0f / (R * R);
  const T sc = t_sqrt(c);
  auto dotp = [&](T x, T y) { return co_sum(x * y); };  // (entries off the active lanes are 0)
  auto nrm = [&](T x) { return co_norm(co_sum(x * x)); };
  auto lambda_at = [&](T x) {  // p_lambda | d_lambda
    if constexpr (BALL) return 2.0f / (1.0f - c * dotp(x, x));
    else return 2.0f / hard_clamp(1.0f + c * dotp(x, x), 1e-15f, INFINITY);
  };
  auto mob_add = [&](T x, T y, T cc) {  // p_mobius_add
    const T x2 = dotp(x, x), y2 = dotp(y, y), xy = dotp(x, y);
    const T fa = 1.0f + 2.0f * cc * xy + cc * y2;
    const T fb = 1.0f - cc * x2;
    const T den = hard_clamp(1.0f + 2.0f * cc * xy + cc * cc * x2 * y2, 1e-15f, INFINITY);
    return (fa * x + fb * y) / den;
  };
  const T cs = BALL ? c : -c;  // curvature sign of the gyro-addition: ball c, projected sphere -K
  // exp_map_mu0
  T mu;
  {
    const T n = hard_clamp(nrm(m), 1e-15f, INFINITY);
    if constexpr (BALL) {
      const T t = t_tanh(hard_clamp(sc * n, -15.0f, 15.0f));
      mu = t * m / (sc * n);
    } else {
      const T r = n / R;
      mu = t_tan(r) * m / r;
    }
  }
  const T v = co_sel(act, e * sigma, zero);  // Normal(0, sigma).rsample
  const T lam = lambda_at(mu);
  const T u = v / lam;                       // poincare.py:152-157 | spherical_projected.py:178-181
  // exp_map(u, at = mu)
  T z;
  {
    const T n = hard_clamp(nrm(u), 1e-15f, INFINITY);
    if constexpr (BALL) {
      const T t = t_tanh(hard_clamp(sc / 2.0f * lambda_at(mu) * n, -15.0f, 15.0f));
      z = mob_add(mu, t * u / (sc * n), cs);
    } else {
      const T r = n / R;
      const T t = t_tan(r * lambda_at(mu) / 2.0f);
      z = mob_add(mu, t * u / r, cs);
    }
  }
  z = co_sel(act, z, zero);
  *z_lane = z;
  // the model image of a ball / projected-sphere point, and the log-det through it
  auto prod = [&](T x, T y) {  // <x, y>_L | <x, y> on the (d + 1)-entry images
    const T sres = co_sum(x * y);
    if constexpr (BALL) return sres - 2.0f * (co_lane0(x) * co_lane0(y));
    else return sres;
  };
  auto to_model = [&](T y) {
    const T n = nrm(y);
    const T n2 = n * n, r2 = R * R;
    if constexpr (BALL) {
      const T den = r2 - n2;
      return co_sel(lane == 0, (R * (r2 + n2)) / den, (2.0f * r2 * y) / den);
    } else {
      const T den = n2 + r2;
      return co_sel(lane == 0, (R * (r2 - n2)) / den, (2.0f * r2 * y) / den);
    }
  };
  const float nm1 = (float)(d - 1);  // (n - 1) of the (d + 1)-entry model
  auto logdet = [&](T at, T zz) {
    const T zl = to_model(zz), ml = to_model(at);
    T alpha, coef;
    if constexpr (BALL) {  // log_map<kHyperboloid>, logdet_u<kHyperboloid>
      alpha = -prod(ml, zl) / (R * R);
      coef = g_acosh(alpha) / g_sqrt(alpha * alpha - 1.0f);
    } else {
      alpha = prod(ml, zl) / (R * R);
      coef = t_acos(hard_clamp(alpha, -1.0f, 1.0f)) / g_sqrt(1.0f - alpha * alpha);
    }
    const T uu = co_sel(lane <= d, coef * (zl - alpha * ml), zero);
    if constexpr (BALL) {
      const T r = g_sqrt(prod(uu, uu)) / R;
      return nm1 * (t_log(R) + g_logsinh(r) - t_log(r));
    } else {
      const T r = co_norm(co_sum(uu * uu)) / R;
      T cr, sr;
      t_cos_sin(r, &cr, &sr);
      return nm1 * (t_log(R) + t_log(hard_clamp(t_abs(sr), 1e-5f, INFINITY)) - t_log(hard_clamp(r, 1e-5f, INFINITY)));
    }
  };
  const T logdet_q = logdet(mu, z);
  // prior at the origin: log_map(z, at = 0), v0 = u0 * lambda(0)
  const T origin = zero;
  const T sub = mob_add(-origin, z, cs);
  T u0;
  if constexpr (BALL) {
    const T sn = hard_clamp(nrm(sub), 1e-15f, INFINITY);
    const T f = 2.0f / sc / lambda_at(origin) * p_artanh(sc * sn);
    u0 = f * sub / sn;
  } else {
    const T nm = hard_clamp(nrm(sub), 1e-15f, INFINITY) / R;
    const T f = 2.0f / lambda_at(origin) * t_atan(nm);
    u0 = f * (sub / nm);
  }
  const T v0 = co_sel(act, u0 * lambda_at(origin), zero);  // poincare.py:160-164 | spherical_projected.py:184-188
  const T logdet_p = logdet(origin, z);
  const T nq = co_sum(co_sel(act, normal_logprob_term(v, sigma), zero));
  const T np = co_sum(co_sel(act, normal_logprob_term(v0, cst<T>(1.0f)), zero));
  if (ex) {
    ex->lq = nq - logdet_q;
    ex->lp = np - logdet_p;
    ex->mu = co_sel(act, mu, zero);
    ex->sg = sigma;
  }
  return (nq - logdet_q) - (np - logdet_p);
}

// dispatch on the (runtime, wave-uniform) kind; false if the kind has no cooperative form
template <typename T>
__device__ __forceinline__ bool coop_eval(int kind, T m, T l, float e, T rp, int d, int lane, T* z_lane, T* kl,
                                          CoopExtra<T>* ex = nullptr) {
  kind = resolve_universal(kind, rp);  // `u`: Poincare ball / projected sphere / Euclidean by the sign of K (wave-uniform)
  switch (kind) {
    case kEuclidean: *kl = coop_component<kEuclidean, T>(m, l, e, rp, d, lane, z_lane, ex); return true;
    case kHyperboloid: *kl = coop_component<kHyperboloid, T>(m, l, e, rp, d, lane, z_lane, ex); return true;
    case kSphere: *kl = coop_component<kSphere, T>(m, l, e, rp, d, lane, z_lane, ex); return true;
    case kPoincare: *kl = coop_projected<kPoincare, T>(m, l, e, rp, d, lane, z_lane, ex); return true;
    case kProjSphere: *kl = coop_projected<kProjSphere, T>(m, l, e, rp, d, lane, z_lane, ex); return true;
    default: return false;
  }
}

// lane that holds entry `idx` of z after coop_eval: kinds whose ambient dimension is d + 1 (h, s) keep entry i on lane i,
// the others (e, p, d, u) entry i - 1
__host__ __device__ inline bool coop_z_shifted(int kind) { return kind != kHyperboloid && kind != kSphere; }

// host side: every component fits one wave
inline bool coop_eligible(const CompTable& t) {
  for (int i = 0; i < t.n; ++i)
    if (t.c[i].true_dim + 1 > 64) return false;
  return true;
}

}  // namespace mv
