// mvae_api.hip -- the operator-level part of the C ABI (include/mvae_hip.h): manifold primitives and their backward,
// the reference's guarded scalar functions, the per-component operators, generic dense layers, log-likelihood helpers.
#include "mvae_common.hpp"
#include "mvae_coop.hpp"
#include <atomic>
constexpr int kMaxDevices = 64;  // per-device "attribute set" flags of the kernels that need more than 64 KB of dynamic LDS

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a, long long b) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}
int hip_fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

extern "C" int mvae_abi_version(void) { return MVAE_ABI_VERSION; }
extern "C" const char* mvae_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ generic kernels (API)
template <bool RELU>
__global__ __launch_bounds__(256) void k_linear_fwd(const float* x, const float* W, const float* b, float* y, int M,
                                                    int N, int K) {
  __shared__ float red[4][16][17];
  job_linear_fwd<RELU>(red, x, K, W, K, b, y, N, M, N, K, blockIdx.y, blockIdx.x);
}

__global__ __launch_bounds__(256) void k_linear_bwd(const float* x, const float* W, const float* dy, float* dW,
                                                    float* db, float* dx, int M, int N, int K, int relu_in, int n_dw,
                                                    int n_dx) {
  __shared__ float red[4][16][17];
  int b = blockIdx.x;
  const int ntN = (N + 15) / 16, ntK = (K + 15) / 16, ntM = (M + 15) / 16;
  if (b < n_dx) {  // dx[M,K] = dy[M,N] W[N,K]
    job_nn(red, dy, N, M, b / ntK, W, K, K, b % ntK, N, relu_in ? x : nullptr, K, dx, K);
    return;
  }
  b -= n_dx;
  if (b < n_dw) {  // dW[N,K] = dy^T x
    job_tn(red, dy, N, N, b / ntK, x, K, K, b % ntK, M, dW, K);
    return;
  }
  b -= n_dw;
  (void)ntM;
  (void)ntN;
  job_colsum(&red[0][0][0], dy, N, M, N, b * kColsPerBlock, db);
}

// Linear backward with FEW outputs and a WIDE input: job_linear_bwd_skn (mvae_common.hpp), one workgroup per 32 columns
// of K plus one for the bias gradient.
template <int NN>  // NN = N rounded up to a multiple of 4
__global__ __launch_bounds__(256) void k_linear_bwd_skn(const float* x, const float* W, const float* dy, float* dW,
                                                        float* db, float* dx, int M, int N, int K, int relu_in) {
  __shared__ SknLds<NN> lds;
  job_linear_bwd_skn<NN, false>(lds, (int)blockIdx.x, x, W, dy, dW, db, dx, M, N, K, relu_in);
}

// ------------------------------------------------------------------------------------------------ primitives (API)
// Every primitive is ONE register-level template over the scalar type (prim_eval): T = float is the forward kernel,
// T = Dual the backward kernel (one thread per (row, input entry) evaluates the primitive along that input direction
// and contracts the output tangents with the upstream gradient -- the custom derivative rules of mvae_math.hpp apply,
// so torch.autograd sees exactly the reference's gradients).
enum PrimOp {
  OP_EXP0 = 0, OP_LOG0, OP_PT0, OP_IPT0, OP_SAMPLE, OP_ISAMPLE, OP_LOGDET, OP_EXP, OP_LOG, OP_DIST, OP_DIST_GYRO,
  OP_LPROD, OP_LNORM, OP_TO_BALL, OP_TO_AMBIENT, OP_LAMBDA, OP_MOBADD,
  // kind-independent diagonal-normal pieces (a = value / eps / loc; b = loc, c3 = scale, BOTH broadcast over at_rows)
  OP_NORMAL_LOGPROB, OP_NORMAL_RSAMPLE, OP_NORMAL_KL, OP_COUNT
};

// lengths of the row vectors of (op, RESOLVED kind, d): inputs a, b, c; outputs o1, o2.  b is the operand that may be
// broadcast over leading sample dims (row r reads b[r % at_rows]).
struct PrimShape {
  int na, nb, nc, n1, n2;
};
__host__ __device__ inline PrimShape prim_shape(int op, int kind, int d) {
  const int A = ambient_dim(kind, d);
  const bool proj = kind == kPoincare || kind == kProjSphere;
  switch (op) {
    case OP_EXP0: return PrimShape{d, 0, 0, A, 0};
    case OP_LOG0: return PrimShape{A, 0, 0, A, 0};
    case OP_PT0:
    case OP_IPT0: return PrimShape{A, A, 0, A, 0};
    case OP_SAMPLE: return PrimShape{d, A, 0, A, A};
    case OP_ISAMPLE: return PrimShape{A, A, 0, A, d};
    case OP_LOGDET:
      if (proj) return PrimShape{0, A, A, 1, 0};
      if (kind == kEuclidean) return PrimShape{0, 0, 0, 1, 0};
      return PrimShape{A, 0, 0, 1, 0};
    case OP_EXP:
    case OP_LOG:
    case OP_MOBADD: return PrimShape{A, A, 0, A, 0};
    case OP_LNORM:
    case OP_LAMBDA: return PrimShape{A, 0, 0, 1, 0};
    case OP_TO_BALL: return PrimShape{A, 0, 0, d, 0};       // h -> p, s -> d: ambient d+1 -> d
    case OP_TO_AMBIENT: return PrimShape{d, 0, 0, d + 1, 0};  // p -> h, d -> s
    case OP_NORMAL_LOGPROB: return PrimShape{d, d, d, 1, 0};
    case OP_NORMAL_RSAMPLE: return PrimShape{d, d, d, d, 0};
    case OP_NORMAL_KL: return PrimShape{d, 0, d, 1, 0};
    default: return PrimShape{A, A, 0, 1, 0};  // geodesic distances, Lorentz product
  }
}

template <int OP, int KIND, int DMAX, typename T>
__device__ __forceinline__ void prim_eval(const T* a, const T* b, const T* c3, T rp, int d, T* o1, T* o2) {
  constexpr int AMAX = DMAX + 1;
  MV_BOUNDS(AMAX);
  const int A = ambient_dim(KIND, d);
  T R = cst<T>(0.f);
  if constexpr (KIND != kEuclidean) R = radius_of(rp);
  if constexpr (OP == OP_EXP0) {
    exp_map_mu0<KIND, AMAX>(a, d, R, o1);
  } else if constexpr (OP == OP_LOG0) {
    log_map_mu0<KIND, AMAX>(a, A, R, o1);
  } else if constexpr (OP == OP_PT0) {
    pt_mu0<KIND, AMAX>(a, b, A, R, o1);
  } else if constexpr (OP == OP_IPT0) {
    inv_pt_mu0<KIND, AMAX>(a, b, A, R, o1);
  } else if constexpr (OP == OP_SAMPLE) {  // a = v[d], b = at[A] -> o1 = z, o2 = u
    if constexpr (KIND == kEuclidean) {
      MV_FOR(i, 0, d) o2[i] = a[i];
    } else if constexpr (KIND == kPoincare || KIND == kProjSphere) {
      T lam;
      if constexpr (KIND == kPoincare) lam = p_lambda<AMAX>(b, A, 1.0f / (R * R));
      else lam = d_lambda<AMAX>(b, A, 1.0f / (R * R));
      MV_FOR(i, 0, d) o2[i] = a[i] / lam;
    } else {
      T x[AMAX];
      x[0] = cst<T>(0.f);
      MV_FOR(i, 1, A) x[i] = a[i - 1];
      pt_mu0<KIND, AMAX>(x, b, A, R, o2);
    }
    exp_map<KIND, AMAX>(o2, b, A, R, o1);
  } else if constexpr (OP == OP_ISAMPLE) {  // a = z[A], b = at[A] -> o1 = u[A], o2 = v[d]
    log_map<KIND, AMAX>(a, b, A, R, o1);
    if constexpr (KIND == kEuclidean) {
      MV_FOR(i, 0, d) o2[i] = o1[i];
    } else if constexpr (KIND == kPoincare || KIND == kProjSphere) {
      T lam;
      if constexpr (KIND == kPoincare) lam = p_lambda<AMAX>(b, A, 1.0f / (R * R));
      else lam = d_lambda<AMAX>(b, A, 1.0f / (R * R));
      MV_FOR(i, 0, d) o2[i] = o1[i] * lam;
    } else {
      T w[AMAX];
      inv_pt_mu0<KIND, AMAX>(o1, b, A, R, w);
      MV_FOR(i, 1, A) o2[i - 1] = w[i];
    }
  } else if constexpr (OP == OP_LOGDET) {  // a = u (h, s) ; b = mu, c3 = z (p, d)
    if constexpr (KIND == kEuclidean) o1[0] = cst<T>(0.f);
    else if constexpr (KIND == kPoincare) o1[0] = p_logdet<AMAX>(b, c3, A, R);
    else if constexpr (KIND == kProjSphere) o1[0] = d_logdet<AMAX>(b, c3, A, R);
    else o1[0] = logdet_u<KIND, AMAX>(a, A, R);
  } else if constexpr (OP == OP_EXP) {  // a = tangent vector at b
    exp_map<KIND, AMAX>(a, b, A, R, o1);
  } else if constexpr (OP == OP_LOG) {  // a = point, b = base point
    log_map<KIND, AMAX>(a, b, A, R, o1);
  } else if constexpr (OP == OP_DIST || OP == OP_DIST_GYRO) {  // geodesic distance between the points a and b
    o1[0] = geodesic_distance<KIND, AMAX>(a, b, A, R, OP == OP_DIST_GYRO);
  } else if constexpr (OP == OP_LPROD) {  // hyperbolics.py:72-78 (h); the plain dot product elsewhere
    if constexpr (KIND == kHyperboloid) o1[0] = lorentz_product<AMAX>(a, b, A);
    else o1[0] = dot<AMAX>(a, b, A);
  } else if constexpr (OP == OP_LNORM) {  // hyperbolics.py:81-84 (h): guarded sqrt of <x,x>_L; torch.norm elsewhere
    if constexpr (KIND == kHyperboloid) o1[0] = g_sqrt(lorentz_product<AMAX>(a, a, A));
    else o1[0] = norm2<AMAX>(a, A);
  } else if constexpr (OP == OP_TO_BALL) {  // lorentz_to_poincare hyperbolics.py:151-152 | spherical_to_projected spherical.py:132-133
    if constexpr (KIND == kHyperboloid || KIND == kSphere) {
      MV_FOR(i, 1, A) o1[i - 1] = R * a[i] / (R + a[0]);
    } else {
      MV_FOR(i, 0, d) o1[i] = a[i];
    }
  } else if constexpr (OP == OP_TO_AMBIENT) {  // poincare_to_lorentz poincare.py:167-170 | projected_to_spherical spherical_projected.py:191-196
    if constexpr (KIND == kPoincare) poincare_to_lorentz<AMAX>(a, d, R, o1);
    else if constexpr (KIND == kProjSphere) projected_to_spherical<AMAX>(a, d, R, o1);
    else {
      MV_FOR(i, 0, d + 1) o1[i] = cst<T>(0.f);
    }
  } else if constexpr (OP == OP_LAMBDA) {  // geoopt lambda_x (p) | lambda_x spherical_projected.py:124-129 (d)
    if constexpr (KIND == kPoincare) o1[0] = p_lambda<AMAX>(a, A, 1.0f / (R * R));
    else if constexpr (KIND == kProjSphere) o1[0] = d_lambda<AMAX>(a, A, 1.0f / (R * R));
    else o1[0] = cst<T>(2.0f);
  } else if constexpr (OP == OP_NORMAL_LOGPROB) {  // sum_i log N(a_i; loc_i, scale_i), torch Normal.log_prob(.).sum(-1)
    T acc = cst<T>(0.f);
    MV_FOR(i, 0, d) {
      T term = normal_logprob_term(a[i] - b[i], c3[i]);
      acc = (i == 0) ? term : acc + term;
    }
    o1[0] = acc;
  } else if constexpr (OP == OP_NORMAL_RSAMPLE) {  // Normal.rsample: loc + eps * scale
    MV_FOR(i, 0, d) o1[i] = b[i] + a[i] * c3[i];
  } else if constexpr (OP == OP_NORMAL_KL) {  // kl_divergence(N(a, c3), N(0, 1)).sum(-1)  (torch _kl_normal_normal)
    T acc = cst<T>(0.f);
    MV_FOR(i, 0, d) {
      T var_ratio = (c3[i] / 1.0f) * (c3[i] / 1.0f);
      T t1 = ((a[i] - 0.0f) / 1.0f) * ((a[i] - 0.0f) / 1.0f);
      T term = 0.5f * (var_ratio + t1 - 1.0f - t_log(var_ratio));
      acc = (i == 0) ? term : acc + term;
    }
    o1[0] = acc;
  } else {  // OP_MOBADD: geoopt mobius_add with c = 1/R^2 (p) | mob_add with K = 1/R^2 (d), spherical_projected.py:107-113
    if constexpr (KIND == kPoincare) p_mobius_add<AMAX>(a, b, A, 1.0f / (R * R), o1);
    else if constexpr (KIND == kProjSphere) p_mobius_add<AMAX>(a, b, A, -(1.0f / (R * R)), o1);
    else {
      MV_FOR(i, 0, A) o1[i] = a[i] + b[i];
    }
  }
}

template <int OP, int KIND, int DMAX>
__device__ __forceinline__ void prim_row(const float* a, const float* b, const float* c3, float* o1, float* o2, int d,
                                         float rp, int64_t r, int64_t at_rows) {
  constexpr int AMAX = DMAX + 1;
  MV_BOUNDS(AMAX);
  const PrimShape sh = prim_shape(OP, KIND, d);
  float ta[AMAX], tb[AMAX], tc[AMAX], t1[AMAX], t2[AMAX];
  const int64_t ar = r % at_rows;
  MV_FOR(i, 0, sh.na) ta[i] = a[r * sh.na + i];
  MV_FOR(i, 0, sh.nb) tb[i] = b[ar * sh.nb + i];
  const int64_t cr = (OP >= OP_NORMAL_LOGPROB) ? ar : r;  // the normal ops broadcast BOTH parameters
  MV_FOR(i, 0, sh.nc) tc[i] = c3[cr * sh.nc + i];
  prim_eval<OP, KIND, DMAX, float>(ta, tb, tc, rp, d, t1, t2);
  MV_FOR(i, 0, sh.n1) o1[r * sh.n1 + i] = t1[i];
  if (o2) {
    MV_FOR(i, 0, sh.n2) o2[r * sh.n2 + i] = t2[i];
  }
}

#define MV_PRIM_KIND_SWITCH(FN, ...)                                   \
  if constexpr (OP >= OP_NORMAL_LOGPROB) {                             \
    FN<OP, kEuclidean, DMAX>(__VA_ARGS__); /* kind-independent */      \
  } else                                                               \
  switch (kind) {                                                      \
    case kEuclidean: FN<OP, kEuclidean, DMAX>(__VA_ARGS__); break;     \
    case kHyperboloid: FN<OP, kHyperboloid, DMAX>(__VA_ARGS__); break; \
    case kSphere: FN<OP, kSphere, DMAX>(__VA_ARGS__); break;           \
    case kProjSphere: FN<OP, kProjSphere, DMAX>(__VA_ARGS__); break;   \
    default: FN<OP, kPoincare, DMAX>(__VA_ARGS__); break;              \
  }

template <int OP, int DMAX>
__global__ __launch_bounds__(256) void k_prim(int kind, const float* a, const float* b, const float* c3, float* o1,
                                              float* o2, int64_t rows, int64_t at_rows, int d,
                                              const float* radius_param) {
  float rp = (kind == kEuclidean || !radius_param) ? 0.f : radius_param[0];
  kind = resolve_universal(kind, rp);  // for `u`, radius_param holds the curvature K
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    MV_PRIM_KIND_SWITCH(prim_row, a, b, c3, o1, o2, d, rp, r, at_rows)
  }
}

// Backward of one (row, input entry j): j enumerates a, then b, then c3, then the radius / curvature parameter.
// ga/gb/gc are per-ROW gradients ([rows, na], [rows, nb], [rows, nc]; a broadcast `b` is reduced by the caller) and
// gr[rows] the per-row terms of the radius gradient (the caller sums them in index order: no atomics).
template <int OP, int KIND, int DMAX>
__device__ __forceinline__ void prim_bwd_item(const float* a, const float* b, const float* c3, const float* g1,
                                              const float* g2, float* ga, float* gb, float* gc, float* gr, int d,
                                              Dual rp, int64_t r, int64_t at_rows, int j) {
  constexpr int AMAX = DMAX + 1;
  MV_BOUNDS(AMAX);
  const PrimShape sh = prim_shape(OP, KIND, d);
  Dual ta[AMAX], tb[AMAX], tc[AMAX], t1[AMAX], t2[AMAX];
  const int64_t ar = r % at_rows;
  MV_FOR(i, 0, sh.na) ta[i] = Dual{a[r * sh.na + i], j == i ? 1.f : 0.f};
  MV_FOR(i, 0, sh.nb) tb[i] = Dual{b[ar * sh.nb + i], j == sh.na + i ? 1.f : 0.f};
  const int64_t cr = (OP >= OP_NORMAL_LOGPROB) ? ar : r;
  MV_FOR(i, 0, sh.nc) tc[i] = Dual{c3[cr * sh.nc + i], j == sh.na + sh.nb + i ? 1.f : 0.f};
  rp.d = (j == sh.na + sh.nb + sh.nc) ? rp.d : 0.f;
  MV_FOR(i, 0, AMAX) t2[i] = Dual{0.f, 0.f};
  prim_eval<OP, KIND, DMAX, Dual>(ta, tb, tc, rp, d, t1, t2);
  float g = 0.f;
  MV_FOR(i, 0, sh.n1) g += g1[r * sh.n1 + i] * t1[i].d;
  if (g2) {
    MV_FOR(i, 0, sh.n2) g += g2[r * sh.n2 + i] * t2[i].d;
  }
  if (j < sh.na) {
    if (ga) ga[r * sh.na + j] = g;
  } else if (j < sh.na + sh.nb) {
    if (gb) gb[r * sh.nb + (j - sh.na)] = g;
  } else if (j < sh.na + sh.nb + sh.nc) {
    if (gc) gc[r * sh.nc + (j - sh.na - sh.nb)] = g;
  } else if (gr) {
    gr[r] = g;
  }
}

template <int OP, int DMAX>
__global__ __launch_bounds__(256) void k_prim_bwd(int kind, const float* a, const float* b, const float* c3,
                                                  const float* g1, const float* g2, float* ga, float* gb, float* gc,
                                                  float* gr, int64_t rows, int64_t at_rows, int d,
                                                  const float* radius_param) {
  // the radius / curvature direction carries the chain through resolve_universal (d radius / d K for `u`)
  Dual rp = Dual{(kind == kEuclidean || !radius_param) ? 0.f : radius_param[0], 1.f};
  kind = resolve_universal(kind, rp);
  const PrimShape sh = prim_shape(OP, kind, d);
  const int nin = sh.na + sh.nb + sh.nc + 1;
  const int64_t items = rows * nin;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t r = it / nin;
    const int j = (int)(it - r * nin);
    MV_PRIM_KIND_SWITCH(prim_bwd_item, a, b, c3, g1, g2, ga, gb, gc, gr, d, rp, r, at_rows, j)
  }
}
#undef MV_PRIM_KIND_SWITCH

static int prim_check(int kind, int64_t rows, int d, const float* rp) {
  if (kind < 0 || kind >= kNumKinds) return fail(MVAE_E_BADARG, "unknown manifold kind%s (%lld)", "", kind);
  if (rows < 0 || d < 1) return fail(MVAE_E_BADARG, "bad rows/d%s (%lld)", "", d);
  if (d > MVAE_MAX_TRUE_DIM) return fail(MVAE_E_UNSUPPORTED, "true_dim > MVAE_MAX_TRUE_DIM%s (%lld)", "", d);
  if (kind != MVAE_EUCLIDEAN && !rp) return fail(MVAE_E_BADARG, "radius_param is NULL%s", "");
  return 0;
}

template <int OP>
static int launch_prim(int kind, const float* a, const float* b, const float* c3, float* o1, float* o2, int64_t rows,
                       int64_t at_rows, int d, const float* rp, void* stream) {
  int rc = prim_check(kind, rows, d, rp);
  if (rc) return rc;
  if (rows == 0) return 0;
  if (at_rows < 1) at_rows = rows;
  int grid = (int)((rows + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
#define PRIM_CASE(B) \
  case B: hipLaunchKernelGGL((k_prim<OP, B>), dim3(grid), dim3(256), 0, s, kind, a, b, c3, o1, o2, rows, at_rows, d, rp); break;
  switch (bucket_of(d)) {
    PRIM_CASE(2) PRIM_CASE(4) PRIM_CASE(8) PRIM_CASE(16) PRIM_CASE(32) PRIM_CASE(64)
  }
#undef PRIM_CASE
  LAUNCH_CHECK("primitive launch");
  return 0;
}

template <int OP>
static int launch_prim_bwd(int kind, const float* a, const float* b, const float* c3, const float* g1, const float* g2,
                           float* ga, float* gb, float* gc, float* gr, int64_t rows, int64_t at_rows, int d,
                           const float* rp, void* stream) {
  int rc = prim_check(kind, rows, d, rp);
  if (rc) return rc;
  if (rows == 0) return 0;
  if (at_rows < 1) at_rows = rows;
  const int64_t items = rows * (3 * (int64_t)(d + 1) + 1);  // upper bound on rows * inputs
  int grid = (int)((items + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipStream_t s = (hipStream_t)stream;
#define PRIM_CASE(B)                                                                                                   \
  case B: hipLaunchKernelGGL((k_prim_bwd<OP, B>), dim3(grid), dim3(256), 0, s, kind, a, b, c3, g1, g2, ga, gb, gc, gr, \
                             rows, at_rows, d, rp); break;
  switch (bucket_of(d)) {
    PRIM_CASE(2) PRIM_CASE(4) PRIM_CASE(8) PRIM_CASE(16) PRIM_CASE(32) PRIM_CASE(64)
  }
#undef PRIM_CASE
  LAUNCH_CHECK("primitive backward launch");
  return 0;
}

extern "C" int mvae_exp_map_mu0(int kind, const float* x, float* out, int64_t rows, int d, const float* rp, void* st) {
  if (!x || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_EXP0>(kind, x, nullptr, nullptr, out, nullptr, rows, rows, d, rp, st);
}
extern "C" int mvae_inverse_exp_map_mu0(int kind, const float* x, float* out, int64_t rows, int d, const float* rp,
                                        void* st) {
  if (!x || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_LOG0>(kind, x, nullptr, nullptr, out, nullptr, rows, rows, d, rp, st);
}
extern "C" int mvae_parallel_transport_mu0(int kind, const float* x, const float* dst, float* out, int64_t rows, int d,
                                           const float* rp, void* st) {
  if (!x || !dst || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_PT0>(kind, x, dst, nullptr, out, nullptr, rows, rows, d, rp, st);
}
extern "C" int mvae_inverse_parallel_transport_mu0(int kind, const float* x, const float* src, float* out,
                                                   int64_t rows, int d, const float* rp, void* st) {
  if (!x || !src || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_IPT0>(kind, x, src, nullptr, out, nullptr, rows, rows, d, rp, st);
}
extern "C" int mvae_sample_projection_mu0(int kind, const float* v, const float* at, float* z, float* u, int64_t rows,
                                          int64_t at_rows, int d, const float* rp, void* st) {
  if (!v || !at || !z) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_SAMPLE>(kind, v, at, nullptr, z, u, rows, at_rows, d, rp, st);
}
extern "C" int mvae_inverse_sample_projection_mu0(int kind, const float* z, const float* at, float* u, float* v,
                                                  int64_t rows, int64_t at_rows, int d, const float* rp, void* st) {
  if (!z || !at || !u || !v) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_ISAMPLE>(kind, z, at, nullptr, u, v, rows, at_rows, d, rp, st);
}
extern "C" int mvae_logdet(int kind, const float* u, const float* mu, const float* z, float* out, int64_t rows,
                           int64_t at_rows, int d, const float* rp, void* st) {
  if (!out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if ((kind == MVAE_POINCARE || kind == MVAE_PROJ_SPHERE || kind == MVAE_UNIVERSAL) && (!mu || !z))
    return fail(MVAE_E_BADARG, "logdet of a projected model needs mu and z%s", "");
  if ((kind == MVAE_HYPERBOLOID || kind == MVAE_SPHERE) && !u) return fail(MVAE_E_BADARG, "logdet needs u%s", "");
  return launch_prim<OP_LOGDET>(kind, u, mu, z, out, nullptr, rows, at_rows, d, rp, st);
}
extern "C" int mvae_exp_map(int kind, const float* x, const float* at, float* out, int64_t rows, int64_t at_rows,
                            int d, const float* rp, void* st) {
  if (!x || !at || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_EXP>(kind, x, at, nullptr, out, nullptr, rows, at_rows, d, rp, st);
}
extern "C" int mvae_inverse_exp_map(int kind, const float* x, const float* at, float* out, int64_t rows,
                                    int64_t at_rows, int d, const float* rp, void* st) {
  if (!x || !at || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return launch_prim<OP_LOG>(kind, x, at, nullptr, out, nullptr, rows, at_rows, d, rp, st);
}
extern "C" int mvae_geodesic_distance(int kind, int variant, const float* x, const float* y, float* out, int64_t rows,
                                      int64_t y_rows, int d, const float* rp, void* st) {
  if (!x || !y || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (variant == MVAE_DIST_GYRO) return launch_prim<OP_DIST_GYRO>(kind, x, y, nullptr, out, nullptr, rows, y_rows, d, rp, st);
  if (variant != MVAE_DIST_GEODESIC) return fail(MVAE_E_BADARG, "unknown distance variant%s (%lld)", "", variant);
  return launch_prim<OP_DIST>(kind, x, y, nullptr, out, nullptr, rows, y_rows, d, rp, st);
}

extern "C" int mvae_manifold_aux(int op, int kind, const float* x, const float* y, float* out, int64_t rows, int d,
                                 const float* rp, void* st) {
  if (!x || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  switch (op) {
    case OP_LPROD:
      if (!y) return fail(MVAE_E_BADARG, "null pointer%s", "");
      return launch_prim<OP_LPROD>(kind, x, y, nullptr, out, nullptr, rows, rows, d, rp, st);
    case OP_LNORM: return launch_prim<OP_LNORM>(kind, x, nullptr, nullptr, out, nullptr, rows, rows, d, rp, st);
    case OP_TO_BALL:
      if (kind != MVAE_HYPERBOLOID && kind != MVAE_SPHERE) return fail(MVAE_E_BADARG, "TO_BALL is defined for h and s%s", "");
      return launch_prim<OP_TO_BALL>(kind, x, nullptr, nullptr, out, nullptr, rows, rows, d, rp, st);
    case OP_TO_AMBIENT:
      if (kind != MVAE_POINCARE && kind != MVAE_PROJ_SPHERE) return fail(MVAE_E_BADARG, "TO_AMBIENT is defined for p and d%s", "");
      if (d + 1 > MVAE_MAX_TRUE_DIM) return fail(MVAE_E_UNSUPPORTED, "true_dim + 1 > MVAE_MAX_TRUE_DIM%s (%lld)", "", d);
      return launch_prim<OP_TO_AMBIENT>(kind, x, nullptr, nullptr, out, nullptr, rows, rows, d, rp, st);
    case OP_LAMBDA:
      if (kind != MVAE_POINCARE && kind != MVAE_PROJ_SPHERE) return fail(MVAE_E_BADARG, "LAMBDA is defined for p and d%s", "");
      return launch_prim<OP_LAMBDA>(kind, x, nullptr, nullptr, out, nullptr, rows, rows, d, rp, st);
    case OP_MOBADD:
      if (!y) return fail(MVAE_E_BADARG, "null pointer%s", "");
      if (kind != MVAE_POINCARE && kind != MVAE_PROJ_SPHERE) return fail(MVAE_E_BADARG, "MOBADD is defined for p and d%s", "");
      return launch_prim<OP_MOBADD>(kind, x, y, nullptr, out, nullptr, rows, rows, d, rp, st);
  }
  return fail(MVAE_E_BADARG, "unknown auxiliary op%s (%lld)", "", op);
}

extern "C" int mvae_normal_op(int op, const float* a, const float* loc, const float* scale, float* out, int64_t rows,
                              int64_t param_rows, int d, void* st) {
  if (!a || !scale || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  switch (op) {
    case OP_NORMAL_LOGPROB:
      if (!loc) return fail(MVAE_E_BADARG, "null pointer%s", "");
      return launch_prim<OP_NORMAL_LOGPROB>(MVAE_EUCLIDEAN, a, loc, scale, out, nullptr, rows, param_rows, d, nullptr, st);
    case OP_NORMAL_RSAMPLE:
      if (!loc) return fail(MVAE_E_BADARG, "null pointer%s", "");
      return launch_prim<OP_NORMAL_RSAMPLE>(MVAE_EUCLIDEAN, a, loc, scale, out, nullptr, rows, param_rows, d, nullptr, st);
    case OP_NORMAL_KL:
      return launch_prim<OP_NORMAL_KL>(MVAE_EUCLIDEAN, a, nullptr, scale, out, nullptr, rows, rows, d, nullptr, st);
  }
  return fail(MVAE_E_BADARG, "unknown normal op%s (%lld)", "", op);
}

extern "C" int mvae_primitive_backward(int op, int kind, const float* a, const float* b, const float* c3,
                                       const float* g1, const float* g2, float* ga, float* gb, float* gc, float* gr,
                                       int64_t rows, int64_t at_rows, int d, const float* rp, void* st) {
  if (!g1) return fail(MVAE_E_BADARG, "null upstream gradient%s", "");
#define BWD_CASE(O) \
  case O: return launch_prim_bwd<O>(kind, a, b, c3, g1, g2, ga, gb, gc, gr, rows, at_rows, d, rp, st);
  switch (op) {
    BWD_CASE(OP_EXP0) BWD_CASE(OP_LOG0) BWD_CASE(OP_PT0) BWD_CASE(OP_IPT0) BWD_CASE(OP_SAMPLE) BWD_CASE(OP_ISAMPLE)
    BWD_CASE(OP_LOGDET) BWD_CASE(OP_EXP) BWD_CASE(OP_LOG) BWD_CASE(OP_DIST) BWD_CASE(OP_DIST_GYRO)
    BWD_CASE(OP_LPROD) BWD_CASE(OP_LNORM) BWD_CASE(OP_TO_BALL) BWD_CASE(OP_TO_AMBIENT) BWD_CASE(OP_LAMBDA)
    BWD_CASE(OP_MOBADD) BWD_CASE(OP_NORMAL_LOGPROB) BWD_CASE(OP_NORMAL_RSAMPLE) BWD_CASE(OP_NORMAL_KL)
  }
#undef BWD_CASE
  return fail(MVAE_E_BADARG, "unknown primitive op%s (%lld)", "", op);
}

// ------------------------------------------------------------------------------------------------ scalar functions (API)
// The reference's guarded scalar functions (ops/common.py:28-147) and their custom derivative rules, evaluated by the
// DEVICE code the manifold kernels use (value through the float overloads, derivative through the Dual rules), so the
// rules can be pinned directly against vectors recorded from the reference.
__global__ __launch_bounds__(256) void k_scalar_fn(int fn, const float* x, float* y, float* dy, int64_t n, float lo,
                                                   float hi) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i];
    const Dual xd = Dual{xv, 1.f};
    float v;
    Dual r;
    switch (fn) {
      case MVAE_FN_CLAMP: v = leaky_clamp(xv, lo, hi); r = leaky_clamp(xd, lo, hi); break;
      case MVAE_FN_ATANH: v = g_atanh(xv); r = g_atanh(xd); break;
      case MVAE_FN_ACOSH: v = g_acosh(xv); r = g_acosh(xd); break;
      case MVAE_FN_COSH: v = g_cosh(xv); r = g_cosh(xd); break;
      case MVAE_FN_SINH: v = g_sinh(xv); r = g_sinh(xd); break;
      case MVAE_FN_SQRT: v = g_sqrt(xv); r = g_sqrt(xd); break;
      case MVAE_FN_LOGSINH: v = g_logsinh(xv); r = g_logsinh(xd); break;
      case MVAE_FN_LOGCOSH: v = g_logcosh(xv); r = g_logcosh(xd); break;
      case MVAE_FN_COSH_SINH_PAIR: {  // the shared-exp pair the manifolds use: y = cosh, dy = sinh (values)
        float c, s_;
        g_cosh_sinh(xv, &c, &s_);
        v = c;
        r = Dual{c, s_};
      } break;
      case MVAE_FN_COS_SIN_PAIR: {  // y = cos, dy = sin (values)
        float c, s_;
        t_cos_sin(xv, &c, &s_);
        v = c;
        r = Dual{c, s_};
      } break;
      case MVAE_FN_SOFTPLUS: v = t_softplus(xv); r = t_softplus(xd); break;
      case MVAE_FN_ACOS: v = t_acos(xv); r = t_acos(xd); break;
      case MVAE_FN_TAN: v = t_tan(xv); r = t_tan(xd); break;
      case MVAE_FN_LOG1P_POS: v = mvf::log1p_pos(xv); r = Dual{v, 1.0f / (1.0f + xv)}; break;
      case MVAE_FN_EXP: v = t_exp(xv); r = t_exp(xd); break;
      case MVAE_FN_LOG: v = t_log(xv); r = t_log(xd); break;
      case MVAE_FN_STD: v = t_softplus(xv) + 1e-5f; r = t_softplus(xd) + 1e-5f; break;  // component.py:72
      default: v = NAN; r = Dual{NAN, NAN}; break;
    }
    y[i] = v;
    if (dy) dy[i] = r.d;
  }
}

extern "C" int mvae_scalar_fn(int fn, const float* x, float* y, float* dy, int64_t n, float lo, float hi, void* st) {
  if (!x || !y || n < 0) return fail(MVAE_E_BADARG, "null pointer / bad size%s", "");
  if (fn < 0 || fn >= MVAE_FN_COUNT) return fail(MVAE_E_BADARG, "unknown scalar function%s (%lld)", "", fn);
  if (n == 0) return 0;
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_scalar_fn, dim3(grid), dim3(256), 0, (hipStream_t)st, fn, x, y, dy, n, lo, hi);
  LAUNCH_CHECK("scalar function launch");
  return 0;
}

__global__ __launch_bounds__(256) void k_mul(const float* a, const float* b, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] * b[i];
}
// out[r][j] = g[r][j] * s[r]
__global__ __launch_bounds__(256) void k_scale_rows(const float* g, const float* sc, float* out, int64_t rows, int D) {
  const int64_t n = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = g[i] * sc[i / D];
}
extern "C" int mvae_scale_rows(const float* g, const float* sc, float* out, int64_t rows, int D, void* st) {
  if (!g || !sc || !out || rows < 0 || D < 1) return fail(MVAE_E_BADARG, "null pointer / bad size%s", "");
  if (rows == 0) return 0;
  int grid = (int)((rows * D + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_scale_rows, dim3(grid), dim3(256), 0, (hipStream_t)st, g, sc, out, rows, D);
  LAUNCH_CHECK("scale rows launch");
  return 0;
}
extern "C" int mvae_mul(const float* a, const float* b, float* out, int64_t n, void* st) {
  if (!a || !b || !out || n < 0) return fail(MVAE_E_BADARG, "null pointer / bad size%s", "");
  if (n == 0) return 0;
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_mul, dim3(grid), dim3(256), 0, (hipStream_t)st, a, b, out, n);
  LAUNCH_CHECK("mul launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ component kernels (API)
static bool no_coop_env() {
  const char* nc = getenv("MVAE_NO_COOP");
  return nc && nc[0] && nc[0] != '0';
}
template <int DMAX>
__global__ __launch_bounds__(256) void k_comp_fwd(CompTable t, const float* heads, int heads_ld, const float* eps,
                                                  int eps_ld, const float* radii, float* z, int z_ld, float* kl,
                                                  float* lq, float* lp, float* mu, float* sd, int64_t rows,
                                                  int64_t head_rows) {
  const int64_t items = rows * t.n;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t r = it % rows;
    const int ci = (int)(it / rows);
    const int64_t hr = r % head_rows;
    const bool first = r < head_rows;
    comp_fwd_row<DMAX>(t.c[ci], heads + hr * heads_ld, eps + r * eps_ld, radii, z + r * z_ld, nullptr,
                       kl ? kl + (int64_t)ci * rows + r : nullptr, lq ? lq + (int64_t)ci * rows + r : nullptr,
                       lp ? lp + (int64_t)ci * rows + r : nullptr, (mu && first) ? mu + hr * z_ld : nullptr,
                       (sd && first) ? sd + hr * eps_ld : nullptr);
  }
}

template <int DMAX>
__global__ __launch_bounds__(256) void k_comp_bwd(CompTable t, const float* heads, int heads_ld, const float* eps,
                                                  int eps_ld, const float* radii, const float* dz, int z_ld,
                                                  const float* dkl, float dkl_scalar, float* dheads, float* drad_rows,
                                                  int64_t rows) {
  const int64_t items = rows * t.total_dirs;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t r = it / t.total_dirs;
    const int gd = (int)(it % t.total_dirs);
    int ci = 0;
    while (gd >= t.dir_off[ci + 1]) ++ci;
    const int dir = gd - t.dir_off[ci];
    const mvae_component_desc& c = t.c[ci];
    const float w = dkl ? dkl[(int64_t)ci * rows + r] : dkl_scalar;
    float g = comp_bwd_dir<DMAX>(c, heads + r * heads_ld, eps + r * eps_ld, radii, dz + r * z_ld, w, dir);
    if (dir < c.true_dim) dheads[r * heads_ld + c.mean_col + dir] = g;
    else if (dir < c.true_dim + c.logvar_dim) dheads[r * heads_ld + c.logvar_col + (dir - c.true_dim)] = g;
    else drad_rows[(int64_t)c.radius_idx * rows + r] = g;  // per-row term; summed in row order by k_rowsum_fixed
  }
}

// The same two operators for LARGE true dimensions (d >= 9: `h40`, `p40`, ...) in the wave-cooperative form of mvae_coop.hpp: one
// WAVE per (row, component) forward, per (row, component, input direction) backward, lane = vector entry, no scratch memory.
__global__ __launch_bounds__(256) void k_comp_fwd_coop(CompTable t, const float* heads, int heads_ld, const float* eps,
                                                       int eps_ld, const float* radii, float* z, int z_ld, float* kl,
                                                       float* lq, float* lp, float* mu, float* sd, int64_t rows,
                                                       int64_t head_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (it >= rows * t.n) return;
  const int64_t r = it % rows;
  const int ci = (int)(it / rows);
  const int64_t hr = r % head_rows;
  const bool first = r < head_rows;
  const mvae_component_desc c = t.c[ci];
  const int d = c.true_dim, lvd = c.logvar_dim, j = lane - 1;
  const bool act = lane >= 1 && lane <= d;
  const float* hrow = heads + hr * heads_ld;
  const float m = act ? hrow[c.mean_col + j] : 0.f;
  const float l = act ? hrow[c.logvar_col + (lvd == 1 ? 0 : j)] : 0.f;
  const float e = act ? eps[r * eps_ld + c.eps_col + j] : 0.f;
  const float rp = c.kind == kEuclidean ? 0.f : radii[c.radius_idx];
  float zl = 0.f, klv = 0.f;
  CoopExtra<float> ex{0.f, 0.f, 0.f, 0.f};
  (void)coop_eval<float>(c.kind, m, l, e, rp, d, lane, &zl, &klv, &ex);
  const int A = ambient_dim(c.kind, d), idx = coop_z_shifted(c.kind) ? j : lane;
  if (idx >= 0 && idx < A) {
    z[r * z_ld + c.z_col + idx] = zl;
    if (mu && first) mu[hr * z_ld + c.z_col + idx] = ex.mu;
  }
  if (sd && first && act && j < lvd) sd[hr * eps_ld + c.eps_col + j] = ex.sg;
  if (lane == 0) {
    if (kl) kl[(int64_t)ci * rows + r] = klv;
    if (lq) {
      lq[(int64_t)ci * rows + r] = ex.lq;
      lp[(int64_t)ci * rows + r] = ex.lp;
    }
  }
}

__global__ __launch_bounds__(256) void k_comp_bwd_coop(CompTable t, const float* heads, int heads_ld, const float* eps,
                                                       int eps_ld, const float* radii, const float* dz, int z_ld,
                                                       const float* dkl, float dkl_scalar, float* dheads,
                                                       float* drad_rows, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (it >= rows * t.total_dirs) return;
  const int64_t r = it / t.total_dirs;
  const int gd = (int)(it % t.total_dirs);
  int ci = 0;
  while (gd >= t.dir_off[ci + 1]) ++ci;
  const int dir = gd - t.dir_off[ci];
  const mvae_component_desc c = t.c[ci];
  const int d = c.true_dim, lvd = c.logvar_dim, j = lane - 1;
  const bool act = lane >= 1 && lane <= d;
  const float* hrow = heads + r * heads_ld;
  const float mv_ = hrow[c.mean_col + (act ? j : 0)];
  const float lv_ = hrow[c.logvar_col + ((act && lvd != 1) ? j : 0)];
  const float ev_ = eps[r * eps_ld + c.eps_col + (act ? j : 0)];
  const float rv_ = c.kind == kEuclidean ? 0.f : radii[c.radius_idx];
  const Dual m{act ? mv_ : 0.f, (act && dir == j) ? 1.f : 0.f};
  const Dual l{act ? lv_ : 0.f, (act && (lvd == 1 ? dir == d : dir == d + j)) ? 1.f : 0.f};
  const Dual rp{rv_, dir == d + lvd ? 1.f : 0.f};
  Dual zl{0.f, 0.f}, klv{0.f, 0.f};
  (void)coop_eval<Dual>(c.kind, m, l, act ? ev_ : 0.f, rp, d, lane, &zl, &klv);
  const int A = ambient_dim(c.kind, d), idx = coop_z_shifted(c.kind) ? j : lane;
  const float dzl = (idx >= 0 && idx < A) ? dz[r * z_ld + c.z_col + idx] : 0.f;
  const float w = dkl ? dkl[(int64_t)ci * rows + r] : dkl_scalar;
  const float g = w * klv.d + wave_sum(dzl * zl.d);
  if (lane == 0) {
    if (dir < d) dheads[r * heads_ld + c.mean_col + dir] = g;
    else if (dir < d + lvd) dheads[r * heads_ld + c.logvar_col + (dir - d)] = g;
    else drad_rows[(int64_t)c.radius_idx * rows + r] = g;
  }
}

// out[i] = sum_r part[i][r], one workgroup per i, fixed order (thread t adds r = t, t+256, ...; wave sums by DPP; the four
// wave totals in wave order): the radius gradient of the standalone component backward, bit-reproducible.
__global__ __launch_bounds__(256) void k_rowsum_fixed(const float* part, float* out, int64_t rows) {
  __shared__ float sm[4];
  const float* p = part + (int64_t)blockIdx.x * rows;
  float s = 0.f;
  for (int64_t r = threadIdx.x; r < rows; r += 256) s += p[r];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

extern "C" int mvae_component_forward(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld,
                                      const float* eps, int eps_ld, const float* radii, float* z, int z_ld, float* kl,
                                      float* log_q, float* log_p, float* mu, float* sd, int64_t rows,
                                      int64_t head_rows, void* stream) {
  if (!heads || !eps || !z || rows < 0 || head_rows < 1) return fail(MVAE_E_BADARG, "null pointer / bad rows%s", "");
  if ((log_q == nullptr) != (log_p == nullptr)) return fail(MVAE_E_BADARG, "log_q and log_p go together%s", "");
  CompTable t;
  int dmax;
  unsigned char all[kMaxComp];
  memset(all, 1, sizeof(all));
  int rc = fill_table(&t, comps, ncomp, all, &dmax);
  if (rc) return rc;
  for (int i = 0; i < ncomp; ++i)
    if (comps[i].kind != MVAE_EUCLIDEAN && !radii) return fail(MVAE_E_BADARG, "radii is NULL%s", "");
  if (rows == 0) return 0;
  int grid = (int)((rows * ncomp + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (bucket_of(dmax) > 8 && coop_eligible(t) && !no_coop_env()) {
    const int64_t waves = rows * ncomp;
    hipLaunchKernelGGL(k_comp_fwd_coop, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld,
                       radii, z, z_ld, kl, log_q, log_p, mu, sd, rows, head_rows);
    LAUNCH_CHECK("cooperative component forward launch");
    return 0;
  }
  DMAX_SWITCH(dmax, hipLaunchKernelGGL((k_comp_fwd<DM>), dim3(grid), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld,
                                       radii, z, z_ld, kl, log_q, log_p, mu, sd, rows, head_rows));
  LAUNCH_CHECK("component forward launch");
  return 0;
}

extern "C" int64_t mvae_component_backward_workspace_floats(int ncomp, int64_t rows) {
  return (ncomp < 0 || rows < 0) ? -1 : (int64_t)ncomp * rows;
}

extern "C" int mvae_component_backward(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld,
                                       const float* eps, int eps_ld, const float* radii, const float* dz, int z_ld,
                                       const float* dkl, float dkl_scalar, float* dheads, float* dradii,
                                       float* workspace, int64_t rows, void* stream) {
  if (!heads || !eps || !dz || !dheads || rows < 0) return fail(MVAE_E_BADARG, "null pointer / bad rows%s", "");
  if (dradii && !workspace) return fail(MVAE_E_BADARG, "dradii needs the [ncomp, rows] workspace%s", "");
  CompTable t;
  int dmax;
  unsigned char tr[kMaxComp];
  memset(tr, dradii ? 1 : 0, sizeof(tr));
  int rc = fill_table(&t, comps, ncomp, tr, &dmax);
  if (rc) return rc;
  for (int i = 0; i < ncomp; ++i)
    if (comps[i].radius_idx < 0 || comps[i].radius_idx >= ncomp)
      return fail(MVAE_E_BADARG, "radius_idx out of range%s (%lld)", "", comps[i].radius_idx);
  if (rows == 0) return 0;
  int grid = (int)((rows * t.total_dirs + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (dradii) {
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)ncomp * (size_t)rows, s);  // Euclidean rows stay 0
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
  }
  if (bucket_of(dmax) > 8 && coop_eligible(t) && !no_coop_env()) {
    const int64_t waves = rows * t.total_dirs;
    hipLaunchKernelGGL(k_comp_bwd_coop, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld,
                       radii, dz, z_ld, dkl, dkl_scalar, dheads, workspace, rows);
  } else {
    DMAX_SWITCH(dmax, hipLaunchKernelGGL((k_comp_bwd<DM>), dim3(grid), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld,
                                         radii, dz, z_ld, dkl, dkl_scalar, dheads, workspace, rows));
  }
  if (dradii) hipLaunchKernelGGL(k_rowsum_fixed, dim3(ncomp), dim3(256), 0, s, workspace, dradii, rows);
  LAUNCH_CHECK("component backward launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ component kernels, float64 chain
// run.py:77,98-101: the reference's CLI default computes in float64.  These are mvae_component_forward / _backward with every
// intermediate of the latent chain in float64 (mvae_math.hpp "float64 number types"); heads, eps, radii, z, kl, the
// gradients stay float32 tensors, i.e. the dense layers on either side are float32.  What it buys: in float32 the sphere's
// alpha = <mu, z> / R^2 and the hyperboloid's <mu, z>_L / R^2 lose 3-4 digits at the warm-up radii and the
// acos derivative needs its cap; here neither happens.  True dimensions <= 8 (the per-thread templates; no cooperative form).
template <int DMAX>
__global__ __launch_bounds__(256) void k_comp_fwd64(CompTable t, const float* heads, int heads_ld, const float* eps, int eps_ld,
                                                    const float* radii, float* z, int z_ld, float* kl, float* lq, float* lp,
                                                    float* mu, float* sd, int64_t rows, int64_t head_rows) {
  const int64_t items = rows * t.n;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t r = it % rows;
    const int ci = (int)(it / rows);
    const int64_t hr = r % head_rows;
    const bool first = r < head_rows;
    comp_fwd_row64<DMAX>(t.c[ci], heads + hr * heads_ld, eps + r * eps_ld, radii, z + r * z_ld,
                         kl ? kl + (int64_t)ci * rows + r : nullptr, lq ? lq + (int64_t)ci * rows + r : nullptr,
                         lp ? lp + (int64_t)ci * rows + r : nullptr, (mu && first) ? mu + hr * z_ld : nullptr,
                         (sd && first) ? sd + hr * eps_ld : nullptr);
  }
}
template <int DMAX>
__global__ __launch_bounds__(256) void k_comp_bwd64(CompTable t, const float* heads, int heads_ld, const float* eps, int eps_ld,
                                                    const float* radii, const float* dz, int z_ld, const float* dkl,
                                                    float dkl_scalar, float* dheads, float* drad_rows, int64_t rows) {
  const int64_t items = rows * t.total_dirs;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t r = it / t.total_dirs;
    const int gd = (int)(it % t.total_dirs);
    int ci = 0;
    while (gd >= t.dir_off[ci + 1]) ++ci;
    const int dir = gd - t.dir_off[ci];
    const mvae_component_desc& c = t.c[ci];
    const float w = dkl ? dkl[(int64_t)ci * rows + r] : dkl_scalar;
    float g = comp_bwd_dir64<DMAX>(c, heads + r * heads_ld, eps + r * eps_ld, radii, dz + r * z_ld, w, dir);
    if (dir < c.true_dim) dheads[r * heads_ld + c.mean_col + dir] = g;
    else if (dir < c.true_dim + c.logvar_dim) dheads[r * heads_ld + c.logvar_col + (dir - c.true_dim)] = g;
    else drad_rows[(int64_t)c.radius_idx * rows + r] = g;
  }
}
#define DMAX_SWITCH64(dmax, ...)                                  \
  if (bucket_of(dmax) <= 2) { constexpr int DM = 2; __VA_ARGS__; } \
  else if (bucket_of(dmax) <= 4) { constexpr int DM = 4; __VA_ARGS__; } \
  else { constexpr int DM = 8; __VA_ARGS__; }

extern "C" int mvae_component_forward_f64(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld,
                                          const float* eps, int eps_ld, const float* radii, float* z, int z_ld, float* kl,
                                          float* log_q, float* log_p, float* mu, float* sd, int64_t rows, int64_t head_rows,
                                          void* stream) {
  if (!heads || !eps || !z || rows < 0 || head_rows < 1) return fail(MVAE_E_BADARG, "null pointer / bad rows%s", "");
  if ((log_q == nullptr) != (log_p == nullptr)) return fail(MVAE_E_BADARG, "log_q and log_p go together%s", "");
  CompTable t;
  int dmax;
  unsigned char all[kMaxComp];
  memset(all, 1, sizeof(all));
  int rc = fill_table(&t, comps, ncomp, all, &dmax);
  if (rc) return rc;
  if (bucket_of(dmax) > 8) return fail(MVAE_E_UNSUPPORTED, "float64 component chain: true dimensions <= 8%s", "");
  for (int i = 0; i < ncomp; ++i)
    if (comps[i].kind != MVAE_EUCLIDEAN && !radii) return fail(MVAE_E_BADARG, "radii is NULL%s", "");
  if (rows == 0) return 0;
  int grid = (int)((rows * ncomp + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  DMAX_SWITCH64(dmax, hipLaunchKernelGGL((k_comp_fwd64<DM>), dim3(grid), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld, radii,
                                         z, z_ld, kl, log_q, log_p, mu, sd, rows, head_rows));
  LAUNCH_CHECK("float64 component forward launch");
  return 0;
}

extern "C" int mvae_component_backward_f64(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld,
                                           const float* eps, int eps_ld, const float* radii, const float* dz, int z_ld,
                                           const float* dkl, float dkl_scalar, float* dheads, float* dradii, float* workspace,
                                           int64_t rows, void* stream) {
  if (!heads || !eps || !dz || !dheads || rows < 0) return fail(MVAE_E_BADARG, "null pointer / bad rows%s", "");
  if (dradii && !workspace) return fail(MVAE_E_BADARG, "dradii needs the [ncomp, rows] workspace%s", "");
  CompTable t;
  int dmax;
  unsigned char tr[kMaxComp];
  memset(tr, dradii ? 1 : 0, sizeof(tr));
  int rc = fill_table(&t, comps, ncomp, tr, &dmax);
  if (rc) return rc;
  if (bucket_of(dmax) > 8) return fail(MVAE_E_UNSUPPORTED, "float64 component chain: true dimensions <= 8%s", "");
  for (int i = 0; i < ncomp; ++i)
    if (comps[i].radius_idx < 0 || comps[i].radius_idx >= ncomp)
      return fail(MVAE_E_BADARG, "radius_idx out of range%s (%lld)", "", comps[i].radius_idx);
  if (rows == 0) return 0;
  int grid = (int)((rows * t.total_dirs + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (dradii) {
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)ncomp * (size_t)rows, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
  }
  DMAX_SWITCH64(dmax, hipLaunchKernelGGL((k_comp_bwd64<DM>), dim3(grid), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld, radii,
                                         dz, z_ld, dkl, dkl_scalar, dheads, workspace, rows));
  if (dradii) hipLaunchKernelGGL(k_rowsum_fixed, dim3(ncomp), dim3(256), 0, s, workspace, dradii, rows);
  LAUNCH_CHECK("float64 component backward launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ dense layers (API)

extern "C" int mvae_linear_forward(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K,
                                   int relu, void* stream) {
  if (!x || !W || !y || M < 0 || N < 1 || K < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (M == 0) return 0;
  if (M > (1 << 20) * 16) return fail(MVAE_E_UNSUPPORTED, "M too large%s", "");
  if (M >= kTiledMinRows && linear_forward_tiled(x, W, b, y, M, N, K, relu, (hipStream_t)stream)) {
    LAUNCH_CHECK("tiled linear forward launch");
    return 0;
  }
  dim3 grid((N + 15) / 16, (unsigned)((M + 15) / 16));
  hipStream_t s = (hipStream_t)stream;
  if (relu) hipLaunchKernelGGL(k_linear_fwd<true>, grid, dim3(256), 0, s, x, W, b, y, (int)M, N, K);
  else hipLaunchKernelGGL(k_linear_fwd<false>, grid, dim3(256), 0, s, x, W, b, y, (int)M, N, K);
  LAUNCH_CHECK("linear forward launch");
  return 0;
}

extern "C" int mvae_linear_backward(const float* x, const float* W, const float* dy, int relu_in, float* dW, float* db,
                                    float* dx, int64_t M, int N, int K, void* stream) {
  if (!x || !W || !dy || !dW || !db || M < 1 || N < 1 || K < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (N <= kSknN && K >= 1024 && (K & 63) == 0 && M >= 64 && M <= 0x7fffffff &&
      ((((uintptr_t)x) | ((uintptr_t)W) | ((uintptr_t)dW) | ((uintptr_t)dx)) & 15) == 0) {
#define MV_SKN(NN_)                                                                                                  \
  hipLaunchKernelGGL(k_linear_bwd_skn<NN_>, dim3(K / 32 + 1), dim3(256), 0, (hipStream_t)stream, x, W, dy, dW, db, dx, \
                     (int)M, N, K, relu_in)
    if (N <= 4) MV_SKN(4); else if (N <= 8) MV_SKN(8); else if (N <= 12) MV_SKN(12); else MV_SKN(16);
#undef MV_SKN
    LAUNCH_CHECK("skinny linear backward launch");
    return 0;
  }
  const int ntN = (N + 15) / 16, ntK = (K + 15) / 16, ntM = (int)((M + 15) / 16);
  const int n_dx = dx ? ntM * ntK : 0, n_dw = ntN * ntK, n_db = (N + kColsPerBlock - 1) / kColsPerBlock;
  hipLaunchKernelGGL(k_linear_bwd, dim3(n_dx + n_dw + n_db), dim3(256), 0, (hipStream_t)stream, x, W, dy, dW, db, dx,
                     (int)M, N, K, relu_in, n_dw, n_dx);
  LAUNCH_CHECK("linear backward launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ log-likelihood helpers (API)
// bce[r] = sum_j BCE-with-logits(logits[r][j], x[r % x_rows][j]); one wavefront per row.
__global__ __launch_bounds__(256) void k_bce_rows(const float* logits, const float* x, float* out, int64_t rows,
                                                  int64_t x_rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* lr = logits + r * D;
  const float* xr = x + (r % x_rows) * D;
  float s = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float y = lr[j], t = xr[j];
    const float e = expf(-fabsf(y));
    s += (1.f - t) * y - (fminf(y, 0.f) - mvf::log1p_pos(e));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) out[r] = s;
}

// log p(x)[b] = logsumexp_n(-bce[n][b] + log_p[n][b] - log_q[n][b]) - log n ;
// mi[b] = logsumexp_n(log_q[n][b] - log_p[n][b]) - log n        (vae.py:113-117)
// One workgroup per batch column b; thread t takes the samples t, t + 256, ...; block max, then block sum of exp(. - max)
// (wave sums by DPP, the four wave totals added in wave order).
__global__ __launch_bounds__(256) void k_loglik_reduce(const float* bce, const float* log_p, const float* log_q,
                                                       float* log_px, float* mi, int n, int B) {
  __shared__ float sm[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float m1 = -INFINITY, m2 = -INFINITY;
  for (int i = tid; i < n; i += 256) {
    const size_t o = (size_t)i * B + b;
    const float lp = log_p[o], lq = log_q[o];
    m1 = fmaxf(m1, -bce[o] + lp - lq);
    m2 = fmaxf(m2, lq - lp);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    m1 = fmaxf(m1, __shfl_xor(m1, off));
    m2 = fmaxf(m2, __shfl_xor(m2, off));
  }
  if ((tid & 63) == 0) {
    sm[0][tid >> 6] = m1;
    sm[1][tid >> 6] = m2;
  }
  __syncthreads();
  m1 = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
  m2 = fmaxf(fmaxf(sm[1][0], sm[1][1]), fmaxf(sm[1][2], sm[1][3]));
  // torch.logsumexp subtracts 0 instead of an infinite maximum: a column of -inf terms gives -inf, not exp(-inf + inf) = NaN
  m1 = isinf(m1) ? 0.f : m1;
  m2 = isinf(m2) ? 0.f : m2;
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  for (int i = tid; i < n; i += 256) {
    const size_t o = (size_t)i * B + b;
    const float lp = log_p[o], lq = log_q[o];
    s1 += expf((-bce[o] + lp - lq) - m1);
    s2 += expf((lq - lp) - m2);
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) {
    sm[0][tid >> 6] = s1;
    sm[1][tid >> 6] = s2;
  }
  __syncthreads();
  if (tid == 0) {
    const float ln = logf((float)n);
    log_px[b] = m1 + logf((sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3])) - ln;
    mi[b] = m2 + logf((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3])) - ln;
  }
}

extern "C" int mvae_bce_rows(const float* logits, const float* x, float* out, int64_t rows, int64_t x_rows, int D,
                             void* stream) {
  if (!logits || !x || !out || rows < 0 || x_rows < 1 || D < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_bce_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, x, out,
                     rows, x_rows, D);
  LAUNCH_CHECK("bce rows launch");
  return 0;
}

// ---- decoder + per-row BCE of the log-likelihood estimator in ONE launch (vae.py:98-109 for the MLP decoder, ffnn_vae.py:52-60):
//   out[r] = sum_j BCE-with-logits( (relu(z[r] W_d0^T + b_d0) W_l^T + b_l)[j], x[r % x_rows][j] ).
// n * B = 64 000 rows for the reference's n = 500: as three launches the hidden layer (102 MB) and the logits (200 MB) went
// to HBM and back, and the logits contraction ran on the generic LDS-tiled kernel.  Here neither exists in memory:
//   * workgroup = 64 rows, wave = 16 rows; the wave's hidden-layer block [16][H] is computed ONCE (K = z_dim: VALU FMAs from
//     LDS copies of W_d0 / b_d0) straight into the A fragments of the f32-input MFMA and stays in registers (H / 4 per lane);
//   * the logits are produced 16 columns at a time: W_l's row block [16][H] -- 16 H contiguous floats -- goes to LDS by LDS-DMA
//     (double buffered, requested one column tile ahead, one barrier per tile) and is shared by the four waves; H / 4 MFMAs
//     per wave and tile on four independent accumulators; the epilogue adds the bias, takes the targets from x
//     (L2-resident) and adds the BCE terms to four per-lane row sums; a DPP row sum at the very end.
// Next to the 8-pass f32 MFMA the VALU instructions of BOTH waves of a SIMD are step time (mvae_f32pp.hip), so the loop is
// written for few of them: LDS-DMA instead of load + ds_write, scalar bases + 32-bit byte offsets, bare v_exp_f32 / v_log_f32.
// First version (register-staged tiles, expf / log1p, 125 VALU per 100 MFMAs): 414 us = 62 % of the f32 MFMA peak for the
// reference's shape; this one (~45 VALU): 352-356 us = 72-73 % (MFMA busy cycles / SIMD cycles by the counters: 0.73 -> see
// DESIGN.md).  -DMV_DBR_NOEPI (tools/build_variant.py): the loop without the BCE terms, for A/B timing.
#ifndef MV_DBR_PF
#define MV_DBR_PF 1
#endif
template <int NCH, int ZP>  // H = 16 NCH ; z_dim in slices of ZP columns
__global__ __launch_bounds__(256, 2) void k_decode_bce_rows(const float* z, int64_t rows, int Z, const float* Wd0,
                                                            const float* bd0, const float* Wl, const float* bl,
                                                            const float* x, int64_t x_rows, int D, float* out) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  constexpr int H = 16 * NCH;
  float* bt = dyn;                 // [2][16][H]: W_l row blocks (contiguous in memory and here: NCH pieces of 1 KB)
  float* wd_s = dyn + 2 * 16 * H;  // [H][ZP] (zero past z_dim)
  float* bd_s = wd_s + H * ZP;     // [H]
  // (two workgroups per CU.  Three -- the W_d0 copy overlaid on the second buffer, 51 KB each -- measured 357.7 us against 352:
  // 1000 workgroups are 1.95 rounds of 512 slots but 1.3 rounds of 768)
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (in an SGPR: piece addresses and their guards are scalar)
  const int64_t r0 = (int64_t)blockIdx.x * 64 + wave * 16;
  const int ntD = D >> 4;
  // W_l's row block nt -> LDS buffer `buf` by LDS-DMA: wave w moves the 1 KB pieces w, w + 4, ... -- no staging registers, no
  // ds_write, and no VALU: scalar base + the lane's fixed byte offset (next to the 8-pass f32 MFMA every VALU instruction of
  // either wave of the SIMD is step time: the counters of the first version showed 125 of them per 100 MFMAs)
  // (addresses as scalar base + zero-extended 32-bit BYTE offset: the form the global instructions take an SGPR base for)
  const unsigned lane16 = lane * 16;
  auto request = [&](int nt, int buf) __attribute__((always_inline)) {
    const char* src = reinterpret_cast<const char*>(Wl + (size_t)nt * (16 * H) + wave * 256);  // (scalar)
#pragma unroll
    for (int u = 0; u < (NCH + 3) / 4; ++u)
      if (wave + 4 * u < NCH)  // (scalar branch)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + u * 4096 + lane16),
                                         (__attribute__((address_space(3))) void*)(bt + buf * (16 * H) + (wave + 4 * u) * 256), 16, 0, 0);
  };
  request(0, 0);
  for (int e = tid; e < H; e += 256) bd_s[e] = bd0[e];
  // hidden layer of this wave's 16 rows, as A fragments: a[c][t] = relu(b_d0[k] + <z[row i], W_d0[k]>), k = 16 c + 4 q + t.
  // K = z_dim in slices of ZP columns (one slice for z_dim <= ZP; up to four of 16 for the many-component models): per slice
  // the ZP columns of W_d0 are staged in LDS (zero past z_dim) and this lane's ZP z values sit in registers.
  f32x4 a[NCH];
  const int64_t zrow = (r0 + i < rows ? r0 + i : rows - 1) * Z;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = bd_s[16 * c + 4 * q + t];
      asm volatile("" : "+v"(v));
      a[c][t] = v;
    }
  for (int s0 = 0; s0 < Z; s0 += ZP) {
    if (s0 > 0) __syncthreads();  // the previous slice has been consumed
    float zr[ZP];
#pragma unroll
    for (int j = 0; j < ZP; ++j) {
      const float v = z[zrow + (s0 + j < Z ? s0 + j : 0)];
      zr[j] = s0 + j < Z ? v : 0.f;
    }
    // (eight requests in flight per thread: as a plain loop every LDS store waited for its own load's round trip -- 13 to 25
    // of them per slice)
    for (int e0 = tid; e0 < H * ZP; e0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u < H * ZP ? e0 + 256 * u : 0;
        const int k = e / ZP, j = e - k * ZP;
        v[u] = Wd0[(size_t)k * Z + (s0 + j < Z ? s0 + j : 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u;
        if (e < H * ZP) wd_s[e] = s0 + (e % ZP) < Z ? v[u] : 0.f;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (first slice: the DMA is invisible to the compiler's LDS dependence tracking)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = 16 * c + 4 * q + t;
        float v = a[c][t];
#pragma unroll
        for (int j4 = 0; j4 < ZP; j4 += 4) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wd_s + k * ZP + j4);
          v = fmaf(zr[j4], w[0], v);
          v = fmaf(zr[j4 + 1], w[1], v);
          v = fmaf(zr[j4 + 2], w[2], v);
          v = fmaf(zr[j4 + 3], w[3], v);
        }
        // (opaque to the SLP vectorizer: left alone it packs this phase into wide vectors whose shuffles spill ~700 registers,
        // and the loop's pointers come back from scratch INSIDE the loop, behind the piece requests: 367 -> 616 us)
        asm volatile("" : "+v"(v));
        a[c][t] = v;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = a[c][t];
      v = v < 0.f ? 0.f : v;  // torch.relu: NaN propagates
      asm volatile("" : "+v"(v));
      a[c][t] = v;
    }
  // targets' rows (x broadcast over the samples) of this lane's four output rows
  unsigned xo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t rr = r0 + 4 * q + r < rows ? r0 + 4 * q + r : rows - 1;
    xo[r] = (unsigned)((rr % x_rows) * D + i) * 4u;  // bytes
  }
  const unsigned i4 = i * 4;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int nt = 0; nt < ntD; ++nt) {
    const int buf = nt & 1;
    // (the tile index through readfirstlane -- opaque and uniform: otherwise loop strength reduction turns every address below into a loop-carried 64-bit
    // VGPR pointer with its own VALU increment per tile -- 12 of the loop's 51 VALU instructions)
    const int nts = __builtin_amdgcn_readfirstlane(nt);
    if (nt + 1 < ntD) request(nts + 1, buf ^ 1);  // (uniform)
    const char* blt = reinterpret_cast<const char*>(bl + nts * 16);  // (scalar bases)
    const char* xt = reinterpret_cast<const char*>(x + nts * 16);
    const float bias = *reinterpret_cast<const float*>(blt + i4);
    float tv[4];  // targets: requested here, first USED in the epilogue (a use up here would wait for the piece requests too)
#pragma unroll
    for (int r = 0; r < 4; ++r) tv[r] = *reinterpret_cast<const float*>(xt + xo[r]);
    // four independent chains (two -- a dependent MFMA 64 cycles after its predecessor -- measured slower: 337 -> 356 us)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const float* brow = bt + buf * (16 * H) + i * H + 4 * q;
    // B fragments PF chunks ahead of the MFMAs that use them
    constexpr int PF = MV_DBR_PF;
    f32x4 bq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) bq[u] = *reinterpret_cast<const f32x4*>(brow + 16 * (u < NCH ? u : 0));
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += PF) {
      f32x4 bn[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) bn[u] = *reinterpret_cast<const f32x4*>(brow + 16 * (c0 + PF + u < NCH ? c0 + PF + u : 0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (c0 + u < NCH) {
          acc0 = mfma16(a[c0 + u][0], bq[u][0], acc0);
          acc1 = mfma16(a[c0 + u][1], bq[u][1], acc1);
          acc2 = mfma16(a[c0 + u][2], bq[u][2], acc2);
          acc3 = mfma16(a[c0 + u][3], bq[u][3], acc3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < PF; ++u) bq[u] = bn[u];
    }
    const f32x4 acc = (acc0 + acc1) + (acc2 + acc3);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // F.binary_cross_entropy_with_logits: (1 - t) y + softplus(-y) = (1 - t) y - min(y, 0) + log(1 + exp(-|y|)), the
      // two transcendentals as bare v_exp_f32 / v_log_f32 (their arguments need no range handling here; |error| per term
      // ~1e-7 against a row sum of several hundred)
      const float y = acc[r] + bias;
#ifdef MV_DBR_NOEPI
      rs[r] += (1.f - tv[r]) * y;
#else
      const float e = __builtin_amdgcn_exp2f(fabsf(y) * -1.4426950408889634f);      // exp(-|y|) in (0, 1]
      const float l2 = __builtin_amdgcn_logf(1.f + e);                              // log2(1 + e): argument in [1, 2]
      rs[r] += fmaf(l2, 0.6931471805599453f, fmaf(1.f - tv[r], y, -fminf(y, 0.f)));
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next row block has landed (requested ~100 MFMAs ago)
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float tot = row16_sum(rs[r]);
    if (i == 0 && r0 + 4 * q + r < rows) out[r0 + 4 * q + r] = tot;
  }
}

// ---- the same launch with 32 rows per wave on v_mfma_f32_32x32x2_f32.  k_decode_bce_rows above keeps a
// wave's 16-row hidden block in registers and reads one B fragment (ds_read_b128) per four MFMAs; its counters say the loop is
// bound by the non-MFMA issue slots (45 VALU + 25 ds_read_b128 per 100 MFMAs, LDS bank-conflict cycles 45 % of the LDS-active
// ones) next to an MFMA pipe that is busy 73 % of the time.  Here a wave owns 32 rows: the A fragments of its [32][H] hidden
// block are H / 2 registers per lane (200 for H = 400 -- one wave per SIMD, the 512-register budget), one B fragment feeds four
// 32x32x2 MFMAs = twice the flops, a tile is 32 logits columns (half the barriers and loop overhead per flop), and the row
// blocks of W_l sit in LDS with a row stride of H + 4 floats (one DMA piece never straddles a row): the 16 lanes a
// ds_read_b128 serves together hit 16 different 4-bank groups.  K labelling: MFMA step (m, s) contracts k = 8 m + 4 h + s on
// the half-wave h = lane >> 5 -- lane (j, h) reads W_l[col j][8 m + 4 h .. + 3] as ONE 16-byte vector for four steps, and the
// hidden layer is computed straight into that order.  Output: lane (j = lane & 31, h) holds rows 8 (r >> 2) + 4 h + (r & 3),
// r = 0 .. 15, of column j.
template <int MCH, int ZP>  // H = 8 MCH ; z_dim in slices of ZP columns
__global__ __launch_bounds__(256, 1) void k_decode_bce_rows32(const float* z, int64_t rows, int Z, const float* Wd0,
                                                              const float* bd0, const float* Wl, const float* bl,
                                                              const float* x, int64_t x_rows, int D, float* out) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int H = 8 * MCH, LDW = H + 4;        // LDS row stride of a W_l row (floats)
  constexpr int RB = H * 4;                      // bytes per row
  constexpr int NPC = (RB + 1023) / 1024;        // DMA pieces per row; the last one covers LASTL lanes
  constexpr int LASTL = (RB - (NPC - 1) * 1024) / 16;
  float* bt = dyn;                   // [2][32][LDW]
  float* wd_s = dyn + 2 * 32 * LDW;  // [H][ZP] (zero past z_dim)
  float* bd_s = wd_s + H * ZP;       // [H]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * 128 + wave * 32;
  const int ntD = (D + 31) >> 5;
  const unsigned lane16 = lane * 16;
  // W_l's rows 32 nt .. 32 nt + 31 -> LDS buffer `buf` by LDS-DMA, row by row: wave w moves rows w, w + 4, ... (rows past D:
  // the last row again -- their columns are masked in the epilogue)
  auto request = [&](int nt, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = wave + 4 * u;
      const int gr = 32 * nt + r < D ? 32 * nt + r : D - 1;  // (scalar)
      const char* src = reinterpret_cast<const char*>(Wl + (size_t)gr * H);
      float* dst = bt + buf * (32 * LDW) + r * LDW;
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
        if (pc + 1 < NPC || LASTL == 64) {
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + pc * 1024 + lane16),
                                           (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
        } else if (lane < LASTL) {
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + pc * 1024 + lane16),
                                           (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
        }
      }
    }
  };
  request(0, 0);
  for (int e = tid; e < H; e += 256) bd_s[e] = bd0[e];
  // hidden layer of this wave's 32 rows as A fragments: a[m][s] = relu(b_d0[k] + <z[row j], W_d0[k]>), k = 8 m + 4 h + s
  f32x4 a[MCH];
  const int64_t zrow = (r0 + j < rows ? r0 + j : rows - 1) * Z;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MCH; ++m) {
    f32x4 v = *reinterpret_cast<const f32x4*>(bd_s + 8 * m + 4 * h);
    asm volatile("" : "+v"(v));
    a[m] = v;
  }
#ifdef MV_DBR32_NOPRO
  for (int s0 = 0; s0 < 0; s0 += ZP) {
#else
  for (int s0 = 0; s0 < Z; s0 += ZP) {
#endif
    if (s0 > 0) __syncthreads();  // the previous slice has been consumed
    float zr[ZP];
#pragma unroll
    for (int c = 0; c < ZP; ++c) {
      const float v = z[zrow + (s0 + c < Z ? s0 + c : 0)];
      zr[c] = s0 + c < Z ? v : 0.f;
    }
    for (int e0 = tid; e0 < H * ZP; e0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u < H * ZP ? e0 + 256 * u : 0;
        const int k = e / ZP, c = e - k * ZP;
        v[u] = Wd0[(size_t)k * Z + (s0 + c < Z ? s0 + c : 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u;
        if (e < H * ZP) wd_s[e] = s0 + (e % ZP) < Z ? v[u] : 0.f;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (first slice: the DMA is invisible to the compiler's LDS dependence tracking)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MCH; ++m) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 8 * m + 4 * h + s;
        float v = a[m][s];
#pragma unroll
        for (int c4 = 0; c4 < ZP; c4 += 4) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wd_s + k * ZP + c4);
          v = fmaf(zr[c4], w[0], v);
          v = fmaf(zr[c4 + 1], w[1], v);
          v = fmaf(zr[c4 + 2], w[2], v);
          v = fmaf(zr[c4 + 3], w[3], v);
        }
        asm volatile("" : "+v"(v));  // (opaque to the SLP vectorizer, see k_decode_bce_rows)
        a[m][s] = v;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MCH; ++m)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v = a[m][s];
      v = v < 0.f ? 0.f : v;  // torch.relu: NaN propagates
      asm volatile("" : "+v"(v));
      a[m][s] = v;
    }
  // targets (x broadcast over the samples): row of output r = x row (r0 + 8 (r >> 2) + (r & 3) + 4 h) mod x_rows.  With
  // x_rows a multiple of 32 (the estimator: x_rows = B = 128) a wave's 32 rows never wrap, so the row part is a SCALAR base per
  // r and the lane contributes one fixed offset (4 h rows + its column) -- no per-lane offset array next to the 200 A registers
  // (with one, the allocator parked the A fragments in AGPRs and fetched each with a v_accvgpr_read in front of its MFMA:
  // next to the f32-input MFMA every VALU instruction is step time, and that was one per MFMA)
  const int xr0 = (int)(r0 % x_rows);  // (uniform)
  const unsigned hoff = (unsigned)(4 * h * D) * 4u;
  float rs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rs[r] = 0.f;
  for (int nt = 0; nt < ntD; ++nt) {
    const int buf = nt & 1;
    const int nts = __builtin_amdgcn_readfirstlane(nt);
    if (nt + 1 < ntD) request(nts + 1, buf ^ 1);  // (uniform)
    const int col = 32 * nts + j;
    const bool cok = col < D;
    const unsigned c4 = (unsigned)(cok ? col : D - 1) * 4u;
    const float bias = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bl) + c4);
    const unsigned lo = hoff + c4;
    float tv[16];  // targets: requested here, first USED in the epilogue
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const char* xb = reinterpret_cast<const char*>(x + (size_t)(xr0 + 8 * (r >> 2) + (r & 3)) * D);  // (scalar)
      tv[r] = *reinterpret_cast<const float*>(xb + lo);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const float* brow = bt + buf * (32 * LDW) + j * LDW + 4 * h;
    // B fragments one chunk (four MFMAs = 256 cycles) ahead
    f32x4 bq = *reinterpret_cast<const f32x4*>(brow);
#pragma unroll
    for (int m0 = 0; m0 < MCH; ++m0) {
      const f32x4 bn = *reinterpret_cast<const f32x4*>(brow + 8 * (m0 + 1 < MCH ? m0 + 1 : 0));
      __builtin_amdgcn_sched_barrier(0);
#ifdef MV_DBR32_NOMMA
      acc0[m0 & 15] += a[m0][0] * bq[0] + a[m0][2] * bq[2];
      acc1[m0 & 15] += a[m0][1] * bq[1] + a[m0][3] * bq[3];
#else
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m0][0], bq[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m0][1], bq[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m0][2], bq[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m0][3], bq[3], acc1, 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
      bq = bn;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // F.binary_cross_entropy_with_logits: (1 - t) y + softplus(-y), as in k_decode_bce_rows
      const float y = acc0[r] + acc1[r] + bias;
#ifdef MV_DBR32_NOEPI
      rs[r] += (1.f - tv[r]) * y;
#else
      const float e = __builtin_amdgcn_exp2f(fabsf(y) * -1.4426950408889634f);
      const float l2 = __builtin_amdgcn_logf(1.f + e);
      const float term = fmaf(l2, 0.6931471805599453f, fmaf(1.f - tv[r], y, -fminf(y, 0.f)));
      rs[r] += cok ? term : 0.f;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next row block has landed (requested ~200 MFMAs ago)
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float tot = row16_sum(rs[r]);
    tot += __shfl_xor(tot, 16);  // the two 16-lane rows of a half-wave
    const int64_t row = r0 + 8 * (r >> 2) + 4 * h + (r & 3);
    if (j == 0 && row < rows) out[row] = tot;
  }
}

extern "C" int mvae_decode_bce_rows(const float* z, int64_t rows, int Z, const float* Wd0, const float* bd0, const float* Wl,
                                    const float* bl, const float* x, int64_t x_rows, int H, int D, float* out,
                                    void* stream) {
  if (!z || !Wd0 || !bd0 || !Wl || !bl || !x || !out || rows < 0 || Z < 1 || x_rows < 1 || H < 1 || D < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (rows == 0) return 0;
  const int nch = H >> 4;
  // (quietly: the caller takes the three-launch route, mvae_linear_forward x 2 + mvae_bce_rows)
  if ((H & 15) || (D & 15) || Z > 64 || !aligned16(Wl) || !(nch == 1 || nch == 4 || nch == 8 || nch == 16 || nch == 25 || nch == 32))
    return MVAE_E_UNSUPPORTED;
  // the kernel addresses the targets with 32-bit BYTE offsets (xo[r]): 4 GB of targets and more are declined here
  if ((uint64_t)x_rows * (uint64_t)D * 4ull >= (1ull << 32)) return MVAE_E_UNSUPPORTED;
  const int zp = Z <= 8 ? 8 : 16;
  // MVAE_DBR32=1: 32 rows per wave on the 32x32x2 MFMA (k_decode_bce_rows32) for the reference's layer width.  Measured
  // (round 6, tools/bench_decode_bce.py, interleaved): 377-385 us against 360-364 us for the 16-row kernel below -- kept for the
  // A/B and its counter dump (profiles/r06_loglik_decoder32_pmc.txt), not the default
  static const bool use32 = [] { const char* e = getenv("MVAE_DBR32"); return e && e[0] && e[0] != '0'; }();
  if (use32 && nch == 25 && (x_rows & 31) == 0) {
    const size_t lds32 = ((size_t)2 * 32 * (H + 4) + (size_t)H * zp + H) * sizeof(float);
    const dim3 grid32((unsigned)((rows + 127) / 128));
#define MV_DBR32(ZP_)                                                                                                \
  do {                                                                                                               \
    static std::atomic<bool> set_[kMaxDevices];                                                                      \
    int dev_ = 0;                                                                                                    \
    (void)hipGetDevice(&dev_);                                                                                       \
    const bool tracked_ = dev_ >= 0 && dev_ < kMaxDevices;                                                           \
    if (!(tracked_ && set_[dev_].load(std::memory_order_acquire))) {                                                 \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_bce_rows32<50, ZP_>),              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32);                   \
      if (e_ != hipSuccess) {                                                                                        \
        (void)hipGetLastError();                                                                                     \
        return MVAE_E_UNSUPPORTED;                                                                                   \
      }                                                                                                              \
      if (tracked_) set_[dev_].store(true, std::memory_order_release);                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((k_decode_bce_rows32<50, ZP_>), grid32, dim3(256), lds32, (hipStream_t)stream, z, rows, Z,    \
                       Wd0, bd0, Wl, bl, x, x_rows, D, out);                                                         \
  } while (0)
    if (zp == 8) MV_DBR32(8); else MV_DBR32(16);
#undef MV_DBR32
    LAUNCH_CHECK("decode + bce rows launch (32-row waves)");
    return 0;
  }
  const size_t lds = ((size_t)2 * 16 * H + (size_t)H * zp + H) * sizeof(float);
  const dim3 grid((unsigned)((rows + 63) / 64));
#define MV_DBR(NCH_, ZP_)                                                                                            \
  do {                                                                                                               \
    /* the attribute is per DEVICE on ROCm: one flag per device ordinal (a failure declines the call: the caller then \
       composes the generic operators) */                                                                            \
    static std::atomic<bool> set_[kMaxDevices];                                                                      \
    int dev_ = 0;                                                                                                    \
    (void)hipGetDevice(&dev_);                                                                                       \
    const bool tracked_ = dev_ >= 0 && dev_ < kMaxDevices;                                                           \
    if (lds > 64 * 1024 && !(tracked_ && set_[dev_].load(std::memory_order_acquire))) {                              \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_bce_rows<NCH_, ZP_>),              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e_ != hipSuccess) {                                                                                        \
        (void)hipGetLastError();                                                                                     \
        return MVAE_E_UNSUPPORTED;                                                                                   \
      }                                                                                                              \
      if (tracked_) set_[dev_].store(true, std::memory_order_release);                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((k_decode_bce_rows<NCH_, ZP_>), grid, dim3(256), lds, (hipStream_t)stream, z, rows, Z, Wd0,    \
                       bd0, Wl, bl, x, x_rows, D, out);                                                              \
  } while (0)
#define MV_DBR_Z(NCH_) do { if (zp == 8) MV_DBR(NCH_, 8); else MV_DBR(NCH_, 16); } while (0)
  if (nch == 1) MV_DBR_Z(1); else if (nch == 4) MV_DBR_Z(4); else if (nch == 8) MV_DBR_Z(8); else if (nch == 16) MV_DBR_Z(16); else if (nch == 25) MV_DBR_Z(25); else MV_DBR_Z(32);
#undef MV_DBR_Z
#undef MV_DBR
  LAUNCH_CHECK("decode + bce rows launch");
  return 0;
}

// The tail of the estimator in two launches (instead of ~10 small ones of the host framework):
//  * k_loglik_reduce_comps = k_loglik_reduce with the per-component terms added up inside (log_p / log_q as the component
//    kernels write them: [ncomp][n][B], components in index order) and, from the same pass over the samples, zmean[b][j] =
//    mean_n z[n][b][j] (thread sums in sample order, wave sums by DPP, wave totals in wave order: deterministic);
//  * k_cov_norm: || (x - mean_b x)^T (zmean - mean_b zmean) ||_F  (vae.py:119-121 with the mean over the samples taken first:
//    mean_n[(x - mean_x)^T (z_n - mean_b z_n)] = (x - mean_x)^T (mean_n z_n - mean_b mean_n z_n)); workgroup = 16 columns of x,
//    two passes over its [B][16] block held in LDS; the per-workgroup sums of squares are added by the last workgroup to
//    arrive, in workgroup order.
template <int ZM>  // z_dim <= ZM
__global__ __launch_bounds__(256) void k_loglik_reduce_comps(const float* bce, const float* log_p, const float* log_q,
                                                             int ncomp, const float* z, int Z, float* log_px, float* mi,
                                                             float* zmean, int n, int B) {
  __shared__ float sm[2][4];
  __shared__ float zs[4][ZM];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t nB = (size_t)n * B;
  auto terms = [&](int i, float* a1, float* a2) {
    const size_t o = (size_t)i * B + b;
    float lp = 0.f, lq = 0.f;
    for (int c = 0; c < ncomp; ++c) {
      lp += log_p[c * nB + o];
      lq += log_q[c * nB + o];
    }
    *a1 = -bce[o] + lp - lq;
    *a2 = lq - lp;
  };
  float m1 = -INFINITY, m2 = -INFINITY;
  float za[ZM];
#pragma unroll
  for (int j = 0; j < ZM; ++j) za[j] = 0.f;
  // the terms of this thread's first four samples stay in registers (n <= 1024: all of them -- one pass over memory, the
  // four samples' requests in flight together)
  float k1[4], k2[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid + 256 * u;
    k1[u] = k2[u] = -INFINITY;
    if (i < n) terms(i, &k1[u], &k2[u]);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    m1 = fmaxf(m1, k1[u]);
    m2 = fmaxf(m2, k2[u]);
  }
  for (int i = tid + 1024; i < n; i += 256) {
    float a1, a2;
    terms(i, &a1, &a2);
    m1 = fmaxf(m1, a1);
    m2 = fmaxf(m2, a2);
  }
  if (zmean) {
    for (int i = tid; i < n; i += 256) {  // (sample order per thread as before)
      const float* zr = z + ((size_t)i * B + b) * Z;
#pragma unroll
      for (int j = 0; j < ZM; ++j)
        if (j < Z) za[j] += zr[j];
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    m1 = fmaxf(m1, __shfl_xor(m1, off));
    m2 = fmaxf(m2, __shfl_xor(m2, off));
  }
  if (zmean) {
#pragma unroll
    for (int j = 0; j < ZM; ++j)
      if (j < Z) {  // uniform
        const float w = wave_sum(za[j]);
        if ((tid & 63) == 0) zs[tid >> 6][j] = w;
      }
  }
  if ((tid & 63) == 0) {
    sm[0][tid >> 6] = m1;
    sm[1][tid >> 6] = m2;
  }
  __syncthreads();
  if (zmean && tid < Z) zmean[(size_t)b * Z + tid] = ((zs[0][tid] + zs[1][tid]) + (zs[2][tid] + zs[3][tid])) / (float)n;
  m1 = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
  m2 = fmaxf(fmaxf(sm[1][0], sm[1][1]), fmaxf(sm[1][2], sm[1][3]));
  // torch.logsumexp subtracts 0 instead of an infinite maximum: a column of -inf terms gives -inf, not exp(-inf + inf) = NaN
  m1 = isinf(m1) ? 0.f : m1;
  m2 = isinf(m2) ? 0.f : m2;
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (tid + 256 * u < n) {  // (k = -inf past n would add exp(-inf) = 0 as well; the guard keeps NaN maxima out)
      s1 += expf(k1[u] - m1);
      s2 += expf(k2[u] - m2);
    }
  for (int i = tid + 1024; i < n; i += 256) {
    float a1, a2;
    terms(i, &a1, &a2);
    s1 += expf(a1 - m1);
    s2 += expf(a2 - m2);
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) {
    sm[0][tid >> 6] = s1;
    sm[1][tid >> 6] = s2;
  }
  __syncthreads();
  if (tid == 0) {
    const float ln = logf((float)n);
    log_px[b] = m1 + logf((sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3])) - ln;
    mi[b] = m2 + logf((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3])) - ln;
  }
}

template <int ZM>  // z_dim <= ZM
__global__ __launch_bounds__(256) void k_cov_norm(const float* x, const float* zmean, int B, int D, int Z, float* part,
                                                  unsigned* counter, float* out) {
  extern __shared__ float dyn[];  // x block [B][16] | zc [B][Z] | column scratch [16][17]
  __shared__ float zbar[ZM];
  __shared__ float zred[4][ZM];
  __shared__ float red[16][17];
  __shared__ float wsum[4];
  __shared__ bool last;
  float* xs = dyn;
  float* zc = dyn + (size_t)B * 16;
  const int tid = threadIdx.x, c = tid & 15, r = tid >> 4;
  const int d = blockIdx.x * 16 + c;
  // this workgroup's 16 columns of x, and zmean
  // (eight rows' requests in flight per thread: as a plain loop every LDS store waited for its own load's round trip)
  for (int b0 = r; b0 < B; b0 += 128) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 16 * u;
      v[u] = x[(size_t)(b < B ? b : 0) * D + (d < D ? d : 0)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 16 * u;
      if (b < B) xs[b * 16 + c] = d < D ? v[u] : 0.f;
    }
  }
  for (int e0 = tid; e0 < B * Z; e0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = zmean[e0 + 256 * u < B * Z ? e0 + 256 * u : 0];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 256 * u < B * Z) zc[e0 + 256 * u] = v[u];
  }
  __syncthreads();
  // column means: thread (r, c) adds the rows b = r, r + 16, ...; the four row groups of a wave meet by lane exchange, the
  // four waves in LDS (wave order: deterministic); zbar[j]: every thread adds the rows tid, tid + 256, ..., waves likewise
  const int wave = tid >> 6;
  {
    float sx = 0.f;
    for (int b = r; b < B; b += 16) sx += xs[b * 16 + c];
    sx += __shfl_xor(sx, 16);
    sx += __shfl_xor(sx, 32);
    if ((tid & 63) < 16) red[wave][c] = sx;
#pragma unroll
    for (int j = 0; j < ZM; ++j)
      if (j < Z) {  // uniform
        float sz = 0.f;
        for (int b = tid; b < B; b += 256) sz += zc[b * Z + j];
        sz = wave_sum(sz);
        if ((tid & 63) == 0) zred[wave][j] = sz;
      }
  }
  __syncthreads();
  const float xbar = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) / (float)B;
  if (tid < Z) zbar[tid] = ((zred[0][tid] + zred[1][tid]) + (zred[2][tid] + zred[3][tid])) / (float)B;
  __syncthreads();
  // cov[j][c] = sum_b (zmean[b][j] - zbar[j]) (x[b][c] - xbar[c]); thread (r, c) takes the rows b = r, r + 16, ...
  float acc[ZM];
#pragma unroll
  for (int j = 0; j < ZM; ++j) acc[j] = 0.f;
  for (int b = r; b < B; b += 16) {
    const float xv = xs[b * 16 + c] - xbar;
#pragma unroll
    for (int j = 0; j < ZM; ++j)
      if (j < Z) acc[j] = fmaf(zc[b * Z + j] - zbar[j], xv, acc[j]);
  }
  // the 16 row groups of a column: 4 per wave by lane exchange, the 4 waves through LDS (one barrier for all j)
  float* cv = dyn;  // [4][ZM][16] over the x block (read for the last time above)
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ZM; ++j)
    if (j < Z) {
      float v = acc[j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if ((tid & 63) < 16) cv[(wave * ZM + j) * 16 + c] = v;
    }
  __syncthreads();
  float sq = 0.f;
  if (tid < 16 && d < D) {
#pragma unroll
    for (int j = 0; j < ZM; ++j)
      if (j < Z) {
        const float v = (cv[(0 * ZM + j) * 16 + c] + cv[(1 * ZM + j) * 16 + c]) + (cv[(2 * ZM + j) * 16 + c] + cv[(3 * ZM + j) * 16 + c]);
        sq = fmaf(v, v, sq);
      }
  }
  // (lanes 0..15 of wave 0 hold the columns' sums of squares)
  if (tid < 64) {
    const float w = wave_sum(tid < 16 ? sq : 0.f);
    if (tid == 0) {
      // (write-through store waited for before the arrival is counted; the last workgroup acquires before it reads the others')
      store4_wt(part, (size_t)blockIdx.x, w);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
  }
  __syncthreads();
  if (last) {  // (uniform) every thread fetches its share of the partials at once -- one thread adding them in a loop paid one
               // fabric round trip per workgroup, ~10 us for 49 -- then a fixed tree: wave sums, the four waves in order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float v = 0.f;
    for (unsigned k = tid; k < gridDim.x; k += 256) v += part[k];
    v = wave_sum(v);
    if ((tid & 63) == 0) wsum[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      out[0] = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
      *counter = 0u;  // re-armed for the next call
    }
  }
}

extern "C" int mvae_loglik_reduce_comps(const float* bce, const float* log_p, const float* log_q, int ncomp, const float* z,
                                        int Z, float* log_px, float* mi, float* zmean, int n, int B, void* stream) {
  if (!bce || !log_p || !log_q || !log_px || !mi || n < 1 || B < 1 || ncomp < 1 || (zmean && (!z || Z < 1 || Z > 64)))
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (zmean && Z > 16)
    hipLaunchKernelGGL(k_loglik_reduce_comps<64>, dim3(B), dim3(256), 0, (hipStream_t)stream, bce, log_p, log_q, ncomp, z, Z,
                       log_px, mi, zmean, n, B);
  else
    hipLaunchKernelGGL(k_loglik_reduce_comps<16>, dim3(B), dim3(256), 0, (hipStream_t)stream, bce, log_p, log_q, ncomp, z, Z,
                       log_px, mi, zmean, n, B);
  LAUNCH_CHECK("loglik reduce (components) launch");
  return 0;
}

extern "C" int64_t mvae_cov_norm_workspace_floats(int D) { return (D + 15) / 16 + 1; }

extern "C" int mvae_cov_norm(const float* x, const float* zmean, int B, int D, int Z, float* workspace, float* out,
                             void* stream) {
  if (!x || !zmean || !workspace || !out || B < 1 || D < 1 || Z < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  size_t lds = ((size_t)B * 16 + (size_t)B * Z) * sizeof(float);
  if (Z > 64 || lds > 48 * 1024) return MVAE_E_UNSUPPORTED;  // (quietly: the caller composes the generic operators)
  const size_t zm = Z > 16 ? 64 : 16;
  if (lds < 4 * zm * 16 * sizeof(float)) lds = 4 * zm * 16 * sizeof(float);  // the column sums of the four waves reuse the block
  const int nwg = (D + 15) / 16;
  if (Z > 16)
    hipLaunchKernelGGL(k_cov_norm<64>, dim3(nwg), dim3(256), lds, (hipStream_t)stream, x, zmean, B, D, Z, workspace,
                       reinterpret_cast<unsigned*>(workspace + nwg), out);
  else
    hipLaunchKernelGGL(k_cov_norm<16>, dim3(nwg), dim3(256), lds, (hipStream_t)stream, x, zmean, B, D, Z, workspace,
                       reinterpret_cast<unsigned*>(workspace + nwg), out);
  LAUNCH_CHECK("cov norm launch");
  return 0;
}

extern "C" int mvae_loglik_reduce(const float* bce, const float* log_p, const float* log_q, float* log_px, float* mi,
                                  int n, int B, void* stream) {
  if (!bce || !log_p || !log_q || !log_px || !mi || n < 1 || B < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  hipLaunchKernelGGL(k_loglik_reduce, dim3(B), dim3(256), 0, (hipStream_t)stream, bce, log_p, log_q,
                     log_px, mi, n, B);
  LAUNCH_CHECK("loglik reduce launch");
  return 0;
}
