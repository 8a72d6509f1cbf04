// mvae_kernels.hip -- gfx950 kernels + C ABI (include/mvae_hip.h) for the per-batch hot path of mvae.
//
// The ELBO step (reference: ModelVAE.train_step, mt/mvae/models/vae.py:149-166) is SEVEN launches; each one is bounded
// by a grid-wide data dependency (every output of launch k is needed by every workgroup of launch k+1):
//
//   1 k_enc_fwd     h  = relu(x W_e0^T + b)                                   MFMA NT, 16x16 tiles, 4-way split-K
//   2 k_latent_fwd  heads = h W_heads^T + b  ->  per-component exp_map_mu0 / softplus / wrapped-normal sample /
//                   KL  ->  concat_z  ->  hd = relu(z W_d0^T + b)             MFMA NT + per-row manifold math
//   3 k_dec1_fwd    logits = hd W_logits^T + b ; BCE-with-logits row partials ; g = sigmoid(logits) - x
//   4 k_dec1_bwd    dW_logits = g^T hd ; db_logits ; dhd = (g W_logits) * [hd>0] ; step statistics
//   5 k_latent_bwd  dz = dhd W_d0 -> component backward (forward-mode duals, one thread per input direction)
//                   -> dheads ; dh = (dheads W_heads) * [h>0] ; dW_d0 = dhd^T z ; db_d0
//   6 k_enc_bwd     dW_e0 = dh^T x ; db_e0 ; dW_heads = dheads^T h ; db_heads ; radius gradients
//   7 k_optim       fused Adam over the flat parameter buffer + SGD on the radii
//
// Nothing here synchronises or allocates, so the host layer can capture any number of steps into one HIP graph.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mvae_hip.h"
#include "mvae_gemm.hpp"
#include "mvae_math.hpp"

using namespace mv;

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, const char* a = "", long long b = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}
static int hip_fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}
#define LAUNCH_CHECK(where)                         \
  do {                                              \
    hipError_t e_ = hipGetLastError();              \
    if (e_ != hipSuccess) return hip_fail(e_, where); \
  } while (0)

extern "C" int mvae_abi_version(void) { return MVAE_ABI_VERSION; }
extern "C" const char* mvae_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ tables
constexpr int kMaxComp = MVAE_MAX_COMPONENTS;
constexpr int kRadiiRegion = 64;  // floats reserved at the start of the flat buffers for the raw radius parameters
constexpr int kHeadsMax = 256;    // max heads_dim / z_dim held in LDS by the latent kernels
constexpr int kRows = 16;         // batch rows per workgroup in the latent kernels (one MFMA tile)

struct CompTable {
  int n;
  int total_dirs;
  mvae_component_desc c[kMaxComp];
  int dir_off[kMaxComp + 1];  // prefix sum of derivative directions per component (d + logvar_dim + trainable radius)
  unsigned char trainable[kMaxComp];
};

static int bucket_of(int dmax) {
  if (dmax <= 2) return 2;
  if (dmax <= 4) return 4;
  if (dmax <= 8) return 8;
  if (dmax <= 16) return 16;
  if (dmax <= 32) return 32;
  return 64;
}

static int fill_table(CompTable* t, const mvae_component_desc* comps, int ncomp, const unsigned char* trainable,
                      int* dmax_out) {
  if (!comps || ncomp < 1 || ncomp > kMaxComp) return fail(MVAE_E_BADARG, "ncomp out of range%s (%lld)", "", ncomp);
  memset(t, 0, sizeof(*t));
  t->n = ncomp;
  int dmax = 0, off = 0;
  for (int i = 0; i < ncomp; ++i) {
    const mvae_component_desc& c = comps[i];
    if (c.kind < 0 || c.kind > 3) return fail(MVAE_E_BADARG, "unknown manifold kind%s (%lld)", "", c.kind);
    if (c.true_dim < 1 || c.true_dim > MVAE_MAX_TRUE_DIM)
      return fail(MVAE_E_UNSUPPORTED, "true_dim outside [1, MVAE_MAX_TRUE_DIM]%s (%lld)", "", c.true_dim);
    if (c.logvar_dim != 1 && c.logvar_dim != c.true_dim)
      return fail(MVAE_E_BADARG, "logvar_dim must be 1 or true_dim%s (%lld)", "", c.logvar_dim);
    t->c[i] = c;
    t->trainable[i] = (trainable && c.kind != MVAE_EUCLIDEAN) ? trainable[i] : 0;
    t->dir_off[i] = off;
    off += c.true_dim + c.logvar_dim + (t->trainable[i] ? 1 : 0);
    if (c.true_dim > dmax) dmax = c.true_dim;
  }
  t->dir_off[ncomp] = off;
  t->total_dirs = off;
  *dmax_out = dmax;
  return 0;
}

// ------------------------------------------------------------------------------------------------ component device code
template <int DMAX, typename T>
__device__ __forceinline__ void comp_eval(int kind, const T* m, const T* l, int lvd, const float* e, int d, T rp, T* z,
                                          T* kl, T* lq, T* lp, T* mu, T* sg) {
  switch (kind) {
    case kEuclidean: component_forward<kEuclidean, DMAX, T>(m, l, lvd, e, d, rp, z, kl, lq, lp, mu, sg); break;
    case kHyperboloid: component_forward<kHyperboloid, DMAX, T>(m, l, lvd, e, d, rp, z, kl, lq, lp, mu, sg); break;
    case kSphere: component_forward<kSphere, DMAX, T>(m, l, lvd, e, d, rp, z, kl, lq, lp, mu, sg); break;
    default: component_forward<kPoincare, DMAX, T>(m, l, lvd, e, d, rp, z, kl, lq, lp, mu, sg); break;
  }
}

// forward for one (row, component); pointers are to the start of the row
template <int DMAX>
__device__ __forceinline__ void comp_fwd_row(const mvae_component_desc& c, const float* heads_row, const float* eps_row,
                                             const float* radii, float* z_row, float* z_row2, float* kl, float* lq,
                                             float* lp, float* mu_row, float* std_row) {
  float m[DMAX], l[DMAX], e[DMAX], z[DMAX + 1], mu[DMAX + 1], sg[DMAX];
  const int d = c.true_dim, lvd = c.logvar_dim;
  for (int i = 0; i < d; ++i) {
    m[i] = heads_row[c.mean_col + i];
    e[i] = eps_row[c.eps_col + i];
  }
  for (int i = 0; i < lvd; ++i) l[i] = heads_row[c.logvar_col + i];
  float rp = (c.kind == kEuclidean) ? 0.f : radii[c.radius_idx];
  float klv = 0.f, lqv = 0.f, lpv = 0.f;
  comp_eval<DMAX, float>(c.kind, m, l, lvd, e, d, rp, z, kl ? &klv : nullptr, lq ? &lqv : nullptr,
                         lq ? &lpv : nullptr, mu_row ? mu : nullptr, std_row ? sg : nullptr);
  const int A = ambient_dim(c.kind, d);
  for (int i = 0; i < A; ++i) z_row[c.z_col + i] = z[i];
  if (z_row2)
    for (int i = 0; i < A; ++i) z_row2[c.z_col + i] = z[i];
  if (kl) *kl = klv;
  if (lq) {
    *lq = lqv;
    *lp = lpv;
  }
  if (mu_row)
    for (int i = 0; i < A; ++i) mu_row[c.z_col + i] = mu[i];
  if (std_row)
    for (int i = 0; i < lvd; ++i) std_row[c.eps_col + i] = sg[i];
}

// d(loss)/d(input direction `dir`) for one (row, component): loss = <dz, z> + dkl * kl
template <int DMAX>
__device__ __forceinline__ float comp_bwd_dir(const mvae_component_desc& c, const float* heads_row,
                                              const float* eps_row, const float* radii, const float* dz_row, float dkl,
                                              int dir) {
  Dual m[DMAX], l[DMAX], z[DMAX + 1];
  float e[DMAX];
  const int d = c.true_dim, lvd = c.logvar_dim;
  for (int i = 0; i < d; ++i) {
    m[i] = Dual{heads_row[c.mean_col + i], (dir == i) ? 1.f : 0.f};
    e[i] = eps_row[c.eps_col + i];
  }
  for (int i = 0; i < lvd; ++i) l[i] = Dual{heads_row[c.logvar_col + i], (dir == d + i) ? 1.f : 0.f};
  Dual rp = Dual{(c.kind == kEuclidean) ? 0.f : radii[c.radius_idx], (dir == d + lvd) ? 1.f : 0.f};
  Dual kl;
  comp_eval<DMAX, Dual>(c.kind, m, l, lvd, e, d, rp, z, &kl, nullptr, nullptr, nullptr, nullptr);
  const int A = ambient_dim(c.kind, d);
  float g = dkl * kl.d;
  for (int i = 0; i < A; ++i) g += dz_row[c.z_col + i] * z[i].d;
  return g;
}

// ------------------------------------------------------------------------------------------------ tile jobs
// y tile = act(x W^T + b); all 256 threads of the workgroup participate.
template <bool RELU>
__device__ __forceinline__ void job_linear_fwd(float (*red)[16][17], const float* x, int ldx, const float* W, int ldw,
                                               const float* b, float* y, int ldy, int M, int N, int K, int mt, int nt) {
  const int wave = threadIdx.x >> 6;
  const bool vx = aligned16(x) && (ldx & 3) == 0, vw = aligned16(W) && (ldw & 3) == 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_nt(x, ldx, M, mt * 16, W, ldw, N, nt * 16, K, wave, 4, vx, vw, acc);
  float s = reduce_tiles(red, acc);
  const int m = mt * 16 + (threadIdx.x >> 4), n = nt * 16 + (threadIdx.x & 15);
  if (m < M && n < N) {
    float v = s + (b ? b[n] : 0.f);
    if (RELU) v = v > 0.f ? v : 0.f;
    y[(size_t)m * ldy + n] = v;
  }
}

// out[p][q] = sum_m P[m][p] Q[m][q]   (tile pt, qt)
__device__ __forceinline__ void job_tn(float (*red)[16][17], const float* P, int ldp, int NP, int pt, const float* Q,
                                       int ldq, int NQ, int qt, int Mrows, float* out, int ldo) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_tn(P, ldp, NP, pt * 16, Q, ldq, NQ, qt * 16, Mrows, wave, 4, acc);
  float s = reduce_tiles(red, acc);
  const int p = pt * 16 + (threadIdx.x >> 4), q = qt * 16 + (threadIdx.x & 15);
  if (p < NP && q < NQ) out[(size_t)p * ldo + q] = s;
}

// out[m][n] = (sum_k G[m][k] W[k][n]) * [mask[m][n] > 0]   (tile mt, nt)
__device__ __forceinline__ void job_nn(float (*red)[16][17], const float* G, int ldg, int M, int mt, const float* W,
                                       int ldw, int N, int nt, int K, const float* mask, int ldmask, float* out,
                                       int ldo) {
  const int wave = threadIdx.x >> 6;
  const bool vg = aligned16(G) && (ldg & 3) == 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_nn(G, ldg, M, mt * 16, W, ldw, N, nt * 16, K, wave, 4, vg, acc);
  float s = reduce_tiles(red, acc);
  const int m = mt * 16 + (threadIdx.x >> 4), n = nt * 16 + (threadIdx.x & 15);
  if (m < M && n < N) {
    if (mask && !(mask[(size_t)m * ldmask + n] > 0.f)) s = 0.f;
    out[(size_t)m * ldo + n] = s;
  }
}

// out[c] = sum_m Gm[m][c] for the 64 columns starting at c0 (rows added in index order within 4 interleaved groups)
__device__ __forceinline__ void job_colsum(float* lds /*[4][64]*/, const float* Gm, int ld, int Mrows, int ncols,
                                           int c0, float* out) {
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  float s = 0.f;
  if (c0 + c < ncols)
    for (int m = g; m < Mrows; m += 4) s += Gm[(size_t)m * ld + c0 + c];
  lds[g * 64 + c] = s;
  __syncthreads();
  if (g == 0 && c0 + c < ncols) out[c0 + c] = (lds[c] + lds[64 + c]) + (lds[128 + c] + lds[192 + c]);
  __syncthreads();
}

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
  job_colsum(&red[0][0][0], dy, N, M, N, b * 64, db);
}

// ------------------------------------------------------------------------------------------------ primitives (API)
enum PrimOp { OP_EXP0 = 0, OP_LOG0, OP_PT0, OP_IPT0, OP_SAMPLE, OP_ISAMPLE, OP_LOGDET };

template <int OP, int KIND, int DMAX>
__device__ __forceinline__ void prim_row(const float* a, const float* b, const float* c3, float* o1, float* o2, int d,
                                         float rp, int64_t r, int64_t at_rows) {
  constexpr int AMAX = DMAX + 1;
  const int A = ambient_dim(KIND, d);
  float R = (KIND == kEuclidean) ? 0.f : radius_of(rp);
  float t0[AMAX], t1[AMAX], t2[AMAX], t3[AMAX];
  if constexpr (OP == OP_EXP0) {
    for (int i = 0; i < d; ++i) t0[i] = a[r * d + i];
    exp_map_mu0<KIND>(t0, d, R, t1);
    for (int i = 0; i < A; ++i) o1[r * A + i] = t1[i];
  } else if constexpr (OP == OP_LOG0) {
    for (int i = 0; i < A; ++i) t0[i] = a[r * A + i];
    log_map_mu0<KIND>(t0, A, R, t1);
    for (int i = 0; i < A; ++i) o1[r * A + i] = t1[i];
  } else if constexpr (OP == OP_PT0 || OP == OP_IPT0) {
    for (int i = 0; i < A; ++i) {
      t0[i] = a[r * A + i];
      t1[i] = b[r * A + i];
    }
    if constexpr (OP == OP_PT0) pt_mu0<KIND>(t0, t1, A, R, t2);
    else inv_pt_mu0<KIND>(t0, t1, A, R, t2);
    for (int i = 0; i < A; ++i) o1[r * A + i] = t2[i];
  } else if constexpr (OP == OP_SAMPLE) {  // a = v[rows,d], b = at[at_rows,A] -> o1 = z, o2 = u
    const int64_t ar = r % at_rows;
    for (int i = 0; i < A; ++i) t1[i] = b[ar * A + i];
    if constexpr (KIND == kEuclidean) {
      for (int i = 0; i < d; ++i) t2[i] = a[r * d + i];
    } else if constexpr (KIND == kPoincare) {
      float lam = p_lambda(t1, A, 1.0f / (R * R));
      for (int i = 0; i < d; ++i) t2[i] = a[r * d + i] / lam;
    } else {
      t0[0] = 0.f;
      for (int i = 0; i < d; ++i) t0[i + 1] = a[r * d + i];
      pt_mu0<KIND>(t0, t1, A, R, t2);
    }
    exp_map<KIND, AMAX>(t2, t1, A, R, t3);
    for (int i = 0; i < A; ++i) {
      o1[r * A + i] = t3[i];
      if (o2) o2[r * A + i] = t2[i];
    }
  } else if constexpr (OP == OP_ISAMPLE) {  // a = z[rows,A], b = at -> o1 = u[rows,A], o2 = v[rows,d]
    const int64_t ar = r % at_rows;
    for (int i = 0; i < A; ++i) {
      t0[i] = a[r * A + i];
      t1[i] = b[ar * A + i];
    }
    log_map<KIND, AMAX>(t0, t1, A, R, t2);
    for (int i = 0; i < A; ++i) o1[r * A + i] = t2[i];
    if constexpr (KIND == kEuclidean) {
      for (int i = 0; i < d; ++i) o2[r * d + i] = t2[i];
    } else if constexpr (KIND == kPoincare) {
      float lam = p_lambda(t1, A, 1.0f / (R * R));
      for (int i = 0; i < d; ++i) o2[r * d + i] = t2[i] * lam;
    } else {
      inv_pt_mu0<KIND>(t2, t1, A, R, t3);
      for (int i = 0; i < d; ++i) o2[r * d + i] = t3[i + 1];
    }
  } else {  // OP_LOGDET: a = u (h,s) ; b = mu, c3 = z (p)
    if constexpr (KIND == kEuclidean) {
      o1[r] = 0.f;
    } else if constexpr (KIND == kPoincare) {
      const int64_t ar = r % at_rows;
      for (int i = 0; i < A; ++i) {
        t0[i] = b[ar * A + i];
        t1[i] = c3[r * A + i];
      }
      o1[r] = p_logdet<AMAX>(t0, t1, A, R);
    } else {
      for (int i = 0; i < A; ++i) t0[i] = a[r * A + i];
      o1[r] = logdet_u<KIND>(t0, A, R);
    }
  }
}

template <int OP, int DMAX>
__global__ __launch_bounds__(256) void k_prim(int kind, const float* a, const float* b, const float* c3, float* o1,
                                              float* o2, int64_t rows, int64_t at_rows, int d,
                                              const float* radius_param) {
  const float rp = (kind == kEuclidean || !radius_param) ? 0.f : radius_param[0];
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    switch (kind) {
      case kEuclidean: prim_row<OP, kEuclidean, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
      case kHyperboloid: prim_row<OP, kHyperboloid, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
      case kSphere: prim_row<OP, kSphere, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
      default: prim_row<OP, kPoincare, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
    }
  }
}

template <int OP>
static int launch_prim(int kind, const float* a, const float* b, const float* c3, float* o1, float* o2, int64_t rows,
                       int64_t at_rows, int d, const float* rp, void* stream) {
  if (kind < 0 || kind > 3) return fail(MVAE_E_BADARG, "unknown manifold kind%s (%lld)", "", kind);
  if (rows < 0 || d < 1) return fail(MVAE_E_BADARG, "bad rows/d%s (%lld)", "", d);
  if (d > MVAE_MAX_TRUE_DIM) return fail(MVAE_E_UNSUPPORTED, "true_dim > MVAE_MAX_TRUE_DIM%s (%lld)", "", d);
  if (kind != MVAE_EUCLIDEAN && !rp) return fail(MVAE_E_BADARG, "radius_param is NULL%s", "");
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
  if (kind == MVAE_POINCARE && (!mu || !z)) return fail(MVAE_E_BADARG, "poincare logdet needs mu and z%s", "");
  if ((kind == MVAE_HYPERBOLOID || kind == MVAE_SPHERE) && !u) return fail(MVAE_E_BADARG, "logdet needs u%s", "");
  return launch_prim<OP_LOGDET>(kind, u, mu, z, out, nullptr, rows, at_rows, d, rp, st);
}

// ------------------------------------------------------------------------------------------------ component kernels (API)
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
                                                  const float* dkl, float dkl_scalar, float* dheads, float* dradii,
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
    else atomicAdd(&dradii[c.radius_idx], g);
  }
}

#define DMAX_SWITCH(dmax, ...) \
  switch (bucket_of(dmax)) {    \
    case 2: { constexpr int DM = 2; __VA_ARGS__; } break;   \
    case 4: { constexpr int DM = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int DM = 8; __VA_ARGS__; } break;   \
    case 16: { constexpr int DM = 16; __VA_ARGS__; } break; \
    case 32: { constexpr int DM = 32; __VA_ARGS__; } break; \
    default: { constexpr int DM = 64; __VA_ARGS__; } break; \
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
  DMAX_SWITCH(dmax, hipLaunchKernelGGL((k_comp_fwd<DM>), dim3(grid), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld,
                                       radii, z, z_ld, kl, log_q, log_p, mu, sd, rows, head_rows));
  LAUNCH_CHECK("component forward launch");
  return 0;
}

extern "C" int mvae_component_backward(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld,
                                       const float* eps, int eps_ld, const float* radii, const float* dz, int z_ld,
                                       const float* dkl, float dkl_scalar, float* dheads, float* dradii, int64_t rows,
                                       void* stream) {
  if (!heads || !eps || !dz || !dheads || rows < 0) return fail(MVAE_E_BADARG, "null pointer / bad rows%s", "");
  CompTable t;
  int dmax;
  unsigned char tr[kMaxComp];
  memset(tr, dradii ? 1 : 0, sizeof(tr));
  int rc = fill_table(&t, comps, ncomp, tr, &dmax);
  if (rc) return rc;
  if (rows == 0) return 0;
  int grid = (int)((rows * t.total_dirs + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  DMAX_SWITCH(dmax, hipLaunchKernelGGL((k_comp_bwd<DM>), dim3(grid), dim3(256), 0, s, t, heads, heads_ld, eps, eps_ld,
                                       radii, dz, z_ld, dkl, dkl_scalar, dheads, dradii, rows));
  LAUNCH_CHECK("component backward launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ dense layers (API)
extern "C" int mvae_linear_forward(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K,
                                   int relu, void* stream) {
  if (!x || !W || !y || M < 0 || N < 1 || K < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (M == 0) return 0;
  if (M > (1 << 20) * 16) return fail(MVAE_E_UNSUPPORTED, "M too large%s", "");
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
  const int ntN = (N + 15) / 16, ntK = (K + 15) / 16, ntM = (int)((M + 15) / 16);
  const int n_dx = dx ? ntM * ntK : 0, n_dw = ntN * ntK, n_db = (N + 63) / 64;
  hipLaunchKernelGGL(k_linear_bwd, dim3(n_dx + n_dw + n_db), dim3(256), 0, (hipStream_t)stream, x, W, dy, dW, db, dx,
                     (int)M, N, K, relu_in, n_dw, n_dx);
  LAUNCH_CHECK("linear backward launch");
  return 0;
}

// ================================================================================================ the fused step
struct mvae_ctx {
  mvae_model_desc d;
  CompTable t;
  int dmax;
  int ldh;    // heads row stride (NH rounded up to 4)
  int ldz;    // z row stride
  // workspace carve (floats)
  int64_t o_h, o_heads, o_z, o_hd, o_g, o_bce_part, o_kl, o_dhd, o_dz, o_dheads, o_dh, o_drpart, o_total;
  int nt_d, nt_h, nt_b;  // 16-wide tile counts of D, H, B
};

static inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }
static inline int64_t up64(int64_t x) { return (x + 63) & ~(int64_t)63; }

static void carve(mvae_ctx* c) {
  const mvae_model_desc& d = c->d;
  const int64_t B = d.batch, H = d.h_dim, D = d.in_dim;
  c->ldh = (int)up4(d.heads_dim);
  c->ldz = (int)up4(d.z_dim);
  c->nt_d = (d.in_dim + 15) / 16;
  c->nt_h = (d.h_dim + 15) / 16;
  c->nt_b = (d.batch + 15) / 16;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += up64(n); return r; };
  c->o_h = take(B * H);
  c->o_heads = take(B * c->ldh);
  c->o_z = take(B * c->ldz);
  c->o_hd = take(B * H);
  c->o_g = take(B * D);
  c->o_bce_part = take((int64_t)c->nt_d * B);
  c->o_kl = take((int64_t)d.ncomp * B);
  c->o_dhd = take(B * H);
  c->o_dz = take(B * c->ldz);
  c->o_dheads = take(B * c->ldh);
  c->o_dh = take(B * H);
  c->o_drpart = take((int64_t)c->nt_b * kMaxComp);
  c->o_total = o;
}

extern "C" int64_t mvae_workspace_floats(const mvae_model_desc* desc) {
  if (!desc) return -1;
  mvae_ctx tmp;
  tmp.d = *desc;
  carve(&tmp);
  return tmp.o_total;
}

extern "C" int mvae_create(const mvae_model_desc* desc, mvae_ctx** out) {
  if (!desc || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (desc->abi_version != MVAE_ABI_VERSION) return fail(MVAE_E_BADARG, "ABI version mismatch%s", "");
  if (desc->arch != 0) return fail(MVAE_E_UNSUPPORTED, "only arch 0 (feed-forward) is built into the fused step%s", "");
  if (desc->batch < 1 || desc->in_dim < 1 || desc->h_dim < 1) return fail(MVAE_E_BADARG, "bad dims%s", "");
  if (desc->heads_dim > kHeadsMax || desc->z_dim > kHeadsMax)
    return fail(MVAE_E_UNSUPPORTED, "heads_dim / z_dim above %s%lld", "", kHeadsMax);
  if (!desc->params || !desc->grads || !desc->adam_m || !desc->adam_v || !desc->step_count || !desc->workspace ||
      !desc->stats)
    return fail(MVAE_E_BADARG, "null buffer in model desc%s", "");
  if (desc->off_radii != 0) return fail(MVAE_E_BADARG, "radii must sit at offset 0 of the flat buffers%s", "");
  const int64_t offs[] = {desc->off_w_heads, desc->off_b_heads, desc->off_w_e0, desc->off_b_e0, desc->off_w_d0,
                          desc->off_b_d0, desc->off_w_logits, desc->off_b_logits};
  for (int64_t o : offs)
    if (o < kRadiiRegion || (o & 3) || o >= desc->n_params)
      return fail(MVAE_E_ALIGN, "segment offsets must be multiples of 4 floats, >= 64 and < n_params%s (%lld)", "", o);
  if ((desc->n_params & 3) || !aligned16(desc->params) || !aligned16(desc->grads) || !aligned16(desc->adam_m) ||
      !aligned16(desc->adam_v) || !aligned16(desc->workspace))
    return fail(MVAE_E_ALIGN, "flat buffers must be 16-byte aligned with n_params %% 4 == 0%s", "");
  mvae_ctx* c = new mvae_ctx();
  c->d = *desc;
  int rc = fill_table(&c->t, desc->comps, desc->ncomp, desc->radius_trainable, &c->dmax);
  if (rc) {
    delete c;
    return rc;
  }
  int eps_dim = 0, z_dim = 0, hd = 0;
  for (int i = 0; i < desc->ncomp; ++i) {
    const mvae_component_desc& k = desc->comps[i];
    eps_dim += k.true_dim;
    z_dim += ambient_dim(k.kind, k.true_dim);
    hd += k.true_dim + k.logvar_dim;
    if (k.radius_idx != i) {
      delete c;
      return fail(MVAE_E_BADARG, "comps[i].radius_idx must equal i in the fused step%s", "");
    }
  }
  if (eps_dim != desc->eps_dim || z_dim != desc->z_dim || hd != desc->heads_dim) {
    delete c;
    return fail(MVAE_E_BADARG, "heads_dim / z_dim / eps_dim inconsistent with the component table%s", "");
  }
  c->d.comps = nullptr;
  c->d.radius_trainable = nullptr;
  carve(c);
  *out = c;
  return 0;
}

extern "C" void mvae_destroy(mvae_ctx* ctx) { delete ctx; }

// ---- 1: encoder layer
__global__ __launch_bounds__(256) void k_enc_fwd(const float* x, const float* W, const float* b, float* h, int B, int H,
                                                 int D) {
  __shared__ float red[4][16][17];
  job_linear_fwd<true>(red, x, D, W, D, b, h, H, B, H, D, blockIdx.y, blockIdx.x);
}

// ---- 2: heads GEMM + latent components + first decoder layer, 16 batch rows per workgroup
template <int DMAX>
__global__ __launch_bounds__(256) void k_latent_fwd(CompTable t, const float* h, const float* Wh, const float* bh,
                                                    const float* eps, int eps_ld, const float* radii, const float* Wd0,
                                                    const float* bd0, float* heads, int ldh, float* z, int ldz,
                                                    float* z_user, float* kl, float* kl_user, float* hd, int B, int H,
                                                    int NH, int Z) {
  __shared__ float red[4][16][17];
  __shared__ float heads_s[kRows][kHeadsMax];
  __shared__ float z_s[kRows][kHeadsMax];
  const int tid = threadIdx.x, m0 = blockIdx.x * kRows;
  const int wave = tid >> 6;
  const bool vh = aligned16(h) && (H & 3) == 0, vw = aligned16(Wh) && (H & 3) == 0;
  // heads tile(s): [16 rows] x [NH] = h W_heads^T + b
  for (int nt = 0; nt * 16 < NH; ++nt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_nt(h, H, B, m0, Wh, H, NH, nt * 16, H, wave, 4, vh, vw, acc);
    float s = reduce_tiles(red, acc);
    const int r = tid >> 4, n = nt * 16 + (tid & 15);
    if (n < NH) {
      float v = s + bh[n];
      heads_s[r][n] = v;
      if (m0 + r < B) heads[(size_t)(m0 + r) * ldh + n] = v;
    }
  }
  __syncthreads();
  // latent components: one thread per (row, component)
  for (int it = tid; it < kRows * t.n; it += 256) {
    const int r = it % kRows, ci = it / kRows;
    if (m0 + r < B) {
      const size_t row = m0 + r;
      float klv;
      comp_fwd_row<DMAX>(t.c[ci], heads_s[r], eps + row * eps_ld, radii, z_s[r], z + row * ldz, &klv, nullptr, nullptr,
                         nullptr, nullptr);
      kl[(size_t)ci * B + row] = klv;
      if (kl_user) kl_user[(size_t)ci * B + row] = klv;
      if (z_user) {
        const mvae_component_desc& c = t.c[ci];
        const int A = ambient_dim(c.kind, c.true_dim);
        for (int i = 0; i < A; ++i) z_user[row * Z + c.z_col + i] = z_s[r][c.z_col + i];
      }
    }
  }
  __syncthreads();
  // first decoder layer: hd = relu(z W_d0^T + b)   (K = Z is tiny: plain FMAs, one output per thread-iteration)
  for (int idx = tid; idx < kRows * H; idx += 256) {
    const int r = idx / H, c = idx - r * H;
    if (m0 + r < B) {
      const float* w = Wd0 + (size_t)c * Z;
      float acc = 0.f;
      for (int j = 0; j < Z; ++j) acc = fmaf(z_s[r][j], w[j], acc);
      acc += bd0[c];
      hd[(size_t)(m0 + r) * H + c] = acc > 0.f ? acc : 0.f;
    }
  }
}

// ---- 3: output layer + BCE-with-logits + its gradient
__global__ __launch_bounds__(256) void k_dec1_fwd(const float* hd, const float* W, const float* b, const float* x,
                                                  float* g, float* bce_part, float* logits_user, int B, int H, int D) {
  __shared__ float red[4][16][17];
  const int wave = threadIdx.x >> 6;
  const int mt = blockIdx.y, nt = blockIdx.x;
  const bool v1 = aligned16(hd) && (H & 3) == 0, v2 = aligned16(W) && (H & 3) == 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_nt(hd, H, B, mt * 16, W, H, D, nt * 16, H, wave, 4, v1, v2, acc);
  float s = reduce_tiles(red, acc);
  const int m = mt * 16 + (threadIdx.x >> 4), n = nt * 16 + (threadIdx.x & 15);
  float loss = 0.f;
  if (m < B && n < D) {
    const float y = s + b[n];
    const float t = x[(size_t)m * D + n];
    // F.binary_cross_entropy_with_logits (image_reconstruction.py:81-82): (1-t)*y - log_sigmoid(y)
    const float e = expf(-fabsf(y));
    const float log_sig = fminf(y, 0.f) - log1pf(e);
    loss = (1.f - t) * y - log_sig;
    const float sig = (y >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
    g[(size_t)m * D + n] = sig - t;  // d(sum bce)/d(logit)
    if (logits_user) logits_user[(size_t)m * D + n] = y;
  }
  // sum over the tile's 16 columns: the 16 lanes of one row are contiguous
  loss += __shfl_xor(loss, 8, 16);
  loss += __shfl_xor(loss, 4, 16);
  loss += __shfl_xor(loss, 2, 16);
  loss += __shfl_xor(loss, 1, 16);
  if ((threadIdx.x & 15) == 0 && m < B) bce_part[(size_t)nt * B + m] = loss;
}

// ---- 4: backward of the output layer + statistics
__global__ __launch_bounds__(256) void k_dec1_bwd(const float* g, const float* hd, const float* W, float* dW, float* db,
                                                  float* dhd, const float* bce_part, const float* kl, float* bce_user,
                                                  float* stats, float beta, int B, int H, int D, int ncomp, int n_dhd,
                                                  int n_dw, int n_db) {
  __shared__ float red[4][16][17];
  int b = blockIdx.x;
  const int ntH = (H + 15) / 16, ntD = (D + 15) / 16;
  if (b < n_dhd) {  // dhd = (g W) * [hd > 0]
    job_nn(red, g, D, B, b / ntH, W, H, H, b % ntH, D, hd, H, dhd, H);
    return;
  }
  b -= n_dhd;
  if (b < n_dw) {  // dW_logits[D,H] = g^T hd
    job_tn(red, g, D, D, b / ntH, hd, H, H, b % ntH, B, dW, H);
    return;
  }
  b -= n_dw;
  if (b < n_db) {
    job_colsum(&red[0][0][0], g, D, B, D, b * 64, db);
    return;
  }
  // statistics block (BatchStats, stats.py:144-212): sums over the batch of bce, kl_i, elbo
  float* sm = &red[0][0][0];  // >= 256 floats
  const int tid = threadIdx.x;
  float bce_acc = 0.f, elbo_acc = 0.f;
  for (int r = tid; r < B; r += 256) {
    float bce = 0.f;
    for (int nt = 0; nt < ntD; ++nt) bce += bce_part[(size_t)nt * B + r];
    if (bce_user) bce_user[r] = bce;
    float klr = 0.f;
    for (int i = 0; i < ncomp; ++i) klr = (i == 0) ? kl[r] : klr + kl[(size_t)i * B + r];
    bce_acc += bce;
    elbo_acc += (-bce - beta * klr);
  }
  auto block_sum = [&](float v) -> float {
    sm[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) sm[tid] += sm[tid + s];
      __syncthreads();
    }
    float r = sm[0];
    __syncthreads();
    return r;
  };
  const float bce_sum = block_sum(bce_acc);
  const float elbo_sum = block_sum(elbo_acc);
  float kl_total = 0.f;
  const int last = 4 + ncomp;
  for (int i = 0; i < ncomp; ++i) {
    float a = 0.f;
    for (int r = tid; r < B; r += 256) a += kl[(size_t)i * B + r];
    const float s = block_sum(a);
    kl_total += s;
    if (tid == 0) {
      stats[4 + i] += s;
      stats[last + 4 + i] = s;
    }
  }
  if (tid == 0) {
    stats[0] += bce_sum;
    stats[1] += kl_total;
    stats[2] += elbo_sum;
    stats[3] += 1.f;
    stats[last + 0] = bce_sum;
    stats[last + 1] = kl_total;
    stats[last + 2] = elbo_sum;
    stats[last + 3] = 1.f;
  }
}

// ---- 5: backward through the first decoder layer, the latent components and the heads
template <int DMAX>
__global__ __launch_bounds__(256) void k_latent_bwd(CompTable t, const float* dhd, const float* Wd0, const float* heads,
                                                    int ldh, const float* eps, int eps_ld, const float* radii,
                                                    const float* z, int ldz, const float* h, const float* Wh,
                                                    float* dheads, float* dh, float* dWd0, float* dbd0, float* drpart,
                                                    float beta, int B, int H, int NH, int Z, int n_rows, int n_dw) {
  __shared__ float red[4][16][17];
  __shared__ float dz_s[kRows][kHeadsMax];
  __shared__ float dheads_s[kRows][kHeadsMax];
  __shared__ float dr_s[kMaxComp][kRows];
  int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (b >= n_rows) {
    b -= n_rows;
    if (b < n_dw) {  // dW_d0[H,Z] = dhd^T z
      const int ntZ = (Z + 15) / 16;
      job_tn(red, dhd, H, H, b / ntZ, z, ldz, Z, b % ntZ, B, dWd0, Z);
      return;
    }
    b -= n_dw;
    job_colsum(&red[0][0][0], dhd, H, B, H, b * 64, dbd0);
    return;
  }
  const int m0 = b * kRows, wave = tid >> 6;
  const bool vg = aligned16(dhd) && (H & 3) == 0;
  // dz tile(s) = dhd[16 rows] W_d0
  for (int nt = 0; nt * 16 < Z; ++nt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_nn(dhd, H, B, m0, Wd0, Z, Z, nt * 16, H, wave, 4, vg, acc);
    float s = reduce_tiles(red, acc);
    const int r = tid >> 4, n = nt * 16 + (tid & 15);
    if (n < Z) dz_s[r][n] = s;
  }
  for (int i = tid; i < kMaxComp * kRows; i += 256) (&dr_s[0][0])[i] = 0.f;
  __syncthreads();
  // component backward: one thread per (row, component, input direction)
  for (int it = tid; it < kRows * t.total_dirs; it += 256) {
    const int r = it % kRows, gd = it / kRows;
    if (m0 + r >= B) continue;
    int ci = 0;
    while (gd >= t.dir_off[ci + 1]) ++ci;
    const int dir = gd - t.dir_off[ci];
    const mvae_component_desc& c = t.c[ci];
    const size_t row = m0 + r;
    float gv = comp_bwd_dir<DMAX>(c, heads + row * ldh, eps + row * eps_ld, radii, dz_s[r], beta, dir);
    if (dir < c.true_dim) dheads_s[r][c.mean_col + dir] = gv;
    else if (dir < c.true_dim + c.logvar_dim) dheads_s[r][c.logvar_col + (dir - c.true_dim)] = gv;
    else dr_s[ci][r] = gv;
  }
  __syncthreads();
  if (tid < t.n) {
    float s = 0.f;
    for (int r = 0; r < kRows; ++r) s += dr_s[tid][r];
    drpart[(size_t)b * kMaxComp + tid] = s;
  }
  for (int idx = tid; idx < kRows * NH; idx += 256) {
    const int r = idx / NH, n = idx - r * NH;
    if (m0 + r < B) dheads[(size_t)(m0 + r) * ldh + n] = dheads_s[r][n];
  }
  // dh = (dheads W_heads) * [h > 0]   (K = NH is small: plain FMAs)
  for (int idx = tid; idx < kRows * H; idx += 256) {
    const int r = idx / H, c = idx - r * H;
    if (m0 + r < B) {
      float acc = 0.f;
      for (int n = 0; n < NH; ++n) acc = fmaf(dheads_s[r][n], Wh[(size_t)n * H + c], acc);
      const size_t o = (size_t)(m0 + r) * H + c;
      dh[o] = (h[o] > 0.f) ? acc : 0.f;
    }
  }
}

// ---- 6: encoder / heads weight gradients + radius gradients
__global__ __launch_bounds__(256) void k_enc_bwd(CompTable t, const float* dh, const float* x, const float* dheads,
                                                 int ldh, const float* h, const float* drpart, float* dWe0,
                                                 float* dbe0, float* dWh, float* dbh, float* dradii, int B, int H,
                                                 int D, int NH, int n_we0, int n_wh, int n_be0, int n_bh, int nt_b) {
  __shared__ float red[4][16][17];
  int b = blockIdx.x;
  if (b < n_we0) {  // dW_e0[H,D] = dh^T x
    const int ntD = (D + 15) / 16;
    job_tn(red, dh, H, H, b / ntD, x, D, D, b % ntD, B, dWe0, D);
    return;
  }
  b -= n_we0;
  if (b < n_wh) {  // dW_heads[NH,H] = dheads^T h
    const int ntH = (H + 15) / 16;
    job_tn(red, dheads, ldh, NH, b / ntH, h, H, H, b % ntH, B, dWh, H);
    return;
  }
  b -= n_wh;
  if (b < n_be0) {
    job_colsum(&red[0][0][0], dh, H, B, H, b * 64, dbe0);
    return;
  }
  b -= n_be0;
  if (b < n_bh) {
    job_colsum(&red[0][0][0], dheads, ldh, B, NH, b * 64, dbh);
    return;
  }
  // radius gradients: fixed-order sum of the per-workgroup partials of launch 5
  const int tid = threadIdx.x;
  if (tid < kRadiiRegion) {
    float s = 0.f;
    if (tid < t.n && t.trainable[tid])
      for (int w = 0; w < nt_b; ++w) s += drpart[(size_t)w * kMaxComp + tid];
    // radius_idx == component index in the flat layout
    dradii[tid] = s;
  }
}

// ---- 7: fused optimizer.  torch.optim.Adam (single-tensor CPU formulas, defaults betas=(0.9,0.999), eps=1e-8) over
// every float of the flat buffer past the radii region; torch.optim.SGD(lr=curvature_lr) on trainable radii.
__global__ __launch_bounds__(256) void k_optim(CompTable t, float* p, const float* g, float* m, float* v, int n4,
                                               int* counters, double lr, double curv_lr, int do_curv) {
  __shared__ float sh[2];
  const int tid = threadIdx.x;
  if (tid == 0) {
    const int step = *(volatile int*)&counters[0] + 1;
    const double bc1 = 1.0 - pow(0.9, (double)step);
    const double bc2 = 1.0 - pow(0.999, (double)step);
    sh[0] = (float)(-(lr / bc1));  // addcdiv_(exp_avg, denom, value=-step_size)
    sh[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float neg_step = sh[0], bc2s = sh[1];
  const float w1 = (float)(1.0 - 0.9), b2 = 0.999f, w2 = (float)(1.0 - 0.999);
  const int i4 = blockIdx.x * 256 + tid + kRadiiRegion / 4;
  if (i4 < n4) {
    float4 pp = reinterpret_cast<float4*>(p)[i4];
    const float4 gg = reinterpret_cast<const float4*>(g)[i4];
    float4 mm = reinterpret_cast<float4*>(m)[i4];
    float4 vv = reinterpret_cast<float4*>(v)[i4];
#define ADAM1(P, G, M, V)                         \
  M = M + w1 * (G - M);                           \
  V = V * b2 + (w2 * G) * G;                      \
  P = P + (neg_step * M) / (sqrtf(V) / bc2s + 1e-8f);
    ADAM1(pp.x, gg.x, mm.x, vv.x)
    ADAM1(pp.y, gg.y, mm.y, vv.y)
    ADAM1(pp.z, gg.z, mm.z, vv.z)
    ADAM1(pp.w, gg.w, mm.w, vv.w)
#undef ADAM1
    reinterpret_cast<float4*>(p)[i4] = pp;
    reinterpret_cast<float4*>(m)[i4] = mm;
    reinterpret_cast<float4*>(v)[i4] = vv;
  }
  if (blockIdx.x == 0 && do_curv && tid < t.n && t.trainable[tid]) {
    p[tid] = p[tid] + (float)(-curv_lr) * g[tid];  // SGD: param.add_(grad, alpha=-lr)
  }
  // the last workgroup to finish advances the step counter (every other workgroup has read it by then)
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const int done = atomicAdd(&counters[1], 1);
    if (done == (int)gridDim.x - 1) {
      counters[1] = 0;
      counters[0] = counters[0] + 1;
      __threadfence();
    }
  }
}

static int step_fwd_bwd_impl(mvae_ctx* c, const float* x, const float* eps, float beta, int want_outputs,
                             float* logits, float* concat_z, float* bce, float* kl, void* stream, hipEvent_t* ev) {
#define MARK() do { if (ev) hipEventRecord(*ev++, s); } while (0)
  if (!c || !x || !eps) return fail(MVAE_E_BADARG, "null pointer%s", "");
  const mvae_model_desc& d = c->d;
  const int B = d.batch, H = d.h_dim, D = d.in_dim, NH = d.heads_dim, Z = d.z_dim;
  hipStream_t s = (hipStream_t)stream;
  float* ws = d.workspace;
  float *h = ws + c->o_h, *heads = ws + c->o_heads, *z = ws + c->o_z, *hd = ws + c->o_hd, *g = ws + c->o_g,
        *bce_part = ws + c->o_bce_part, *klw = ws + c->o_kl, *dhd = ws + c->o_dhd, *dheads = ws + c->o_dheads,
        *dh = ws + c->o_dh, *drpart = ws + c->o_drpart;
  const float* P = d.params;
  float* G = d.grads;
  if (!want_outputs) logits = concat_z = bce = kl = nullptr;

  hipLaunchKernelGGL(k_enc_fwd, dim3(c->nt_h, c->nt_b), dim3(256), 0, s, x, P + d.off_w_e0, P + d.off_b_e0, h, B, H, D);
  MARK();
  DMAX_SWITCH(c->dmax, hipLaunchKernelGGL((k_latent_fwd<DM>), dim3(c->nt_b), dim3(256), 0, s, c->t, h,
                                          P + d.off_w_heads, P + d.off_b_heads, eps, d.eps_dim, P + d.off_radii,
                                          P + d.off_w_d0, P + d.off_b_d0, heads, c->ldh, z, c->ldz, concat_z, klw, kl,
                                          hd, B, H, NH, Z));
  MARK();
  hipLaunchKernelGGL(k_dec1_fwd, dim3(c->nt_d, c->nt_b), dim3(256), 0, s, hd, P + d.off_w_logits, P + d.off_b_logits,
                     x, g, bce_part, logits, B, H, D);
  MARK();
  {
    const int n_dhd = c->nt_b * c->nt_h, n_dw = c->nt_d * c->nt_h, n_db = (D + 63) / 64;
    hipLaunchKernelGGL(k_dec1_bwd, dim3(n_dhd + n_dw + n_db + 1), dim3(256), 0, s, g, hd, P + d.off_w_logits,
                       G + d.off_w_logits, G + d.off_b_logits, dhd, bce_part, klw, bce, d.stats, beta, B, H, D,
                       d.ncomp, n_dhd, n_dw, n_db);
    MARK();
  }
  {
    const int n_dw = c->nt_h * ((Z + 15) / 16), n_db = (H + 63) / 64;
    DMAX_SWITCH(c->dmax, hipLaunchKernelGGL((k_latent_bwd<DM>), dim3(c->nt_b + n_dw + n_db), dim3(256), 0, s, c->t,
                                            dhd, P + d.off_w_d0, heads, c->ldh, eps, d.eps_dim, P + d.off_radii, z,
                                            c->ldz, h, P + d.off_w_heads, dheads, dh, G + d.off_w_d0, G + d.off_b_d0,
                                            drpart, beta, B, H, NH, Z, c->nt_b, n_dw));
    MARK();
  }
  {
    const int n_we0 = c->nt_h * c->nt_d, n_wh = ((NH + 15) / 16) * c->nt_h, n_be0 = (H + 63) / 64,
              n_bh = (NH + 63) / 64;
    hipLaunchKernelGGL(k_enc_bwd, dim3(n_we0 + n_wh + n_be0 + n_bh + 1), dim3(256), 0, s, c->t, dh, x, dheads, c->ldh,
                       h, drpart, G + d.off_w_e0, G + d.off_b_e0, G + d.off_w_heads, G + d.off_b_heads,
                       G + d.off_radii, B, H, D, NH, n_we0, n_wh, n_be0, n_bh, c->nt_b);
    MARK();
  }
#undef MARK
  LAUNCH_CHECK("step forward/backward launch");
  return 0;
}

extern "C" int mvae_step_forward_backward(mvae_ctx* c, const float* x, const float* eps, float beta, int want_outputs,
                                          float* logits, float* concat_z, float* bce, float* kl, void* stream) {
  return step_fwd_bwd_impl(c, x, eps, beta, want_outputs, logits, concat_z, bce, kl, stream, nullptr);
}

extern "C" int mvae_step_optimizer(mvae_ctx* c, int do_curvature_step, void* stream) {
  if (!c) return fail(MVAE_E_BADARG, "null pointer%s", "");
  const mvae_model_desc& d = c->d;
  const int n4 = d.n_params / 4;
  const int blocks = (n4 - kRadiiRegion / 4 + 255) / 256;
  hipLaunchKernelGGL(k_optim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c->t, d.params, d.grads, d.adam_m,
                     d.adam_v, n4, d.step_count, (double)d.lr, (double)d.curvature_lr, do_curvature_step);
  LAUNCH_CHECK("optimizer launch");
  return 0;
}

extern "C" int mvae_train_step(mvae_ctx* c, const float* x, const float* eps, float beta, int do_curvature_step,
                               void* stream) {
  int rc = mvae_step_forward_backward(c, x, eps, beta, 0, nullptr, nullptr, nullptr, nullptr, stream);
  if (rc) return rc;
  return mvae_step_optimizer(c, do_curvature_step, stream);
}

extern "C" int mvae_step_profile(mvae_ctx* c, const float* x, const float* eps, float beta, int do_curvature_step,
                                 int iters, float* ms_out, void* stream) {
  if (!c || !x || !eps || !ms_out || iters < 1) return fail(MVAE_E_BADARG, "null pointer / iters < 1%s", "");
  constexpr int NK = MVAE_STEP_KERNELS;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev[NK + 1];
  for (auto& e : ev) {
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return hip_fail(rc, "hipEventCreate");
  }
  double acc[NK] = {0};
  int rc = 0;
  for (int it = 0; it < iters && rc == 0; ++it) {
    hipEventRecord(ev[0], s);
    rc = step_fwd_bwd_impl(c, x, eps, beta, 0, nullptr, nullptr, nullptr, nullptr, stream, &ev[1]);
    if (rc) break;
    rc = mvae_step_optimizer(c, do_curvature_step, stream);
    hipEventRecord(ev[NK], s);
    hipError_t e = hipEventSynchronize(ev[NK]);
    if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
    for (int k = 0; k < NK; ++k) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
      acc[k] += ms;
    }
  }
  for (auto& e : ev) hipEventDestroy(e);
  if (rc) return rc;
  for (int k = 0; k < NK; ++k) ms_out[k] = (float)(acc[k] / iters);
  return 0;
}
