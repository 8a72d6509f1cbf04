// mvae_kernels.hip -- gfx950 kernels + C ABI (include/mvae_hip.h) for the per-batch hot path of mvae.
//
// The ELBO step (reference: ModelVAE.train_step, mt/mvae/models/vae.py:149-166) is SIX launches; each cut is a
// grid-wide data dependency (every output of launch k is needed by every workgroup of launch k+1):
//
//   1 k_enc_fwd     h  = relu(x W_e0^T + b)                      MFMA NT, 16x16 tile / workgroup, 8 waves split K
//   2 k_latent_fwd  one batch ROW per workgroup: heads = h W_heads^T + b -> per-component exp_map_mu0 / softplus /
//                   wrapped-normal sample / KL -> concat_z -> hd = relu(z W_d0^T + b); waves 4..7 evaluate the same
//                   components over dual numbers (d z, d kl per input direction) for launch 5
//   3 k_dec1_fwd    logits = hd W_logits^T + b ; BCE-with-logits row partials ; g = sigmoid(logits) - x
//   4 k_dec1_bwd    dhd = (g W_logits) * [hd>0] ; db_logits (+Adam) ; step statistics (BatchStats)
//   5 k_latent_bwd  rows: dz = dhd W_d0 -> contraction with the dual records of launch 2 -> dheads ;
//                   dh = (dheads W_heads) * [h>0]                  tiles: dW_logits = g^T hd (+Adam)
//   6 k_enc_bwd     dW_e0 = dh^T x, dW_heads, dW_d0, their biases (+Adam) ; radius gradients (+SGD)
//
// In the single-GPU step the optimizer runs in the gradient epilogues, each weight one launch after its last read;
// the two-call path (mvae_step_forward_backward -> all-reduce -> mvae_step_optimizer) uses k_optim instead.
// Nothing here synchronises or allocates, so the host layer can capture any number of steps into one HIP graph.
// Further down: manifold primitives, component operators, generic dense layers, log-likelihood helpers and the
// patch-matrix gathers of the conv architecture -- the rest of the C ABI.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mvae_hip.h"
#include "mvae_gemm.hpp"
#include "mvae_math.hpp"

using namespace mv;

// ------------------------------------------------------------------------------------------------ dev timing hooks
// -DMV_DBG_TIMING: workgroup 0 of the latent kernels stamps wall_clock64() (100 MHz) at phase boundaries.
#ifdef MV_DBG_TIMING
__device__ unsigned long long g_dbg[64];
#define MV_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)
#define MV_STAMP_B(i, blk) do { if (blockIdx.x == (blk) && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)
#ifndef MV_STAMP_BLK
#define MV_STAMP_BLK 200  // which workgroup of launches 5 / 6 stamps its wave-tile phases
#endif
// start / end time and kind of every workgroup of launch L (plain stores to per-workgroup slots: no contention)
__device__ unsigned long long g_span[6][3][2048];
#define MV_SPAN_BEGIN(L) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_span[L][0][blockIdx.x] = wall_clock64(); } while (0)
#define MV_SPAN_END(L, kind) do { if (threadIdx.x == 0 && blockIdx.x < 2048) { g_span[L][1][blockIdx.x] = wall_clock64(); g_span[L][2][blockIdx.x] = (kind); } } while (0)
// the same for another thread of the workgroup, recorded `off` slots further (e.g. the dual waves of launch 2)
#define MV_SPAN_END_T(L, kind, thr, off) do { if (threadIdx.x == (thr) && blockIdx.x + (off) < 2048) { g_span[L][0][blockIdx.x + (off)] = g_span[L][0][blockIdx.x]; g_span[L][1][blockIdx.x + (off)] = wall_clock64(); g_span[L][2][blockIdx.x + (off)] = (kind); } } while (0)
extern "C" int mvae_debug_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * n);
}
extern "C" int mvae_debug_read_spans(unsigned long long* out /* [6][3][2048] */) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 6 * 3 * 2048);
}
#else
#define MV_STAMP(i) do {} while (0)
#define MV_STAMP_B(i, blk) do {} while (0)
#define MV_SPAN_BEGIN(L) do {} while (0)
#define MV_SPAN_END(L, kind) do {} while (0)
#define MV_SPAN_END_T(L, kind, thr, off) do {} while (0)
#endif

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
  short first_dir[kMaxComp];  // the same prefix with EVERY radius direction counted: record index in the dual workspace
  unsigned char trainable[kMaxComp];  // bit 0: trainable radius/curvature, bit 1: in the gradient-clip group (`u`)
  // Placement of the components on the 4 waves of a latent workgroup: components of the same manifold kind share a
  // wave (one instruction stream, no divergence), different kinds run on different waves.
  unsigned char wave_of[kMaxComp];
  unsigned char lane_of[kMaxComp];
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
  int dmax = 0, off = 0, first = 0;
  for (int i = 0; i < ncomp; ++i) {
    const mvae_component_desc& c = comps[i];
    if (c.kind < 0 || c.kind >= kNumKinds) return fail(MVAE_E_BADARG, "unknown manifold kind%s (%lld)", "", c.kind);
    if (c.true_dim < 1 || c.true_dim > MVAE_MAX_TRUE_DIM)
      return fail(MVAE_E_UNSUPPORTED, "true_dim outside [1, MVAE_MAX_TRUE_DIM]%s (%lld)", "", c.true_dim);
    if (c.logvar_dim != 1 && c.logvar_dim != c.true_dim)
      return fail(MVAE_E_BADARG, "logvar_dim must be 1 or true_dim%s (%lld)", "", c.logvar_dim);
    t->c[i] = c;
    // bit 0: SGD-trainable radius / curvature; bit 1: member of the clip_grad_norm_ group (universal curvatures)
    t->trainable[i] = (trainable && c.kind != MVAE_EUCLIDEAN && trainable[i])
                          ? (unsigned char)(1 | (c.kind == MVAE_UNIVERSAL ? 2 : 0))
                          : 0;
    t->dir_off[i] = off;
    off += c.true_dim + c.logvar_dim + (t->trainable[i] ? 1 : 0);
    t->first_dir[i] = (short)first;
    first += c.true_dim + c.logvar_dim + 1;
    if (c.true_dim > dmax) dmax = c.true_dim;
  }
  t->dir_off[ncomp] = off;
  t->total_dirs = off;
  *dmax_out = dmax;
  // wave placement: the kinds present split the 4 waves between them; a kind's components go round-robin over its waves
  int kinds[kNumKinds], nk = 0;
  for (int k = 0; k < kNumKinds; ++k) {
    bool present = false;
    for (int i = 0; i < ncomp; ++i) present |= (comps[i].kind == k);
    if (present) kinds[nk++] = k;
  }
  const int wpk = nk ? (4 / nk > 0 ? 4 / nk : 1) : 1;
  int fill[4] = {0, 0, 0, 0};
  for (int ki = 0; ki < nk; ++ki) {
    int rr = 0;
    for (int i = 0; i < ncomp; ++i)
      if (comps[i].kind == kinds[ki]) {
        const int w = (ki * wpk + (rr++ % wpk)) & 3;
        t->wave_of[i] = (unsigned char)w;
        t->lane_of[i] = (unsigned char)fill[w]++;
      }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ component device code
template <int DMAX, typename T>
__device__ __forceinline__ void comp_eval(int kind, const T* m, const T* l, int lvd, const float* e, int d, T rp, T* z,
                                          T* kl, T* lq, T* lp, T* mu, T* sg) {
  kind = resolve_universal(kind, rp);  // `u`: Poincare ball / projected sphere / Euclidean by the sign of K
#define MV_KIND_SWITCH(DD, LL)                                                                              \
  switch (kind) {                                                                                           \
    case kEuclidean: component_forward<kEuclidean, DMAX, T>(m, l, LL, e, DD, rp, z, kl, lq, lp, mu, sg); break;     \
    case kHyperboloid: component_forward<kHyperboloid, DMAX, T>(m, l, LL, e, DD, rp, z, kl, lq, lp, mu, sg); break; \
    case kSphere: component_forward<kSphere, DMAX, T>(m, l, LL, e, DD, rp, z, kl, lq, lp, mu, sg); break;           \
    case kProjSphere: component_forward<kProjSphere, DMAX, T>(m, l, LL, e, DD, rp, z, kl, lq, lp, mu, sg); break;   \
    default: component_forward<kPoincare, DMAX, T>(m, l, LL, e, DD, rp, z, kl, lq, lp, mu, sg); break;              \
  }
  // the common case (every dimension equals the bucket bound) is instantiated with compile-time dimensions, which
  // folds away every loop guard of the small-vector code
  if (d == DMAX && lvd == DMAX) {
    MV_KIND_SWITCH(DMAX, DMAX)
  } else {
    MV_KIND_SWITCH(d, lvd)
  }
#undef MV_KIND_SWITCH
}

// forward for one (row, component); pointers are to the start of the row
template <int DMAX>
__device__ __forceinline__ void comp_fwd_row(const mvae_component_desc& c, const float* heads_row, const float* eps_row,
                                             const float* radii, float* z_row, float* z_row2, float* kl, float* lq,
                                             float* lp, float* mu_row, float* std_row) {
  MV_BOUNDS(DMAX + 1);
  float m[kN], l[kN], e[kN], z[kN], mu[kN], sg[kN];
  const int d = c.true_dim, lvd = c.logvar_dim;
  MV_FOR(i, 0, d) {
    m[i] = heads_row[c.mean_col + i];
    e[i] = eps_row[c.eps_col + i];
  }
  MV_FOR(i, 0, lvd) l[i] = heads_row[c.logvar_col + i];
  float rp = (c.kind == kEuclidean) ? 0.f : radii[c.radius_idx];
  float klv = 0.f, lqv = 0.f, lpv = 0.f;
  comp_eval<DMAX, float>(c.kind, m, l, lvd, e, d, rp, z, kl ? &klv : nullptr, lq ? &lqv : nullptr,
                         lq ? &lpv : nullptr, mu_row ? mu : nullptr, std_row ? sg : nullptr);
  const int A = ambient_dim(c.kind, d);
  MV_FOR(i, 0, A) z_row[c.z_col + i] = z[i];
  if (z_row2) {
    MV_FOR(i, 0, A) z_row2[c.z_col + i] = z[i];
  }
  if (kl) *kl = klv;
  if (lq) {
    *lq = lqv;
    *lp = lpv;
  }
  if (mu_row) {
    MV_FOR(i, 0, A) mu_row[c.z_col + i] = mu[i];
  }
  if (std_row) {
    MV_FOR(i, 0, lvd) std_row[c.eps_col + i] = sg[i];
  }
}

// Derivative of one (row, component) along input direction `dir` (0..d-1: mean head, d..d+lvd-1: logvar head,
// d+lvd: radius / curvature): zd[i] = d z_i / d dir, returns d kl / d dir.  Needs no upstream gradient, so the latent
// backward kernel runs it while dz is still being reduced.
template <int DMAX>
__device__ __forceinline__ float comp_dual_dir(const mvae_component_desc& c, const float* heads_row,
                                               const float* eps_row, const float* radii, int dir, float* zd) {
  MV_BOUNDS(DMAX + 1);
  Dual m[kN], l[kN], z[kN];
  float e[kN];
  const int d = c.true_dim, lvd = c.logvar_dim;
  MV_FOR(i, 0, d) {
    m[i] = Dual{heads_row[c.mean_col + i], (dir == i) ? 1.f : 0.f};
    e[i] = eps_row[c.eps_col + i];
  }
  MV_FOR(i, 0, lvd) l[i] = Dual{heads_row[c.logvar_col + i], (dir == d + i) ? 1.f : 0.f};
  Dual rp = Dual{(c.kind == kEuclidean) ? 0.f : radii[c.radius_idx], (dir == d + lvd) ? 1.f : 0.f};
  Dual kl;
  comp_eval<DMAX, Dual>(c.kind, m, l, lvd, e, d, rp, z, &kl, nullptr, nullptr, nullptr, nullptr);
  const int A = ambient_dim(c.kind, d);
  MV_FOR(i, 0, A) zd[i] = z[i].d;
  return kl.d;
}

// d(loss)/d(input direction `dir`) for one (row, component): loss = <dz, z> + dkl * kl
template <int DMAX>
__device__ __forceinline__ float comp_bwd_dir(const mvae_component_desc& c, const float* heads_row,
                                              const float* eps_row, const float* radii, const float* dz_row, float dkl,
                                              int dir) {
  MV_BOUNDS(DMAX + 1);
  float zd[kN];
  const float kld = comp_dual_dir<DMAX>(c, heads_row, eps_row, radii, dir, zd);
  const int A = ambient_dim(c.kind, c.true_dim);
  float g = dkl * kld;
  MV_FOR(i, 0, A) g += dz_row[c.z_col + i] * zd[i];
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
    if (RELU) v = v < 0.f ? 0.f : v;  // torch.relu: NaN propagates
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

// out[c] = sum_m Gm[m][c] for the 16 columns starting at c0: thread (g = tid>>4, c = tid&15) adds rows g, g+16, ...
// (loads issued in batches of 8), the 16 row-groups meet in LDS and are added in index order.
constexpr int kColsPerBlock = 16;
__device__ __forceinline__ void job_colsum(float* lds /*>= 16*17 floats*/, const float* Gm, int ld, int Mrows,
                                           int ncols, int c0, float* out) {
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  float s = 0.f;
  if (c0 + c < ncols) {
    const float* col = Gm + c0 + c;
    int m = g;
    for (; m + 16 * 7 < Mrows; m += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(m + 16 * u) * ld];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; m < Mrows; m += 16) s += col[(size_t)m * ld];
  }
  lds[g * 17 + c] = s;
  __syncthreads();
  if (g == 0 && c0 + c < ncols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += lds[q * 17 + c];
    out[c0 + c] = t;
  }
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
  job_colsum(&red[0][0][0], dy, N, M, N, b * kColsPerBlock, db);
}

// ------------------------------------------------------------------------------------------------ primitives (API)
enum PrimOp { OP_EXP0 = 0, OP_LOG0, OP_PT0, OP_IPT0, OP_SAMPLE, OP_ISAMPLE, OP_LOGDET };

template <int OP, int KIND, int DMAX>
__device__ __forceinline__ void prim_row(const float* a, const float* b, const float* c3, float* o1, float* o2, int d,
                                         float rp, int64_t r, int64_t at_rows) {
  constexpr int AMAX = DMAX + 1;
  MV_BOUNDS(AMAX);
  const int A = ambient_dim(KIND, d);
  float R = (KIND == kEuclidean) ? 0.f : radius_of(rp);
  float t0[AMAX], t1[AMAX], t2[AMAX], t3[AMAX];
  if constexpr (OP == OP_EXP0) {
    MV_FOR(i, 0, d) t0[i] = a[r * d + i];
    exp_map_mu0<KIND, AMAX>(t0, d, R, t1);
    MV_FOR(i, 0, A) o1[r * A + i] = t1[i];
  } else if constexpr (OP == OP_LOG0) {
    MV_FOR(i, 0, A) t0[i] = a[r * A + i];
    log_map_mu0<KIND, AMAX>(t0, A, R, t1);
    MV_FOR(i, 0, A) o1[r * A + i] = t1[i];
  } else if constexpr (OP == OP_PT0 || OP == OP_IPT0) {
    MV_FOR(i, 0, A) {
      t0[i] = a[r * A + i];
      t1[i] = b[r * A + i];
    }
    if constexpr (OP == OP_PT0) pt_mu0<KIND, AMAX>(t0, t1, A, R, t2);
    else inv_pt_mu0<KIND, AMAX>(t0, t1, A, R, t2);
    MV_FOR(i, 0, A) o1[r * A + i] = t2[i];
  } else if constexpr (OP == OP_SAMPLE) {  // a = v[rows,d], b = at[at_rows,A] -> o1 = z, o2 = u
    const int64_t ar = r % at_rows;
    MV_FOR(i, 0, A) t1[i] = b[ar * A + i];
    if constexpr (KIND == kEuclidean) {
      MV_FOR(i, 0, d) t2[i] = a[r * d + i];
    } else if constexpr (KIND == kPoincare || KIND == kProjSphere) {
      float lam = (KIND == kPoincare) ? p_lambda<AMAX>(t1, A, 1.0f / (R * R)) : d_lambda<AMAX>(t1, A, 1.0f / (R * R));
      MV_FOR(i, 0, d) t2[i] = a[r * d + i] / lam;
    } else {
      t0[0] = 0.f;
      MV_FOR(i, 1, A) t0[i] = a[r * d + i - 1];
      pt_mu0<KIND, AMAX>(t0, t1, A, R, t2);
    }
    exp_map<KIND, AMAX>(t2, t1, A, R, t3);
    MV_FOR(i, 0, A) {
      o1[r * A + i] = t3[i];
      if (o2) o2[r * A + i] = t2[i];
    }
  } else if constexpr (OP == OP_ISAMPLE) {  // a = z[rows,A], b = at -> o1 = u[rows,A], o2 = v[rows,d]
    const int64_t ar = r % at_rows;
    MV_FOR(i, 0, A) {
      t0[i] = a[r * A + i];
      t1[i] = b[ar * A + i];
    }
    log_map<KIND, AMAX>(t0, t1, A, R, t2);
    MV_FOR(i, 0, A) o1[r * A + i] = t2[i];
    if constexpr (KIND == kEuclidean) {
      MV_FOR(i, 0, d) o2[r * d + i] = t2[i];
    } else if constexpr (KIND == kPoincare || KIND == kProjSphere) {
      float lam = (KIND == kPoincare) ? p_lambda<AMAX>(t1, A, 1.0f / (R * R)) : d_lambda<AMAX>(t1, A, 1.0f / (R * R));
      MV_FOR(i, 0, d) o2[r * d + i] = t2[i] * lam;
    } else {
      inv_pt_mu0<KIND, AMAX>(t2, t1, A, R, t3);
      MV_FOR(i, 1, A) o2[r * d + i - 1] = t3[i];
    }
  } else {  // OP_LOGDET: a = u (h,s) ; b = mu, c3 = z (p)
    if constexpr (KIND == kEuclidean) {
      o1[r] = 0.f;
    } else if constexpr (KIND == kPoincare || KIND == kProjSphere) {
      const int64_t ar = r % at_rows;
      MV_FOR(i, 0, A) {
        t0[i] = b[ar * A + i];
        t1[i] = c3[r * A + i];
      }
      o1[r] = (KIND == kPoincare) ? p_logdet<AMAX>(t0, t1, A, R) : d_logdet<AMAX>(t0, t1, A, R);
    } else {
      MV_FOR(i, 0, A) t0[i] = a[r * A + i];
      o1[r] = logdet_u<KIND, AMAX>(t0, A, R);
    }
  }
}

template <int OP, int DMAX>
__global__ __launch_bounds__(256) void k_prim(int kind, const float* a, const float* b, const float* c3, float* o1,
                                              float* o2, int64_t rows, int64_t at_rows, int d,
                                              const float* radius_param) {
  float rp = (kind == kEuclidean || !radius_param) ? 0.f : radius_param[0];
  kind = resolve_universal(kind, rp);  // for `u`, radius_param holds the curvature K
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    switch (kind) {
      case kEuclidean: prim_row<OP, kEuclidean, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
      case kHyperboloid: prim_row<OP, kHyperboloid, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
      case kSphere: prim_row<OP, kSphere, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
      case kProjSphere: prim_row<OP, kProjSphere, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
      default: prim_row<OP, kPoincare, DMAX>(a, b, c3, o1, o2, d, rp, r, at_rows); break;
    }
  }
}

template <int OP>
static int launch_prim(int kind, const float* a, const float* b, const float* c3, float* o1, float* o2, int64_t rows,
                       int64_t at_rows, int d, const float* rp, void* stream) {
  if (kind < 0 || kind >= kNumKinds) return fail(MVAE_E_BADARG, "unknown manifold kind%s (%lld)", "", kind);
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
  if ((kind == MVAE_POINCARE || kind == MVAE_PROJ_SPHERE || kind == MVAE_UNIVERSAL) && (!mu || !z))
    return fail(MVAE_E_BADARG, "logdet of a projected model needs mu and z%s", "");
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
// large, 16-byte aligned problems go to the LDS-tiled kernel (defined with the conv building blocks below)
static bool linear_forward_tiled(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K,
                                 int relu, hipStream_t s);
constexpr int64_t kTiledMinRows = 512;

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
__device__ __forceinline__ float wave_sum(float v);  // defined with the fused step below
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

extern "C" int mvae_loglik_reduce(const float* bce, const float* log_p, const float* log_q, float* log_px, float* mi,
                                  int n, int B, void* stream) {
  if (!bce || !log_p || !log_q || !log_px || !mi || n < 1 || B < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  hipLaunchKernelGGL(k_loglik_reduce, dim3(B), dim3(256), 0, (hipStream_t)stream, bce, log_p, log_q,
                     log_px, mi, n, B);
  LAUNCH_CHECK("loglik reduce launch");
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
  int64_t o_h, o_heads, o_z, o_hd, o_g, o_bce_part, o_kl, o_dhd, o_dz, o_dheads, o_dh, o_drpart, o_duals, o_total;
  int nt_d, nt_h, nt_b;  // 16-wide tile counts of D, H, B
};

static inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }
static inline int64_t up64(int64_t x) { return (x + 63) & ~(int64_t)63; }

// floats per (row, component, input direction) record of the dual workspace: {d kl, d z_0 .. d z_{A-1}}, A <= dmax + 1
static inline int dual_stride(int dmax_bucket) { return dmax_bucket + 2; }

static void carve(mvae_ctx* c, int dmax_bucket) {
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
  c->o_drpart = take(B * kMaxComp);  // [comp][B]
  // [B][heads_dim + ncomp][dual_stride]: every input direction of every component (radius directions included
  // whether or not they are trainable right now)
  c->o_duals = take(B * ((int64_t)d.heads_dim + d.ncomp) * dual_stride(dmax_bucket));
  c->o_total = o;
}

extern "C" int64_t mvae_workspace_floats(const mvae_model_desc* desc) {
  if (!desc) return -1;
  mvae_ctx tmp;
  tmp.d = *desc;
  int dmax = MVAE_MAX_TRUE_DIM;
  if (desc->comps && desc->ncomp >= 1 && desc->ncomp <= kMaxComp) {
    dmax = 1;
    for (int i = 0; i < desc->ncomp; ++i) dmax = desc->comps[i].true_dim > dmax ? desc->comps[i].true_dim : dmax;
  }
  carve(&tmp, bucket_of(dmax));
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
  carve(c, bucket_of(c->dmax));
  *out = c;
  return 0;
}

extern "C" void mvae_destroy(mvae_ctx* ctx) { delete ctx; }

extern "C" int mvae_set_radius_trainable(mvae_ctx* c, const uint8_t* trainable) {
  if (!c || !trainable) return fail(MVAE_E_BADARG, "null ctx / trainable%s", "");
  mvae_component_desc comps[kMaxComp];
  const int n = c->t.n;
  for (int i = 0; i < n; ++i) comps[i] = c->t.c[i];
  return fill_table(&c->t, comps, n, trainable, &c->dmax);
}

// ---------------------------------------------------------------------------------------------- Adam in the epilogue
// torch.optim.Adam, single-tensor CPU formulas, defaults betas=(0.9, 0.999), eps=1e-8:
//   m <- m + (1-b1)(g - m) ; v <- v*b2 + ((1-b2) g) g ; p <- p + (-lr/bc1 * m) / (sqrt(v)/sqrt(bc2) + eps)
// In the single-GPU step the update is applied by the workgroup that produced the gradient tile, in its epilogue,
// one launch after the last read of that weight (see the launch list at the top); a data-parallel run applies it in
// k_optim after the gradient all-reduce instead.
struct AdamArgs {
  float* p;
  float* m;
  float* v;
  const int* counters;  // counters[0] = number of this step (already advanced by launch 1)
  double lr;
};

__device__ __forceinline__ double pow_int(double base, int e) {  // base^e by squaring (e >= 0)
  double r = 1.0, b = base;
  while (e > 0) {
    if (e & 1) r *= b;
    b *= b;
    e >>= 1;
  }
  return r;
}

// thread 0 writes {-lr/bc1, sqrt(bc2)} to sh[0..1]; the caller's next __syncthreads publishes it
__device__ __forceinline__ void adam_consts(float* sh, const int* counters, double lr, int step_offset) {
  if (threadIdx.x == 0) {
    const int step = *(volatile const int*)&counters[0] + step_offset;
    const double bc1 = 1.0 - pow_int(0.9, step);
    const double bc2 = 1.0 - pow_int(0.999, step);
    sh[0] = (float)(-(lr / bc1));
    sh[1] = (float)sqrt(bc2);
  }
}

__device__ __forceinline__ void adam1(float& P, float G, float& M, float& V, float neg_step, float bc2s) {
  const float w1 = (float)(1.0 - 0.9), b2 = 0.999f, w2 = (float)(1.0 - 0.999);
  M = M + w1 * (G - M);
  V = V * b2 + (w2 * G) * G;
  P = P + (neg_step * M) / (sqrtf(V) / bc2s + 1e-8f);
}

// torch.nn.utils.clip_grad_norm_(curvature params, max_norm=1, norm_type=2) (vae.py:161-163): the coefficient
// min(1, 1 / (||g|| + 1e-6)) over the universal components' curvature gradients g (index order).
__device__ __forceinline__ float clip_coef(const CompTable& t, const float* g) {
  float n2 = 0.f;
  for (int j = 0; j < t.n; ++j)
    if (t.trainable[j] & 2) n2 += g[j] * g[j];
  return fminf(1.0f / (sqrtf(n2) + 1e-6f), 1.0f);
}

// ---------------------------------------------------------------------------------------------- step tile jobs
// 8-wave (512-thread) variants for the long contractions (K = 784 / 400): every wave issues ALL of its operand loads
// up front (<= 7 k-chunks per wave) and the eight partial tiles meet in LDS.
// Waves per workgroup of the wave-level dW tiles.  A CU's load path saturates with two tile workgroups (measured in
// launch 6: the ~50 CUs that received a second 4-wave workgroup finished at 4.6 us, the rest at 2.9 us), so launch 6
// is sized to put ONE tile workgroup on every CU: 1225 tiles / 5 waves = 250 workgroups for 256 CUs (6.8 -> 5.6 us
// together with the branch-free ragged tiles).  Launch 5 shares its CUs with the 128 row workgroups either way and
// measured better with 4-wave tile workgroups (6.3 vs 7.2 us).
constexpr int kTileWaves = 5;    // launch 6
constexpr int kTileWaves5 = 4;   // launch 5
constexpr int kW8 = 8;
__device__ __forceinline__ float reduce_tiles8(float (*red)[16][17], f32x4 acc) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, rbase = (lane >> 4) << 2;
  red[wave][rbase + 0][col] = acc[0];
  red[wave][rbase + 1][col] = acc[1];
  red[wave][rbase + 2][col] = acc[2];
  red[wave][rbase + 3][col] = acc[3];
  lds_barrier();
  float s = 0.f;
  if (tid < 256) {
    const int r = tid >> 4, c = tid & 15;
    s = ((red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c])) +
        ((red[4][r][c] + red[5][r][c]) + (red[6][r][c] + red[7][r][c]));
  }
  return s;
}

// 16-byte store with the write-through cache policy, through a raw buffer descriptor built from the (wave-uniform)
// base pointer; `idx` in floats (< 2^30).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(float* base, size_t idx, f32x4 v) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)(idx << 2), 0, /*aux: sc1*/ 16);
}

// dW tile per WAVE (+ optional Adam): out[p][q] = sum_m P[m][p] Q[m][q].  The batch contraction (K = B = 128) is short
// enough for one wave: 32 MFMA steps on two accumulators, all operand loads in flight at once, no LDS, no barrier.
// The four waves of a workgroup take four neighbouring q-tiles (they share the P operand through L1).
// The MFMA is issued with the operand roles swapped (A <- Q, B <- P), so that lane l ends up with the four
// CONSECUTIVE outputs out[p0 + (l&15)][q0 + 4*(l>>4) + 0..3]: gradient, parameter and both Adam moments move as one
// 16-byte access per lane each.
template <bool ADAM, bool FULL = false>
__device__ __forceinline__ void job_tn_wave(const float* P, int ldp, int NP, int pt, const float* Q, int ldq, int NQ,
                                            int qt, int Mrows, float* out, int ldo, const AdamArgs& aa) {
  if (qt * 16 >= NQ) return;
  MV_STAMP_B(16, MV_STAMP_BLK);
  const int lane = threadIdx.x & 63;
  const int pr = pt * 16 + (lane & 15), qc0 = qt * 16 + ((lane >> 4) << 2);
  const bool pok = pr < NP;
  const size_t idx = (size_t)pr * ldo + qc0;
  const bool vec = pok && (qc0 + 3 < NQ) && (ldo & 3) == 0 && aligned16(out) && (!ADAM || aligned16(aa.p));
  float p0[4] = {0.f, 0.f, 0.f, 0.f}, m0[4] = {0.f, 0.f, 0.f, 0.f}, v0[4] = {0.f, 0.f, 0.f, 0.f};
  float neg_step = 0.f, bc2s = 1.f;
  if (ADAM) {
    if (vec) {
      const float4 a = *reinterpret_cast<const float4*>(aa.p + idx);
      const float4 b = *reinterpret_cast<const float4*>(aa.m + idx);
      const float4 c = *reinterpret_cast<const float4*>(aa.v + idx);
      p0[0] = a.x; p0[1] = a.y; p0[2] = a.z; p0[3] = a.w;
      m0[0] = b.x; m0[1] = b.y; m0[2] = b.z; m0[3] = b.w;
      v0[0] = c.x; v0[1] = c.y; v0[2] = c.z; v0[3] = c.w;
    } else if (pok) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (qc0 + r < NQ) {
          p0[r] = aa.p[idx + r];
          m0[r] = aa.m[idx + r];
          v0[r] = aa.v[idx + r];
        }
    }
    // {-lr/bc1, sqrt(bc2)} of this step, published by launch 1 (k_enc_fwd)
    neg_step = reinterpret_cast<const float*>(aa.counters)[2];
    bc2s = reinterpret_cast<const float*>(aa.counters)[3];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_tn<32, FULL>(Q, ldq, NQ, qt * 16, P, ldp, NP, pt * 16, Mrows, 0, 1, acc);
  MV_STAMP_B(17, MV_STAMP_BLK);
  if (ADAM) {
#pragma unroll
    for (int r = 0; r < 4; ++r) adam1(p0[r], acc[r], m0[r], v0[r], neg_step, bc2s);
  }
  if (vec) {
    // Write-through (sc1) 16-byte stores: nothing in this launch reads these lines again, and every line left dirty
    // in the XCD's L2 has to be written back by the end-of-kernel release before the next launch may start -- with
    // plain stores the 5 MB of g/p/m/v of one weight matrix cost ~2 us of idle chip after launches 5 and 6.
    store16_wt(out, idx, acc);
    if (ADAM) {
      store16_wt(aa.p, idx, f32x4{p0[0], p0[1], p0[2], p0[3]});
      store16_wt(aa.m, idx, f32x4{m0[0], m0[1], m0[2], m0[3]});
      store16_wt(aa.v, idx, f32x4{v0[0], v0[1], v0[2], v0[3]});
    }
  } else if (pok) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (qc0 + r < NQ) {
        out[idx + r] = acc[r];
        if (ADAM) {
          aa.p[idx + r] = p0[r];
          aa.m[idx + r] = m0[r];
          aa.v[idx + r] = v0[r];
        }
      }
  }
  MV_STAMP_B(18, MV_STAMP_BLK);
}

// bias gradient (+ optional Adam): out[c] = sum_m Gm[m][c] for 16 columns; any block size that is a multiple of 16
template <bool ADAM>
__device__ __forceinline__ void job_colsum_opt(float* lds /*>= 32*17+2 floats*/, const float* Gm, int ld, int Mrows,
                                               int ncols, int c0, float* out, const AdamArgs& aa) {
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4, ng = blockDim.x >> 4;
  float* sh = lds + 32 * 17;
  float p0 = 0.f, m0 = 0.f, v0 = 0.f;
  const bool fin = (g == 0) && (c0 + c < ncols);
  if (ADAM) {
    if (fin) {
      p0 = aa.p[c0 + c];
      m0 = aa.m[c0 + c];
      v0 = aa.v[c0 + c];
    }
    if (threadIdx.x == 0) {  // published by launch 1 (k_enc_fwd)
      sh[0] = reinterpret_cast<const float*>(aa.counters)[2];
      sh[1] = reinterpret_cast<const float*>(aa.counters)[3];
    }
  }
  float s = 0.f;
  if (c0 + c < ncols) {
    const float* col = Gm + c0 + c;
    int m = g;
    for (; m + ng * 3 < Mrows; m += ng * 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = col[(size_t)(m + ng * u) * ld];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
    for (; m < Mrows; m += ng) s += col[(size_t)m * ld];
  }
  lds[g * 17 + c] = s;
  __syncthreads();
  if (fin) {
    float t = 0.f;
    for (int q = 0; q < ng; ++q) t += lds[q * 17 + c];
    out[c0 + c] = t;
    if (ADAM) {
      adam1(p0, t, m0, v0, sh[0], sh[1]);
      aa.p[c0 + c] = p0;
      aa.m[c0 + c] = m0;
      aa.v[c0 + c] = v0;
    }
  }
}

// XCD-aware tile assignment for the NT layers.  Workgroup L is observed to run on XCD L % 8 (MI355X_MICROARCH.md,
// "Workgroup dispatch"); each XCD has a private L2, so a weight row-block fetched by workgroups on all 8 XCDs crosses
// the fabric 8 times.  Column tiles (= weight row-blocks) are therefore dealt to XCDs: XCD k owns nt = k, k+8, ... and
// runs every row tile mt of those; the activation rows are the only operand every XCD fetches.  Placement only
// changes speed, never results.  Launch with grid = 8 * ceil(NT/8) * MT; returns false for the padding workgroups.
__device__ __forceinline__ bool xcd_tile(int NT, int MT, int* nt, int* mt, int L = blockIdx.x) {
  const int k = L & 7, s = L >> 3;
  *nt = k + 8 * (s / MT);
  *mt = s % MT;
  return *nt < NT;
}

// ---- 1: encoder layer (512 threads).  In the fused single-GPU step, workgroup (0,0) also advances the step counter.
template <bool FULL>
__global__ __launch_bounds__(512) void k_enc_fwd(const float* x, const float* W, const float* b, float* h, int B, int H,
                                                 int D, int* counters, int bump_step, double lr) {
  __shared__ float red[kW8][16][17];
  const int wave = threadIdx.x >> 6;
  int mt, nt;
  // Once per step: advance the counters and publish Adam's bias-correction scalars for the gradient epilogues of
  // launches 4-6 (double-precision pow / divide / sqrt: ~1 us for one lane -- done here by a padding workgroup of
  // the XCD-aware grid when there is one, so that it is off every critical path).
  MV_SPAN_BEGIN(0);
  const bool real = xcd_tile((H + 15) / 16, (B + 15) / 16, &nt, &mt);
  const bool has_pad = (((H + 15) / 16) & 7) != 0;
  if (threadIdx.x == 0 && (has_pad ? blockIdx.x == gridDim.x - 1 : blockIdx.x == 0)) {
    int step = counters[0];
    if (bump_step) counters[0] = ++step;
    counters[8] = counters[8] + 1;  // batch cursor of the device-side input pipeline (mvae_prepare_batch)
    if (bump_step) {
      const double bc1 = 1.0 - pow_int(0.9, step);
      const double bc2 = 1.0 - pow_int(0.999, step);
      reinterpret_cast<float*>(counters)[2] = (float)(-(lr / bc1));
      reinterpret_cast<float*>(counters)[3] = (float)sqrt(bc2);
    }
  }
  if (!real) return;
  const bool vx = aligned16(x) && (D & 3) == 0, vw = aligned16(W) && (D & 3) == 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_nt<7, FULL>(x, D, B, mt * 16, W, D, H, nt * 16, D, wave, kW8, vx, vw, acc);
  const float s = reduce_tiles8(red, acc);
  if (threadIdx.x < 256) {
    const int m = mt * 16 + (threadIdx.x >> 4), n = nt * 16 + (threadIdx.x & 15);
    if (FULL || (m < B && n < H)) {
      const float v = s + b[n];
      h[(size_t)m * H + n] = v < 0.f ? 0.f : v;  // torch.relu: NaN propagates
    }
  }
  MV_SPAN_END(0, 1);
}

// wave-level sum (all 64 lanes end up with the total): DPP row operations + one readlane, ~50 cycles, instead of six
// dependent ds_bpermute round trips through the LDS crossbar (~130 cycles each) that __shfl_xor lowers to.
__device__ __forceinline__ float wave_sum(float v) {
  int x = __float_as_int(v);
#define MV_DPP_ADD(CTRL, ROWMASK)                                                                       \
  x = __float_as_int(__int_as_float(x) +                                                                \
                     __int_as_float(__builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xF, true)));
  MV_DPP_ADD(0xB1, 0xF)   // quad_perm [1,0,3,2]
  MV_DPP_ADD(0x4E, 0xF)   // quad_perm [2,3,0,1]
  MV_DPP_ADD(0x141, 0xF)  // row_half_mirror
  MV_DPP_ADD(0x140, 0xF)  // row_mirror: every lane of a 16-lane row now holds the row sum
  MV_DPP_ADD(0x142, 0xA)  // row_bcast15 into rows 1 and 3
  MV_DPP_ADD(0x143, 0xC)  // row_bcast31 into rows 2 and 3: lane 63 holds the total
#undef MV_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// parts a head row is split into by the generic heads contraction of k_latent_fwd (256 threads = rows x parts)
__host__ __device__ inline int heads_parts(int NH) {
  const int p = NH >= 256 ? 1 : 256 / NH;
  return p > 8 ? 8 : p;
}

// ---- 2: heads + latent components + first decoder layer; ONE batch row per workgroup.  The phases are short and
// latency-bound, so rows are spread over as many CUs as possible, every global operand is requested in the first
// instructions of the kernel (one memory round trip), and the small reductions are wavefront shuffles.
// FAST: NH <= 16, Z <= 8, H <= 512 (operands of all phases are held in registers from the start).
template <int DMAX, bool FAST>
__global__ __launch_bounds__(512) void k_latent_fwd(CompTable t, const float* h, const float* Wh, const float* bh,
                                                    const float* eps, int eps_ld, const float* radii, const float* Wd0,
                                                    const float* bd0, float* heads, int ldh, float* z, int ldz,
                                                    float* z_user, float* kl, float* kl_user, float* hd, int B, int H,
                                                    int NH, int Z, float* duals) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // [H] the row of h, then [eps_dim] the row of eps
  __shared__ __attribute__((aligned(16))) float heads_s[kHeadsMax];
  __shared__ __attribute__((aligned(16))) float z_s[kHeadsMax];
  __shared__ mvae_component_desc desc_s[kMaxComp];  // per-lane indexed below: LDS, not the kernarg segment
  __shared__ float rad_s[kMaxComp];
  __shared__ signed char comp_at_s[4][kMaxComp];  // [wave][lane] -> component (or -1)
  __shared__ int ndir_s[kMaxComp];   // active input directions of component i (radius included iff trainable)
  __shared__ int first_s[kMaxComp];  // first record of component i inside a row of `duals`
  __shared__ int done_s;             // main waves that have finished their primal components
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t row = blockIdx.x;
  float* h_s = dyn;
  float* eps_s = dyn + ((H + 3) & ~3);
  const bool vec = aligned16(Wh) && (H & 3) == 0;
  MV_STAMP(0);
  MV_SPAN_BEGIN(1);

  // ---- dual waves (threads 256..511, launched iff duals != NULL).  Forward-mode derivatives need no upstream
  // gradient: d z / d(direction) and d kl / d(direction) of every (component, input direction) of this row depend only
  // on the head outputs, eps and the radii.  The ~500-instruction dependent dual chain (3.5 us for a lone lane)
  // therefore runs HERE, on waves 4..7, next to the primal lanes of waves 0..3 (wave 4+w takes the components placed
  // on wave w), instead of on the critical path of launch 5, which only contracts the stored records with dz.
  // Record layout: duals[row][first_dir(ci) + dir][{d kl, d z_0 .. d z_{A-1}}].
  if (tid >= 256) {
    // as many barriers as the main path executes up to "heads_s final": 2 in the prologue, then 2 (register-resident
    // path) or 2 per round of the generic heads contraction + 1
    int nbar = 4;
    if (!FAST) {
      const int per = 256 / heads_parts(NH);
      nbar = 2 + 2 * ((NH + per - 1) / per) + 1;
    }
    for (int i = 0; i < nbar; ++i) lds_barrier();
    const int w = wave - 4;
    int total = 0;
    for (int sidx = 0; sidx < kMaxComp; ++sidx) {
      const int ci = comp_at_s[w][sidx];
      if (ci < 0) break;
      total += ndir_s[ci];
    }
    constexpr int AM = DMAX + 1, DS = DMAX + 2;
    for (int base = 0; base < total; base += 64) {
      const int item = base + lane;
      if (item < total) {
        int rem = item, ci = comp_at_s[w][0], sidx = 0;
        while (rem >= ndir_s[ci]) {
          rem -= ndir_s[ci];
          ci = comp_at_s[w][++sidx];
        }
        const mvae_component_desc& c = desc_s[ci];
        float zd[AM];
        const float kld = comp_dual_dir<DMAX>(c, heads_s, eps_s, rad_s, rem, zd);
        float* rec = duals + ((size_t)row * (NH + t.n) + first_s[ci] + rem) * DS;
        const int A = ambient_dim(c.kind, c.true_dim);
        rec[0] = kld;
#pragma unroll
        for (int i = 0; i < AM; ++i)
          if (i < A) rec[1 + i] = zd[i];
      }
    }
    MV_SPAN_END_T(1, 2, 256, 1024);
    return;  // terminated waves do not count at the remaining barriers
  }

  // ---- request everything
  float hv[2] = {0.f, 0.f};
  float4 wf[4][2];
  float wd[2][8], bd[2] = {0.f, 0.f};
  float bhv = 0.f;
  if (FAST) {
    // Branch-free requests: an index past the end is clamped to a valid address and the value zeroed afterwards
    // (every `if (cond) load` costs a lone wave a taken/not-taken branch; ~30 of them made this prologue 2.5 us).
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = tid + 256 * u;
      const float v = h[row * H + (k < H ? k : 0)];
      hv[u] = k < H ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = wave + 4 * q;
      const int nn = n < NH ? n : 0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = lane * 4 + 256 * u;
        const bool ok = n < NH && k < H;
        float4 v = *reinterpret_cast<const float4*>(Wh + (size_t)nn * H + (k < H ? k : 0));
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        wf[q][u] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + 256 * u;
      const int cc = c < H ? c : 0;
      bd[u] = bd0[cc];
      if (Z == 8) {  // wave-uniform: the row of W_d0 is two 16-byte loads
        const float4 a = *reinterpret_cast<const float4*>(Wd0 + (size_t)cc * 8);
        const float4 b4 = *reinterpret_cast<const float4*>(Wd0 + (size_t)cc * 8 + 4);
        wd[u][0] = a.x; wd[u][1] = a.y; wd[u][2] = a.z; wd[u][3] = a.w;
        wd[u][4] = b4.x; wd[u][5] = b4.y; wd[u][6] = b4.z; wd[u][7] = b4.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = Wd0[(size_t)cc * Z + (j < Z ? j : 0)];
          wd[u][j] = j < Z ? v : 0.f;
        }
      }
    }
    bhv = bh[tid < NH ? tid : 0];
  }
  if (tid < eps_ld) eps_s[tid] = eps[row * eps_ld + tid];
  comp_at_s[tid >> 6][tid & 63] = -1;
  if (tid == 0) done_s = 0;
  lds_barrier();
  if (tid < t.n) {  // staged last so that its wait does not delay the issue of the loads above
    desc_s[tid] = t.c[tid];
    rad_s[tid] = radii[tid];
    ndir_s[tid] = t.dir_off[tid + 1] - t.dir_off[tid];
    first_s[tid] = t.first_dir[tid];
    comp_at_s[t.wave_of[tid]][t.lane_of[tid]] = (signed char)tid;
  }
  if (!FAST)
    for (int k = tid; k < H; k += 256) h_s[k] = h[row * H + k];
  if (FAST) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + 256 * u < H) h_s[tid + 256 * u] = hv[u];
  }
  lds_barrier();
  MV_STAMP(1);

  // ---- heads = h W_heads^T + b
  if (FAST) {
    float part[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // rows past NH were loaded as zeros: no branch, the four reductions interleave
      float p = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = lane * 4 + 256 * u;
        if (k < H) {
          const float4 xv = *reinterpret_cast<const float4*>(h_s + k);
          p = fmaf(xv.x, wf[q][u].x, p);
          p = fmaf(xv.y, wf[q][u].y, p);
          p = fmaf(xv.z, wf[q][u].z, p);
          p = fmaf(xv.w, wf[q][u].w, p);
        }
      }
      part[q] = p;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) part[q] = wave_sum(part[q]);
    const float mine = lane == 0 ? part[0] : lane == 1 ? part[1] : lane == 2 ? part[2] : part[3];
    if (lane < 4 && wave + 4 * lane < NH) heads_s[wave + 4 * lane] = mine;
    lds_barrier();
    if (tid < NH) {
      const float v = heads_s[tid] + bhv;
      heads[row * ldh + tid] = v;
      heads_s[tid] = v;  // same thread wrote nothing else here; published by the barrier below
    }
  } else {
    // generic: thread (n, part) streams its own stretch of row n of W_heads (all of its loads are independent);
    // the `P` partial sums of a row meet in LDS (z_s is free until the components write it)
    const int P = heads_parts(NH), per = 256 / P;
    for (int n0 = 0; n0 < NH; n0 += per) {
      const int nl = tid / P, part = tid - nl * P, n = n0 + nl;
      float p = 0.f;
      if (nl < per && n < NH) {
        if (vec) {
          const int H4 = H >> 2, chunk = (H4 + P - 1) / P;
          const int k0 = part * chunk, k1 = (k0 + chunk < H4) ? k0 + chunk : H4;
          const float4* wrow = reinterpret_cast<const float4*>(Wh + (size_t)n * H);
          const float4* hrow = reinterpret_cast<const float4*>(h_s);
#pragma unroll 8
          for (int k = k0; k < k1; ++k) {
            const float4 wv = wrow[k];
            const float4 xv = hrow[k];
            p = fmaf(xv.x, wv.x, p);
            p = fmaf(xv.y, wv.y, p);
            p = fmaf(xv.z, wv.z, p);
            p = fmaf(xv.w, wv.w, p);
          }
        } else {
          const int chunk = (H + P - 1) / P;
          const int k0 = part * chunk, k1 = (k0 + chunk < H) ? k0 + chunk : H;
#pragma unroll 8
          for (int k = k0; k < k1; ++k) p = fmaf(h_s[k], Wh[(size_t)n * H + k], p);
        }
      }
      z_s[tid] = p;
      lds_barrier();
      if (tid < per && n0 + tid < NH) {
        float sum = 0.f;
        for (int q = 0; q < P; ++q) sum += z_s[tid * P + q];
        heads_s[n0 + tid] = sum + bh[n0 + tid];
      }
      lds_barrier();
    }
    if (tid < NH) heads[row * ldh + tid] = heads_s[tid];
  }
  lds_barrier();
  MV_STAMP(2);

  // ---- latent components: one lane per component, placed by fill_table (kinds on different waves)
  {
    const int ci = comp_at_s[wave][lane];
    if (ci >= 0) {
      float klv;
      comp_fwd_row<DMAX>(desc_s[ci], heads_s, eps_s, rad_s, z_s, z + row * ldz, &klv, nullptr, nullptr, nullptr,
                         nullptr);
      kl[(size_t)ci * B + row] = klv;
      if (kl_user) kl_user[(size_t)ci * B + row] = klv;
    }
  }
  // The four main waves meet on an LDS counter, not on s_barrier: a hardware barrier would also wait for the dual
  // waves, which are still in the middle of their (longer) chains.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(&done_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(&done_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  MV_STAMP(3);
  if (z_user && tid < Z) z_user[row * Z + tid] = z_s[tid];

  // ---- first decoder layer: hd = relu(z W_d0^T + b)   (K = Z is tiny)
  if (FAST) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + 256 * u;
      if (c < H) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < Z) acc = fmaf(z_s[j], wd[u][j], acc);
        acc += bd[u];
        hd[row * H + c] = acc < 0.f ? 0.f : acc;
      }
    }
  } else {
    const bool vz = (Z & 3) == 0 && aligned16(Wd0);
    for (int c = tid; c < H; c += 256) {
      const float* w = Wd0 + (size_t)c * Z;
      float acc = 0.f;
      if (vz) {
        const float4* w4 = reinterpret_cast<const float4*>(w);
        const float4* z4 = reinterpret_cast<const float4*>(z_s);
#pragma unroll 4
        for (int j = 0; j < (Z >> 2); ++j) {
          const float4 a = w4[j], zz = z4[j];
          acc = fmaf(zz.x, a.x, acc);
          acc = fmaf(zz.y, a.y, acc);
          acc = fmaf(zz.z, a.z, acc);
          acc = fmaf(zz.w, a.w, acc);
        }
      } else {
#pragma unroll 4
        for (int j = 0; j < Z; ++j) acc = fmaf(z_s[j], w[j], acc);
      }
      acc += bd0[c];
      hd[row * H + c] = acc < 0.f ? 0.f : acc;
    }
  }
  MV_STAMP(4);
  MV_SPAN_END(1, 1);
}

// ---- 3: output layer + BCE-with-logits + its gradient (512 threads)
template <bool FULL>
__global__ __launch_bounds__(512) void k_dec1_fwd(const float* hd, const float* W, const float* b, const float* x,
                                                  float* g, float* bce_part, float* logits_user, int B, int H, int D) {
  __shared__ float red[kW8][16][17];
  const int wave = threadIdx.x >> 6;
  int mt, nt;
  MV_SPAN_BEGIN(2);
  if (!xcd_tile((D + 15) / 16, (B + 15) / 16, &nt, &mt)) return;
  const int m = mt * 16 + ((threadIdx.x & 255) >> 4), n = nt * 16 + (threadIdx.x & 15);
  const bool ok = threadIdx.x < 256 && m < B && n < D;
  float tv = 0.f, bias = 0.f;
  if (ok) {  // epilogue operands requested up front
    tv = x[(size_t)m * D + n];
    bias = b[n];
  }
  const bool v1 = aligned16(hd) && (H & 3) == 0, v2 = aligned16(W) && (H & 3) == 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_nt<4, FULL>(hd, H, B, mt * 16, W, H, D, nt * 16, H, wave, kW8, v1, v2, acc);
  const float s = reduce_tiles8(red, acc);
  if (threadIdx.x >= 256) return;
  float loss = 0.f;
  if (ok) {
    const float y = s + bias;
    // F.binary_cross_entropy_with_logits (image_reconstruction.py:81-82): (1-t)*y - log_sigmoid(y)
    const float e = expf(-fabsf(y));
    const float log_sig = fminf(y, 0.f) - mvf::log1p_pos(e);
    loss = (1.f - tv) * y - log_sig;
    const float sig = (y >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
    g[(size_t)m * D + n] = sig - tv;  // d(sum bce)/d(logit)
    if (logits_user) logits_user[(size_t)m * D + n] = y;
  }
  // sum over the tile's 16 columns: the 16 lanes of one row are contiguous
  loss += __shfl_xor(loss, 8, 16);
  loss += __shfl_xor(loss, 4, 16);
  loss += __shfl_xor(loss, 2, 16);
  loss += __shfl_xor(loss, 1, 16);
  if ((threadIdx.x & 15) == 0 && m < B) bce_part[(size_t)nt * B + m] = loss;
  MV_SPAN_END(2, 1);
}

// ---- 4: dhd = (g W_logits) * [hd > 0] ; db_logits (+Adam) ; step statistics   (512 threads)
template <bool ADAM, bool FULL>
__global__ __launch_bounds__(512) void k_dec1_bwd(const float* g, const float* hd, const float* W, float* db,
                                                  float* dhd, const float* bce_part, const float* kl, float* bce_user,
                                                  float* stats, float beta, int B, int H, int D, int ncomp, int n_dhd,
                                                  int n_db, AdamArgs ab) {
  __shared__ float red[kW8][16][17];
  int b = blockIdx.x;
  const int ntH = (H + 15) / 16, ntD = (D + 15) / 16;
  MV_SPAN_BEGIN(3);
  if (b < n_dhd) {
    const int mt = b / ntH, nt = b % ntH;
    const int wave = threadIdx.x >> 6;
    const int m = mt * 16 + ((threadIdx.x & 255) >> 4), n = nt * 16 + (threadIdx.x & 15);
    const bool ok = threadIdx.x < 256 && m < B && n < H;
    float mask = 0.f;
    if (ok) mask = hd[(size_t)m * H + n];
    const bool vg = aligned16(g) && (D & 3) == 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_nn<7, FULL>(g, D, B, mt * 16, W, H, H, nt * 16, D, wave, kW8, vg, acc);
    const float s = reduce_tiles8(red, acc);
    if (ok) dhd[(size_t)m * H + n] = (mask > 0.f) ? s : 0.f;
    MV_SPAN_END(3, 1);
    return;
  }
  b -= n_dhd;
  if (b < n_db) {
    job_colsum_opt<ADAM>(&red[0][0][0], g, D, B, D, b * kColsPerBlock, db, ab);
    MV_SPAN_END(3, 2);
    return;
  }
  // statistics block (BatchStats, stats.py:144-212): sums over the batch of bce, kl_i, elbo
  float* sm = &red[0][0][0];  // >= 512 floats
  const int tid = threadIdx.x, nthr = blockDim.x;
  float bce_acc = 0.f, elbo_acc = 0.f;
  for (int r = tid; r < B; r += nthr) {
    float bce = 0.f;
    int nt = 0;
    for (; nt + 7 < ntD; nt += 8) {  // 8 loads in flight, added in index order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = bce_part[(size_t)(nt + u) * B + r];
#pragma unroll
      for (int u = 0; u < 8; ++u) bce += v[u];
    }
    for (; nt < ntD; ++nt) bce += bce_part[(size_t)nt * B + r];
    if (bce_user) bce_user[r] = bce;
    float klr = kl[r];
    int i = 1;
    for (; i + 7 < ncomp; i += 8) {  // 8 loads in flight, added in index order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = kl[(size_t)(i + u) * B + r];
#pragma unroll
      for (int u = 0; u < 8; ++u) klr += v[u];
    }
    for (; i < ncomp; ++i) klr += kl[(size_t)i * B + r];
    bce_acc += bce;
    elbo_acc += (-bce - beta * klr);
  }
  // block-wide sum: wavefront shuffles, then the (<= 8) wave totals meet in LDS -- two barriers per reduction
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (nthr >> 6); ++w) r += sm[w];
    __syncthreads();
    return r;
  };
  const float bce_sum = block_sum(bce_acc);
  const float elbo_sum = block_sum(elbo_acc);
  const int last = 4 + ncomp;
  // per-component KL sums: one wave per component (waves take components round-robin), rows summed in lane order
  {
    const int wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;
    for (int i = wave; i < ncomp; i += nw) {
      float a = 0.f;
      for (int r = lane; r < B; r += 64) a += kl[(size_t)i * B + r];
      a = wave_sum(a);
      if (lane == 0) {
        stats[4 + i] += a;
        stats[last + 4 + i] = a;
        sm[16 + i] = a;
      }
    }
  }
  __syncthreads();
  float kl_total = 0.f;
  if (tid == 0) {
    for (int i = 0; i < ncomp; ++i) kl_total += sm[16 + i];
    stats[0] += bce_sum;
    stats[1] += kl_total;
    stats[2] += elbo_sum;
    stats[3] += 1.f;
    stats[last + 0] = bce_sum;
    stats[last + 1] = kl_total;
    stats[last + 2] = elbo_sum;
    stats[last + 3] = 1.f;
  }
  MV_SPAN_END(3, 3);
}

// ---- 5: backward through the first decoder layer, the latent components and the heads (one batch row per
// workgroup) ; dW_logits = g^T hd (+Adam: W_logits was last read by launch 4)
template <int DMAX, bool FAST, bool ADAM>  // FAST also implies tile-aligned B, H, D (checked on the host)
__global__ __launch_bounds__(64 * kTileWaves5) void k_latent_bwd(CompTable t, const float* dhd, const float* Wd0, int ldh,
                                                    const float* h, const float* Wh, float* dheads, float* dh,
                                                    float* drpart, const float* g, const float* hd, float* dWl,
                                                    float beta, int B, int H, int D, int NH, int Z, int n_rows,
                                                    AdamArgs awl, const float* duals) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // [H] dhd row | [1024] dz partials
  __shared__ float red[4][16][17];
  __shared__ float sh2[2];
  __shared__ float dz_s[kHeadsMax];
  __shared__ float dheads_s[kHeadsMax];
  __shared__ mvae_component_desc desc_s[kMaxComp];
  __shared__ int doff_s[kMaxComp + 1];  // prefix of the ACTIVE input directions (radius included iff trainable)
  __shared__ int first_s[kMaxComp + 1];  // first record of component i inside a row of `duals`
  int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  MV_SPAN_BEGIN(4);
  if (b >= n_rows) {  // dW_logits[D,H] tile
    b -= n_rows;
    const int ntHg = ((H + 15) / 16 + kTileWaves5 - 1) / kTileWaves5;
    if (FAST) job_tn_wave<ADAM, true>(g, D, D, b / ntHg, hd, H, H, (b % ntHg) * kTileWaves5 + wave, B, dWl, H, awl);
    else job_tn_wave<ADAM, false>(g, D, D, b / ntHg, hd, H, H, (b % ntHg) * kTileWaves5 + wave, B, dWl, H, awl);
    MV_SPAN_END(4, 2);
    return;
  }
  if (tid >= 256) return;  // the row path is written for 4 waves
  const size_t row = b;
  MV_STAMP(8);
  const int H4 = (H + 3) & ~3;
  float* dhd_s = dyn;
  float* part = dyn + H4;

  // ---- request everything; the row's own operands (written by the previous launch) first: loads retire in order,
  // so what is needed first must be asked for first
  int ZP = 1, zsh = 0;  // ZP = next power of two >= Z: the (slice, j) split of the thread index is shifts and masks
  while (ZP < Z) {
    ZP <<= 1;
    ++zsh;
  }
  const int nsl = 256 >> zsh;
  const int zj = tid & (ZP - 1), sl = tid >> zsh;
  float dhd_r[2] = {0.f, 0.f};  // FAST: H <= 512
  float wz[16];     // FAST: this thread's W_d0 column slice (H/nsl <= 16 entries)
  float wh[2][16];  // FAST: W_heads[:, c] for the two columns c of this thread
  float hm[2] = {0.f, 0.f};
  // Branch-free requests (an index past the end is clamped to a valid address, the value zeroed afterwards): every
  // `if (cond) load` costs an exec-mask branch and, worse, lets the compiler put a full `s_waitcnt vmcnt(0)` inside
  // it -- the guarded version of this prologue spent ~3 us in serialized round trips.  32-bit unsigned offsets keep
  // the addresses in the scalar-base + vector-offset form.
  const unsigned rowH = (unsigned)row * (unsigned)H;
  if (FAST) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + 256 * u;
      const float v = dhd[rowH + (unsigned)(c < H ? c : 0)];
      dhd_r[u] = c < H ? v : 0.f;
    }
  }
  // the component table (kernarg segment, indexed per lane): tiny, but a wait on the LAST load issued is a wait on
  // every load before it, so it goes ahead of the bulk weight requests
  const int tci = tid <= t.n ? tid : 0;
  const mvae_component_desc desc_r = t.c[tci < t.n ? tci : 0];
  const int doff_r = t.dir_off[tci];
  const int first_r = t.first_dir[tci < t.n ? tci : 0];
  __builtin_amdgcn_sched_barrier(0);  // keep the requests above ahead of the bulk below
  if (FAST) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = sl + q * nsl;
      const bool ok = zj < Z && c < H;
      const float v = Wd0[ok ? (unsigned)(c * Z + zj) : 0u];
      wz[q] = ok ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + 256 * u;
      const unsigned cc = (unsigned)(c < H ? c : 0);
      const float hv = h[rowH + cc];
      hm[u] = c < H ? hv : 0.f;
      // rows n >= NH re-read row 0 and are multiplied by dheads_s[n] = 0 below: selecting on the (uniform) n < NH
      // here would turn every load into a scalar branch with its own wait
#pragma unroll
      for (int n = 0; n < 16; ++n) wh[u][n] = Wh[(unsigned)(n < NH ? n : 0) * (unsigned)H + cc];
    }
  }
  if (tid <= t.n) {
    if (tid < t.n) desc_s[tid] = desc_r;
    doff_s[tid] = doff_r;
    first_s[tid] = first_r;
  }
  if (FAST) {
    if (tid < 16) dheads_s[tid] = 0.f;  // entries [NH, 16) stay zero (see the W_heads requests above)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + 256 * u;
      if (c < H) dhd_s[c] = dhd_r[u];
    }
  } else {
    for (int k = tid; k < H; k += 256) dhd_s[k] = dhd[row * H + k];
  }
  // generic 16-byte path of dz: thread (slice s2, column quad j4) owns rows c = s2, s2 + nslv, ... of W_d0; the
  // first 24 of them are requested here, ahead of the barrier (one round trip instead of one per 4 rows)
  constexpr int kDzB = 24;
  const bool dz_vec = !FAST && (Z & 3) == 0 && aligned16(Wd0);
  const int nj4 = dz_vec ? (Z >> 2) : 1, nslv = 256 / (nj4 > 256 ? 256 : nj4);
  const int j4 = tid % nj4, s2 = tid / nj4;
  float4 wzv[kDzB];
  if (dz_vec) {
#pragma unroll
    for (int u = 0; u < kDzB; ++u) {
      const int c = s2 + u * nslv;
      wzv[u] = *reinterpret_cast<const float4*>(Wd0 + ((s2 < nslv && c < H) ? (size_t)c * Z + 4 * j4 : 0));
    }
  }
  lds_barrier();
  MV_STAMP(9);

  // ---- this thread's dual record (written by launch 3): thread k owns the k-th active direction of the row; the
  // request is in flight while dz is reduced
  constexpr int DS = DMAX + 2;
  const int total = doff_s[t.n];
  const size_t rec0 = (size_t)row * (NH + t.n);
  float du[DS];
  int my_ci = 0, my_dir = 0;
  {  // the first (usually only) item of this thread
    const int gi = tid < total ? tid : 0;
    while (gi >= doff_s[my_ci + 1]) ++my_ci;
    my_dir = gi - doff_s[my_ci];
    const float* rec = duals + (rec0 + first_s[my_ci] + my_dir) * DS;
#pragma unroll
    for (int i = 0; i < DS; ++i) du[i] = rec[i];
  }

  // ---- dz[j] = sum_c dhd[c] W_d0[c][j]: thread (slice, j) accumulates a strided slice of c; wave 3 adds the slices
  // in slice order
  {
    float p = 0.f;
    if (FAST) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // wz[q] = 0 past the end: no guard
        const int c = sl + q * nsl;
        p = fmaf(dhd_s[c < H ? c : 0], wz[q], p);
      }
    } else if (zj < Z && !dz_vec) {
#pragma unroll 4
      for (int c = sl; c < H; c += nsl) p = fmaf(dhd_s[c], Wd0[(size_t)c * Z + zj], p);
    }
    if (FAST) {
      // ZP <= 8: the slices of one wave are the lanes with equal (lane & (ZP-1)): butterfly over the upper lane bits,
      // then the four waves' sums meet in LDS (fixed order)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        if (off >= ZP) p += __shfl_xor(p, off);
      if (lane < ZP) part[wave * ZP + lane] = p;
      lds_barrier();
      if (tid < Z) dz_s[tid] = (part[tid] + part[ZP + tid]) + (part[2 * ZP + tid] + part[3 * ZP + tid]);
    } else if (dz_vec) {
      if (s2 < nslv && j4 < nj4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < kDzB; ++u) {
          const int c = s2 + u * nslv;
          const float dv = c < H ? dhd_s[c < H ? c : 0] : 0.f;  // rows past the end were clamped to row 0
          a.x = fmaf(dv, wzv[u].x, a.x);
          a.y = fmaf(dv, wzv[u].y, a.y);
          a.z = fmaf(dv, wzv[u].z, a.z);
          a.w = fmaf(dv, wzv[u].w, a.w);
        }
#pragma unroll 4
        for (int c = s2 + kDzB * nslv; c < H; c += nslv) {
          const float4 w = *reinterpret_cast<const float4*>(Wd0 + (size_t)c * Z + 4 * j4);
          const float dv = dhd_s[c];
          a.x = fmaf(dv, w.x, a.x);
          a.y = fmaf(dv, w.y, a.y);
          a.z = fmaf(dv, w.z, a.z);
          a.w = fmaf(dv, w.w, a.w);
        }
        *reinterpret_cast<float4*>(part + s2 * Z + 4 * j4) = a;  // nslv * Z <= 1024 floats
      }
      lds_barrier();
      for (int j = tid; j < Z; j += 256) {
        float tot = 0.f;
        for (int q = 0; q < nslv; ++q) tot += part[q * Z + j];
        dz_s[j] = tot;
      }
    } else {
      part[tid] = p;
      lds_barrier();
      if (wave == 3) {
        for (int j = lane; j < Z; j += 64) {
          float tot = 0.f;
          for (int q = 0; q < nsl; ++q) tot += part[q * ZP + j];
          dz_s[j] = tot;
        }
      }
    }
    lds_barrier();
  }
  MV_STAMP(10);
  // ---- d(loss)/d(direction) = beta * d kl + <dz, d z>: one record per (component, input direction)
  for (int gi = tid; gi < total; gi += 256) {
    if (gi >= 256) {  // more than 256 active directions: further items are fetched on demand
      my_ci = 0;
      while (gi >= doff_s[my_ci + 1]) ++my_ci;
      my_dir = gi - doff_s[my_ci];
      const float* rec = duals + (rec0 + first_s[my_ci] + my_dir) * DS;
#pragma unroll
      for (int i = 0; i < DS; ++i) du[i] = rec[i];
    }
    const mvae_component_desc& c = desc_s[my_ci];
    const int A = ambient_dim(c.kind, c.true_dim);
    float gv = beta * du[0];
#pragma unroll
    for (int i = 0; i < DMAX + 1; ++i)
      if (i < A) gv += dz_s[c.z_col + i] * du[1 + i];
    if (my_dir < c.true_dim) dheads_s[c.mean_col + my_dir] = gv;
    else if (my_dir < c.true_dim + c.logvar_dim) dheads_s[c.logvar_col + (my_dir - c.true_dim)] = gv;
    else drpart[(size_t)my_ci * B + row] = gv;
  }
  lds_barrier();
  MV_STAMP(11);
  if (tid < NH) dheads[row * ldh + tid] = dheads_s[tid];
  // ---- dh = (dheads W_heads) * [h > 0]   (K = NH is small)
  if (FAST) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + 256 * u;
      if (c < H) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) acc = fmaf(dheads_s[n], wh[u][n], acc);
        dh[row * H + c] = (hm[u] > 0.f) ? acc : 0.f;
      }
    }
  } else if ((H & 3) == 0 && H <= 1024 && aligned16(Wh) && aligned16(h) && aligned16(dh)) {
    // thread (column quad c4, row group ng): rows n = ng, ng + G, ... of W_heads as 16-byte loads, 20 in flight;
    // the G partial sums of a quad meet in LDS and are added in group order
    const int nq = H >> 2, G = (256 / nq) < 1 ? 1 : ((256 / nq) > 8 ? 8 : 256 / nq);
    const int c4 = tid % nq, ng = tid / nq;
    const bool act = ng < G;
    float4 hmv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < nq) hmv = *reinterpret_cast<const float4*>(h + row * H + 4 * tid);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int nb = ng; nb < NH; nb += 20 * G) {
      float4 w[20];
#pragma unroll
      for (int u = 0; u < 20; ++u) {
        const int n = nb + u * G;
        w[u] = *reinterpret_cast<const float4*>(Wh + ((act && n < NH) ? (size_t)n * H + 4 * c4 : 0));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 20; ++u) {
        const int n = nb + u * G;
        const float dv = (act && n < NH) ? dheads_s[n < NH ? n : 0] : 0.f;
        a.x = fmaf(dv, w[u].x, a.x);
        a.y = fmaf(dv, w[u].y, a.y);
        a.z = fmaf(dv, w[u].z, a.z);
        a.w = fmaf(dv, w[u].w, a.w);
      }
    }
    if (act) *reinterpret_cast<float4*>(part + ((size_t)ng * nq + c4) * 4) = a;  // G * H <= 1024 floats ... see below
    lds_barrier();
    if (tid < nq) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = 0; q < G; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)q * nq + tid) * 4);
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
      }
      tot.x = hmv.x > 0.f ? tot.x : 0.f;
      tot.y = hmv.y > 0.f ? tot.y : 0.f;
      tot.z = hmv.z > 0.f ? tot.z : 0.f;
      tot.w = hmv.w > 0.f ? tot.w : 0.f;
      *reinterpret_cast<float4*>(dh + row * H + 4 * tid) = tot;
    }
  } else {
    for (int c = tid; c < H; c += 256) {
      float acc = 0.f;
#pragma unroll 8
      for (int n = 0; n < NH; ++n) acc = fmaf(dheads_s[n], Wh[(size_t)n * H + c], acc);
      const size_t o = row * H + c;
      dh[o] = (h[o] > 0.f) ? acc : 0.f;
    }
  }
  MV_STAMP(12);
  MV_SPAN_END(4, 1);
}

// ---- 6: dW_e0, dW_heads, dW_d0, their biases (+Adam) ; radius gradients (+SGD)
template <bool ADAM, bool FULL>
__global__ __launch_bounds__(64 * kTileWaves) void k_enc_bwd(CompTable t, const float* dh, const float* x, const float* dheads,
                                                 int ldh, const float* h, const float* dhd, const float* z, int ldz,
                                                 const float* drpart, float* G, float* P, int B, int H, int D, int NH,
                                                 int Z, int n_we0, int n_wh, int n_wd0, int n_be0, int n_bh, int n_bd0,
                                                 int64_t off_w_e0, int64_t off_b_e0, int64_t off_w_heads,
                                                 int64_t off_b_heads, int64_t off_w_d0, int64_t off_b_d0, AdamArgs base,
                                                 double curv_lr, int do_curv) {
  __shared__ float red[4][16][17];
  __shared__ float sh2[2];
  int b = blockIdx.x;
  MV_SPAN_BEGIN(5);
  auto at = [&](int64_t off) {
    AdamArgs a = base;
    a.p += off;
    a.m += off;
    a.v += off;
    return a;
  };
  // Workgroup order: the short jobs first, the 250 dW_e0 tile workgroups last.  The grid has ~80 more workgroups than
  // the chip has CUs, so the last ones dispatched share a CU with the first ones: sharing with a short job costs a tile
  // workgroup little, sharing with another tile workgroup (or a short job sharing with one) was the kernel's tail.
  if (b == 0) {
    // radius gradients: sum over the batch rows of the per-row terms of launch 5 (fixed order: deterministic), and in
    // the fused step torch.optim.SGD(lr=curv_lr) on the trainable radii: param.add_(grad, alpha=-lr)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* gsh = &red[0][0][0];  // per-component batch sums
    if (tid < kRadiiRegion) {
      G[tid] = 0.f;
      gsh[tid] = 0.f;
    }
    __syncthreads();
    for (int ci = wave; ci < t.n; ci += (int)(blockDim.x >> 6)) {
      if (!t.trainable[ci]) continue;
      float s = 0.f;
      for (int r = lane; r < B; r += 64) s += drpart[(size_t)ci * B + r];
      s = wave_sum(s);
      if (lane == 0) gsh[ci] = s;
    }
    __syncthreads();
    if (tid < t.n && t.trainable[tid]) {
      float s = gsh[tid];
      if (ADAM && (t.trainable[tid] & 2)) s *= clip_coef(t, gsh);  // vae.py:161-163 (fused step; else k_optim clips)
      G[tid] = s;
      if (ADAM && do_curv) P[tid] = P[tid] + (float)(-curv_lr) * s;
    }
    MV_SPAN_END(5, 7);
    return;
  }
  b -= 1;
  if (b < n_bd0) {
    job_colsum_opt<ADAM>(&red[0][0][0], dhd, H, B, H, b * kColsPerBlock, G + off_b_d0, at(off_b_d0));
    MV_SPAN_END(5, 6);
    return;
  }
  b -= n_bd0;
  if (b < n_be0) {
    job_colsum_opt<ADAM>(&red[0][0][0], dh, H, B, H, b * kColsPerBlock, G + off_b_e0, at(off_b_e0));
    MV_SPAN_END(5, 4);
    return;
  }
  b -= n_be0;
  if (b < n_bh) {
    job_colsum_opt<ADAM>(&red[0][0][0], dheads, ldh, B, NH, b * kColsPerBlock, G + off_b_heads, at(off_b_heads));
    MV_SPAN_END(5, 5);
    return;
  }
  b -= n_bh;
  if (b < n_wd0) {  // dW_d0[H,Z] = dhd^T z
    const int ntZg = ((Z + 15) / 16 + kTileWaves - 1) / kTileWaves;
    job_tn_wave<ADAM>(dhd, H, H, b / ntZg, z, ldz, Z, (b % ntZg) * kTileWaves + (threadIdx.x >> 6), B, G + off_w_d0, Z,
                      at(off_w_d0));
    MV_SPAN_END(5, 3);
    return;
  }
  b -= n_wd0;
  if (b < n_wh) {  // dW_heads[NH,H] = dheads^T h
    const int ntHg = ((H + 15) / 16 + kTileWaves - 1) / kTileWaves;
    job_tn_wave<ADAM>(dheads, ldh, NH, b / ntHg, h, H, H, (b % ntHg) * kTileWaves + (threadIdx.x >> 6), B, G + off_w_heads, H,
                      at(off_w_heads));
    MV_SPAN_END(5, 2);
    return;
  }
  b -= n_wh;
  {  // dW_e0[H,D] = dh^T x
    const int ntDg = ((D + 15) / 16 + kTileWaves - 1) / kTileWaves;
    job_tn_wave<ADAM, FULL>(dh, H, H, b / ntDg, x, D, D, (b % ntDg) * kTileWaves + (threadIdx.x >> 6), B, G + off_w_e0, D,
                            at(off_w_e0));
    MV_SPAN_END(5, 1);
  }
}

// ---- 7 (data-parallel / two-call path only): fused optimizer over the flat buffer after the gradient all-reduce
__global__ __launch_bounds__(256) void k_optim(CompTable t, float* p, float* g, float* m, float* v, int n4,
                                               int* counters, double lr, double curv_lr, int do_curv) {
  __shared__ float sh[2];
  __shared__ float gsh[kMaxComp];
  const int tid = threadIdx.x;
  adam_consts(sh, counters, lr, 1);
  if (blockIdx.x == 0 && tid < t.n) gsh[tid] = g[tid];
  __syncthreads();
  const float neg_step = sh[0], bc2s = sh[1];
  const int i4 = blockIdx.x * 256 + tid + kRadiiRegion / 4;
  if (i4 < n4) {
    float4 pp = reinterpret_cast<float4*>(p)[i4];
    const float4 gg = reinterpret_cast<const float4*>(g)[i4];
    float4 mm = reinterpret_cast<float4*>(m)[i4];
    float4 vv = reinterpret_cast<float4*>(v)[i4];
    adam1(pp.x, gg.x, mm.x, vv.x, neg_step, bc2s);
    adam1(pp.y, gg.y, mm.y, vv.y, neg_step, bc2s);
    adam1(pp.z, gg.z, mm.z, vv.z, neg_step, bc2s);
    adam1(pp.w, gg.w, mm.w, vv.w, neg_step, bc2s);
    // write-through, as in the tile epilogues: 7.6 MB that nobody in this launch reads again
    store16_wt(p, (size_t)i4 * 4, f32x4{pp.x, pp.y, pp.z, pp.w});
    store16_wt(m, (size_t)i4 * 4, f32x4{mm.x, mm.y, mm.z, mm.w});
    store16_wt(v, (size_t)i4 * 4, f32x4{vv.x, vv.y, vv.z, vv.w});
  }
  if (blockIdx.x == 0 && tid < t.n && t.trainable[tid]) {
    float gv = gsh[tid];
    if (t.trainable[tid] & 2) {  // universal curvature: clipped (after the all-reduce), written back like .grad
      gv *= clip_coef(t, gsh);
      g[tid] = gv;
    }
    if (do_curv) p[tid] = p[tid] + (float)(-curv_lr) * gv;  // SGD: param.add_(grad, alpha=-lr)
  }
  // The last workgroup to arrive advances the step counter.  Every other workgroup consumed counters[0] before its
  // own arrival (the value fed the __syncthreads above), so no fence is needed: the plain stores below only have to
  // be visible to the NEXT launch.  Arrivals are counted on 16 group words (counters[16..31]) first (one hot word
  // would serialise ~600 device-scope atomics at ~12 ns each); the group-completing workgroups meet on counters[1].
  if (tid == 0) {
    constexpr int NG = 16;
    const int grp = blockIdx.x % NG;
    const int gsize = ((int)gridDim.x - grp + NG - 1) / NG;
    if (atomicAdd(&counters[16 + grp], 1) == gsize - 1) {
      counters[16 + grp] = 0;
      const int ngroups = (int)gridDim.x < NG ? (int)gridDim.x : NG;
      if (atomicAdd(&counters[1], 1) == ngroups - 1) {
        counters[1] = 0;
        counters[0] = counters[0] + 1;
      }
    }
  }
}

// fused = single-GPU step (Adam/SGD in the gradient epilogues, no k_optim); otherwise gradients only.
static int step_impl(mvae_ctx* c, const float* x, const float* eps, float beta, bool fused, int do_curv,
                     int want_outputs, float* logits, float* concat_z, float* bce, float* kl, void* stream,
                     hipEvent_t* ev) {
  int ki = 0;  // launch index; with `ev` != NULL launch k is bracketed by ev[2k] (start) / ev[2k+1] (stop)
#define STEP_LAUNCH(KERN, GRID, BLOCK, LDS, ...)                                                              \
  do {                                                                                                         \
    if (ev) hipExtLaunchKernelGGL(KERN, GRID, BLOCK, LDS, s, ev[2 * ki], ev[2 * ki + 1], 0, __VA_ARGS__);      \
    else hipLaunchKernelGGL(KERN, GRID, BLOCK, LDS, s, __VA_ARGS__);                                          \
    ++ki;                                                                                                      \
  } while (0)
  if (!c || !x || !eps) return fail(MVAE_E_BADARG, "null pointer%s", "");
  const mvae_model_desc& d = c->d;
  const int B = d.batch, H = d.h_dim, D = d.in_dim, NH = d.heads_dim, Z = d.z_dim;
  hipStream_t s = (hipStream_t)stream;
  float* ws = d.workspace;
  float *h = ws + c->o_h, *heads = ws + c->o_heads, *z = ws + c->o_z, *hd = ws + c->o_hd, *g = ws + c->o_g,
        *bce_part = ws + c->o_bce_part, *klw = ws + c->o_kl, *dhd = ws + c->o_dhd, *dheads = ws + c->o_dheads,
        *dh = ws + c->o_dh, *drpart = ws + c->o_drpart, *duals = ws + c->o_duals;
  float* P = d.params;
  float* G = d.grads;
  if (!want_outputs) logits = concat_z = bce = kl = nullptr;
  const AdamArgs base = {d.params, d.adam_m, d.adam_v, d.step_count, d.lr};
  auto at = [&](int64_t off) {
    AdamArgs a = base;
    a.p += off;
    a.m += off;
    a.v += off;
    return a;
  };
  // register-resident fast paths of the latent kernels (the BASELINE MLP configs with few components qualify)
  const bool fast = NH <= 16 && Z <= 8 && H <= 512 && (H & 3) == 0 && aligned16(P + d.off_w_heads);
  int zp = 1;
  while (zp < Z) zp <<= 1;
  const bool fast_b = fast && (H + 256 / zp - 1) / (256 / zp) <= 16 && (B % 16 == 0) && (H % 16 == 0) &&
                      (D % 16 == 0);

  // FULL: tile-aligned shapes and 16-byte aligned operands (true for every BASELINE MLP config at B = 128)
  const bool full = (B % 16 == 0) && (H % 16 == 0) && (D % 16 == 0) && aligned16(x) && aligned16(P + d.off_w_e0) &&
                    aligned16(P + d.off_w_logits) && aligned16(ws);
  if (full)
    STEP_LAUNCH(k_enc_fwd<true>, dim3(8 * ((c->nt_h + 7) / 8) * c->nt_b), dim3(512), 0, x, P + d.off_w_e0,
                P + d.off_b_e0, h, B, H, D, d.step_count, fused ? 1 : 0, (double)d.lr);
  else
    STEP_LAUNCH(k_enc_fwd<false>, dim3(8 * ((c->nt_h + 7) / 8) * c->nt_b), dim3(512), 0, x, P + d.off_w_e0,
                P + d.off_b_e0, h, B, H, D, d.step_count, fused ? 1 : 0, (double)d.lr);
  {
    const size_t lds = (((size_t)H + 3) & ~(size_t)3) * sizeof(float) + ((size_t)d.eps_dim + 4) * sizeof(float);
#define LF(DM, FA)                                                                                                   \
  STEP_LAUNCH((k_latent_fwd<DM, FA>), dim3(B), dim3(512), lds, c->t, h, P + d.off_w_heads,                 \
                     P + d.off_b_heads, eps, d.eps_dim, P + d.off_radii, P + d.off_w_d0, P + d.off_b_d0, heads,      \
                     c->ldh, z, c->ldz, concat_z, klw, kl, hd, B, H, NH, Z, duals)
    if (fast) { DMAX_SWITCH(c->dmax, LF(DM, true)); } else { DMAX_SWITCH(c->dmax, LF(DM, false)); }
#undef LF
  }
  if (full)
    STEP_LAUNCH(k_dec1_fwd<true>, dim3(8 * ((c->nt_d + 7) / 8) * c->nt_b), dim3(512), 0, hd, P + d.off_w_logits,
                P + d.off_b_logits, x, g, bce_part, logits, B, H, D);
  else
    STEP_LAUNCH(k_dec1_fwd<false>, dim3(8 * ((c->nt_d + 7) / 8) * c->nt_b), dim3(512), 0, hd, P + d.off_w_logits,
                P + d.off_b_logits, x, g, bce_part, logits, B, H, D);
  {
    const int n_dhd = c->nt_b * c->nt_h, n_db = (D + kColsPerBlock - 1) / kColsPerBlock;
#define DB(AD, FU)                                                                                             \
  STEP_LAUNCH((k_dec1_bwd<AD, FU>), dim3(n_dhd + n_db + 1), dim3(512), 0, g, hd, P + d.off_w_logits,               \
              G + d.off_b_logits, dhd, bce_part, klw, bce, d.stats, beta, B, H, D, d.ncomp, n_dhd, n_db,          \
              at(d.off_b_logits))
    if (fused) { if (full) DB(true, true); else DB(true, false); }
    else { if (full) DB(false, true); else DB(false, false); }
#undef DB
  }
  {
    const int n_dwl = c->nt_d * ((c->nt_h + kTileWaves5 - 1) / kTileWaves5);
    const size_t lds = ((((size_t)H + 3) & ~(size_t)3) + 1024 + 8) * sizeof(float);  // dhd row | dz partials
#define LB(DM, FA, AD)                                                                                              \
  STEP_LAUNCH((k_latent_bwd<DM, FA, AD>), dim3(B + n_dwl), dim3(64 * kTileWaves5), lds, c->t, dhd, P + d.off_w_d0, \
                     c->ldh, h, P + d.off_w_heads, dheads, dh, drpart, g,                                           \
                     hd, G + d.off_w_logits, beta, B, H, D, NH, Z, B, at(d.off_w_logits), duals)
    if (fast_b) {
      if (fused) { DMAX_SWITCH(c->dmax, LB(DM, true, true)); } else { DMAX_SWITCH(c->dmax, LB(DM, true, false)); }
    } else {
      if (fused) { DMAX_SWITCH(c->dmax, LB(DM, false, true)); } else { DMAX_SWITCH(c->dmax, LB(DM, false, false)); }
    }
#undef LB
  }
  {
    const int tw = kTileWaves;
    const int n_we0 = c->nt_h * ((c->nt_d + tw - 1) / tw), n_wh = ((NH + 15) / 16) * ((c->nt_h + tw - 1) / tw),
              n_wd0 = c->nt_h * (((Z + 15) / 16 + tw - 1) / tw);
    const int n_be0 = (H + kColsPerBlock - 1) / kColsPerBlock, n_bh = (NH + kColsPerBlock - 1) / kColsPerBlock,
              n_bd0 = n_be0;
    const int grid = n_we0 + n_wh + n_wd0 + n_be0 + n_bh + n_bd0 + 1;
#define EB(AD, FU)                                                                                                   \
  STEP_LAUNCH((k_enc_bwd<AD, FU>), dim3(grid), dim3(64 * kTileWaves), 0, c->t, dh, x, dheads, c->ldh, h, dhd, z, c->ldz, \
                     drpart, G, P, B, H, D, NH, Z, n_we0, n_wh, n_wd0, n_be0, n_bh, n_bd0, d.off_w_e0, d.off_b_e0,   \
                     d.off_w_heads, d.off_b_heads, d.off_w_d0, d.off_b_d0, base, (double)d.curvature_lr, do_curv)
    if (fused) { if (full) EB(true, true); else EB(true, false); }
    else { if (full) EB(false, true); else EB(false, false); }
#undef EB
  }
#undef STEP_LAUNCH
  LAUNCH_CHECK("step launch");
  return 0;
}

extern "C" int mvae_step_forward_backward(mvae_ctx* c, const float* x, const float* eps, float beta, int want_outputs,
                                          float* logits, float* concat_z, float* bce, float* kl, void* stream) {
  return step_impl(c, x, eps, beta, false, 0, want_outputs, logits, concat_z, bce, kl, stream, nullptr);
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
  return step_impl(c, x, eps, beta, true, do_curvature_step, 0, nullptr, nullptr, nullptr, nullptr, stream, nullptr);
}

extern "C" int mvae_step_profile(mvae_ctx* c, const float* x, const float* eps, float beta, int do_curvature_step,
                                 int iters, float* ms_out, void* stream) {
  if (!c || !x || !eps || !ms_out || iters < 1) return fail(MVAE_E_BADARG, "null pointer / iters < 1%s", "");
  constexpr int NK = MVAE_STEP_KERNELS;
  hipEvent_t ev[2 * NK];
  for (auto& e : ev) {
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return hip_fail(rc, "hipEventCreate");
  }
  double acc[NK] = {0};
  int rc = 0;
  for (int it = 0; it < iters && rc == 0; ++it) {
    // every launch carries its own start/stop event (hipExtLaunchKernelGGL): the difference is the execution time
    // of that dispatch alone, the quantity rocprofv3 --kernel-trace reports
    rc = step_impl(c, x, eps, beta, true, do_curvature_step, 0, nullptr, nullptr, nullptr, nullptr, stream, ev);
    if (rc) break;
    hipError_t e = hipEventSynchronize(ev[2 * NK - 1]);
    if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
    for (int k = 0; k < NK; ++k) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]);
      acc[k] += ms;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc) return rc;
  for (int k = 0; k < NK; ++k) ms_out[k] = (float)(acc[k] / iters);
  return 0;
}

// Adam over a flat parameter buffer whose first 64 floats are the raw radii (SGD on the trainable ones): the optimizer
// of any architecture laid out like StepEngine's buffers (used by the conv path).
extern "C" int mvae_optimizer_step_flat(float* params, float* grads, float* adam_m, float* adam_v,
                                        int64_t n_params, int32_t* counters, int ncomp,
                                        const uint8_t* radius_trainable, double lr, double curvature_lr,
                                        int do_curvature_step, void* stream) {
  if (!params || !grads || !adam_m || !adam_v || !counters || n_params < kRadiiRegion || (n_params & 3) ||
      ncomp < 0 || ncomp > kMaxComp)
    return fail(MVAE_E_BADARG, "null pointer / bad size%s", "");
  CompTable t;
  memset(&t, 0, sizeof(t));
  t.n = ncomp;
  for (int i = 0; i < ncomp; ++i) t.trainable[i] = radius_trainable ? radius_trainable[i] : 0;
  const int n4 = (int)(n_params / 4);
  const int blocks = (n4 - kRadiiRegion / 4 + 255) / 256;
  hipLaunchKernelGGL(k_optim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, params, grads, adam_m, adam_v, n4,
                     counters, lr, curvature_lr, do_curvature_step);
  LAUNCH_CHECK("flat optimizer launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ conv building blocks (API)
// The reference's conv architecture (conv_vae.py:28-79) uses only Conv2d / ConvTranspose2d with kernel 4, stride 2,
// padding 1.  Both are expressed on the dense MFMA contractions above through a patch matrix:
//   Conv2d forward          y[(b,oy,ox), oc]       = im2col(x)[(b,oy,ox), (ic,ky,kx)] . W[oc, (ic,ky,kx)]^T      (NT)
//   ConvTranspose2d forward col[(b,iy,ix),(oc,ky,kx)] = x[(b,iy,ix), ic] . W[ic, (oc,ky,kx)]  then y = col2im(col)   (NN)
// and their backward passes are the same two gathers with the roles of input and output exchanged.
// Activations are addressed through explicit (batch, channel, y, x) strides, so NCHW tensors at the model boundary
// and channel-last tensors between layers use the same kernels.

// col[(b,oy,ox), (c,ky,kx)] = src[b, c, 2oy-1+ky, 2ox-1+kx]  (0 outside), optionally masked by mask[...same index] > 0
__global__ __launch_bounds__(256) void k_im2col(const float* src, const float* mask, float* col, int B, int C, int IH,
                                                int IW, int64_t sb, int64_t sc, int64_t sy, int64_t sx) {
  const int OH = IH / 2, OW = IW / 2, K = C * 16;
  const int64_t total = (int64_t)B * OH * OW * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const int64_t m = i / K;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((int64_t)OW * OH));
    const int c = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
    const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
    float v = 0.f;
    if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) {
      const int64_t o = b * sb + c * sc + iy * sy + ix * sx;
      v = src[o];
      if (mask && !(mask[o] > 0.f)) v = 0.f;
    }
    col[i] = v;
  }
}

// dst[b, c, y, x] = act(bias[c] + sum over the (ky,kx) with y = 2*py-1+ky, x = 2*px-1+kx of col[(b,py,px), (c,ky,kx)])
// (PH = H/2 patch rows).  With mask != NULL the result is multiplied by [mask[b,c,y,x] > 0] (backward through a ReLU).
__global__ __launch_bounds__(256) void k_col2im(const float* col, const float* bias, const float* mask, float* dst,
                                                int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sy,
                                                int64_t sx, int relu) {
  const int PH = H / 2, PW = W / 2, K = C * 16;
  const int64_t total = (int64_t)B * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // enumerate with the channel fastest so that consecutive threads read consecutive (c,ky,kx) groups
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((int64_t)W * H));
    float acc = bias ? bias[c] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ky = ((y + 1) & 1) + 2 * a;  // ky with the parity of y+1
      const int py = (y + 1 - ky) / 2;
      if (py < 0 || py >= PH) continue;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kx = ((x + 1) & 1) + 2 * e;
        const int px = (x + 1 - kx) / 2;
        if (px < 0 || px >= PW) continue;
        acc += col[(((int64_t)b * PH + py) * PW + px) * K + c * 16 + ky * 4 + kx];
      }
    }
    const int64_t o = b * sb + c * sc + y * sy + x * sx;
    if (relu) acc = acc < 0.f ? 0.f : acc;
    if (mask && !(mask[o] > 0.f)) acc = 0.f;
    dst[o] = acc;
  }
}

// ---- the same two gathers with the patch axis ordered (ky, kx, c) -- "taps-major" -- for channel-last tensors.
// With c fastest, a patch row is 16 contiguous runs of C floats of the source, and the four terms of an output pixel are
// contiguous runs of the patch matrix: both directions move 16-byte vectors, fully coalesced (the (c,ky,kx) order of the
// reference's weight layout makes consecutive channels 64 bytes apart in the patch matrix).  The weight matrices are
// permuted to the same order by the host layer (mvae_permute_rc on [OC, C, 16]).  Requires C % 4 == 0, sc == 1.
__global__ __launch_bounds__(256) void k_im2col_tm(const float* src, const float* mask, float* col, int B, int C,
                                                   int IH, int IW, int64_t sb, int64_t sy, int64_t sx) {
  const int OH = IH / 2, OW = IW / 2, K4 = C * 4;  // K / 4
  const int64_t total4 = (int64_t)B * OH * OW * K4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K4) * 4;
    const int64_t m = i / K4;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((int64_t)OW * OH));
    const int tap = k / C, c = k - tap * C, ky = tap >> 2, kx = tap & 3;
    const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) {
      const int64_t o = b * sb + iy * sy + ix * sx + c;
      v = *reinterpret_cast<const float4*>(src + o);
      if (mask) {
        const float4 mk = *reinterpret_cast<const float4*>(mask + o);
        if (!(mk.x > 0.f)) v.x = 0.f;
        if (!(mk.y > 0.f)) v.y = 0.f;
        if (!(mk.z > 0.f)) v.z = 0.f;
        if (!(mk.w > 0.f)) v.w = 0.f;
      }
    }
    *reinterpret_cast<float4*>(col + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void k_col2im_tm(const float* col, const float* bias, const float* mask, float* dst,
                                                   int B, int C, int H, int W, int64_t sb, int64_t sy, int64_t sx,
                                                   int relu) {
  const int PH = H / 2, PW = W / 2, K = C * 16, C4 = C / 4;
  const int64_t total4 = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const int64_t r = i / C4;
    const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((int64_t)W * H));
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ky = ((y + 1) & 1) + 2 * a;  // ky with the parity of y+1
      const int py = (y + 1 - ky) / 2;
      if (py < 0 || py >= PH) continue;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kx = ((x + 1) & 1) + 2 * e;
        const int px = (x + 1 - kx) / 2;
        if (px < 0 || px >= PW) continue;
        const float4 t =
            *reinterpret_cast<const float4*>(col + (((int64_t)b * PH + py) * PW + px) * K + (ky * 4 + kx) * C + c);
        acc.x += t.x;
        acc.y += t.y;
        acc.z += t.z;
        acc.w += t.w;
      }
    }
    const int64_t o = b * sb + y * sy + x * sx + c;
    if (relu) {
      acc.x = acc.x < 0.f ? 0.f : acc.x;
      acc.y = acc.y < 0.f ? 0.f : acc.y;
      acc.z = acc.z < 0.f ? 0.f : acc.z;
      acc.w = acc.w < 0.f ? 0.f : acc.w;
    }
    if (mask) {
      const float4 mk = *reinterpret_cast<const float4*>(mask + o);
      if (!(mk.x > 0.f)) acc.x = 0.f;
      if (!(mk.y > 0.f)) acc.y = 0.f;
      if (!(mk.z > 0.f)) acc.z = 0.f;
      if (!(mk.w > 0.f)) acc.w = 0.f;
    }
    *reinterpret_cast<float4*>(dst + o) = acc;
  }
}

// out[b][c][r] = in[b][r][c]   (channel-last <-> channel-first flattening of a small activation)
__global__ __launch_bounds__(256) void k_permute_rc(const float* in, float* out, int64_t B, int R, int Cc) {
  const int64_t total = B * R * Cc;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i % R);
    const int c = (int)((i / R) % Cc);
    const int64_t b = i / ((int64_t)R * Cc);
    out[i] = in[(b * R + r) * Cc + c];
  }
}

__global__ __launch_bounds__(256) void k_gemm_tn(const float* P, const float* Q, float* out, int M, int NP, int NQ) {
  const int ntQ4 = ((NQ + 15) / 16 + 3) / 4;
  AdamArgs none = {nullptr, nullptr, nullptr, nullptr, 0.0};
  job_tn_wave<false>(P, NP, NP, blockIdx.x / ntQ4, Q, NQ, NQ, (blockIdx.x % ntQ4) * 4 + (threadIdx.x >> 6), M, out, NQ,
                     none);
}

__global__ __launch_bounds__(256) void k_gemm_nn(const float* G, const float* W, const float* mask, float* out, int M,
                                                 int K, int N) {
  __shared__ float red[4][16][17];
  const int ntN = (N + 15) / 16;
  job_nn(red, G, K, M, blockIdx.x / ntN, W, N, N, blockIdx.x % ntN, K, mask, N, out, N);
}

__global__ __launch_bounds__(256) void k_colsum(const float* G, float* out, int M, int N) {
  __shared__ float lds[32 * 17 + 2];
  AdamArgs none = {nullptr, nullptr, nullptr, nullptr, 0.0};
  job_colsum_opt<false>(lds, G, N, M, N, blockIdx.x * kColsPerBlock, out, none);
}

// g[r][j] = sigmoid(logits[r][j]) - x[r][j];  bce[r] = sum_j BCE-with-logits   (one workgroup per row; 16-byte moves
// when D % 4 == 0 and the buffers are aligned; the four wave sums are added in wave order)
__global__ __launch_bounds__(256) void k_bce_fwd_bwd(const float* logits, const float* x, float* bce, float* g,
                                                     int64_t rows, int D) {
  __shared__ float sm[4];
  const int tid = threadIdx.x;
  const int64_t r = blockIdx.x;
  const float* yl = logits + r * D;
  const float* tl = x + r * D;
  float* gl = g + r * D;
  auto term = [](float y, float t, float* gout) -> float {
    const float e = mvf::fexp(-fabsf(y));
    *gout = ((y >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e)) - t;
    return (1.f - t) * y - (fminf(y, 0.f) - mvf::log1p_pos(e));
  };
  float s = 0.f;
  if ((D & 3) == 0 && ((((uintptr_t)logits | (uintptr_t)x | (uintptr_t)g) & 15) == 0)) {
    for (int j = tid * 4; j < D; j += 1024) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(yl + j), t = *reinterpret_cast<const f32x4*>(tl + j);
      f32x4 gv;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float go;
        s += term(y[u], t[u], &go);
        gv[u] = go;
      }
      *reinterpret_cast<f32x4*>(gl + j) = gv;
    }
  } else {
    for (int j = tid; j < D; j += 256) {
      float go;
      s += term(yl[j], tl[j], &go);
      gl[j] = go;
    }
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) sm[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) bce[r] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// BatchStats (stats.py:144-212) for paths that do not run the fused MLP step: one workgroup
__global__ __launch_bounds__(256) void k_batch_stats(const float* bce, const float* kl, float* stats, float beta, int B,
                                                     int ncomp) {
  __shared__ float sm[8];
  const int tid = threadIdx.x;
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
  };
  float b = 0.f, e = 0.f;
  for (int r = tid; r < B; r += 256) {
    float klr = kl[r];
    int i = 1;
    for (; i + 7 < ncomp; i += 8) {  // 8 loads in flight, added in index order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = kl[(size_t)(i + u) * B + r];
#pragma unroll
      for (int u = 0; u < 8; ++u) klr += v[u];
    }
    for (; i < ncomp; ++i) klr += kl[(size_t)i * B + r];
    b += bce[r];
    e += (-bce[r] - beta * klr);
  }
  const float bs = block_sum(b), es = block_sum(e);
  const int last = 4 + ncomp;
  float kt = 0.f;
  for (int i = 0; i < ncomp; ++i) {
    float a = 0.f;
    for (int r = tid; r < B; r += 256) a += kl[(size_t)i * B + r];
    const float sres = block_sum(a);
    kt += sres;
    if (tid == 0) {
      stats[4 + i] += sres;
      stats[last + 4 + i] = sres;
    }
  }
  if (tid == 0) {
    stats[0] += bs; stats[1] += kt; stats[2] += es; stats[3] += 1.f;
    stats[last] = bs; stats[last + 1] = kt; stats[last + 2] = es; stats[last + 3] = 1.f;
  }
}

static int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

static bool taps_major_ok(const void* a, const void* b, const void* c, int C, int64_t sc, int64_t sb, int64_t sy,
                          int64_t sx) {
  return sc == 1 && (C & 3) == 0 && ((sb | sy | sx) & 3) == 0 && aligned16(a) && aligned16(b) && (!c || aligned16(c));
}

extern "C" int mvae_im2col_k4s2p1(const float* src, const float* mask, float* col, int B, int C, int IH, int IW,
                                  int64_t sb, int64_t sc, int64_t sy, int64_t sx, int taps_major, void* stream) {
  if (!src || !col || B < 1 || C < 1 || IH < 2 || IW < 2 || (IH & 1) || (IW & 1))
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (taps_major) {
    if (!taps_major_ok(src, col, mask, C, sc, sb, sy, sx))
      return fail(MVAE_E_ALIGN, "taps-major im2col needs a channel-last source with C %% 4 == 0%s", "");
    hipLaunchKernelGGL(k_im2col_tm, dim3(grid_for((int64_t)B * (IH / 2) * (IW / 2) * C * 4)), dim3(256), 0,
                       (hipStream_t)stream, src, mask, col, B, C, IH, IW, sb, sy, sx);
    LAUNCH_CHECK("im2col launch");
    return 0;
  }
  hipLaunchKernelGGL(k_im2col, dim3(grid_for((int64_t)B * (IH / 2) * (IW / 2) * C * 16)), dim3(256), 0,
                     (hipStream_t)stream, src, mask, col, B, C, IH, IW, sb, sc, sy, sx);
  LAUNCH_CHECK("im2col launch");
  return 0;
}

extern "C" int mvae_col2im_k4s2p1(const float* col, const float* bias, const float* mask, float* dst, int B, int C,
                                  int H, int W, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int relu,
                                  int taps_major, void* stream) {
  if (!col || !dst || B < 1 || C < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (taps_major) {
    if (!taps_major_ok(col, dst, mask, C, sc, sb, sy, sx) || (bias && !aligned16(bias)))
      return fail(MVAE_E_ALIGN, "taps-major col2im needs a channel-last destination with C %% 4 == 0%s", "");
    hipLaunchKernelGGL(k_col2im_tm, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       col, bias, mask, dst, B, C, H, W, sb, sy, sx, relu);
    LAUNCH_CHECK("col2im launch");
    return 0;
  }
  hipLaunchKernelGGL(k_col2im, dim3(grid_for((int64_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, col, bias,
                     mask, dst, B, C, H, W, sb, sc, sy, sx, relu);
  LAUNCH_CHECK("col2im launch");
  return 0;
}

extern "C" int mvae_permute_rc(const float* in, float* out, int64_t B, int R, int Cc, void* stream) {
  if (!in || !out || B < 1 || R < 1 || Cc < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  hipLaunchKernelGGL(k_permute_rc, dim3(grid_for(B * R * Cc)), dim3(256), 0, (hipStream_t)stream, in, out, B, R, Cc);
  LAUNCH_CHECK("permute launch");
  return 0;
}

// ---- LDS-tiled f32 MFMA contraction for the large shapes of the conv architecture (M = B*OH*OW up to 65536 rows).
// C[M,N] = A . B with A(i,k) at A[i*sai + k*sak] and B(k,j) at B[k*sbk + j*sbj]; exactly one stride of each operand
// is 1 (the template flag says which), so one kernel serves
//   NT  y = x W^T        A = x [M,K] (k contiguous),  B = W [N,K] (k contiguous)     Conv2d forward / Linear
//   NN  y = g W          A = g [M,K] (k contiguous),  B = W [K,N] (j contiguous)     ConvTranspose2d forward, dX
//   TN  dW = P^T Q       A = P [Kc,M'] (i contiguous), B = Q [Kc,N] (j contiguous)   weight gradients (split-K slices)
// Workgroup tile BM x BN, K step BK (32: with 16 a 64 x 64 tile has only 512 MFMA cycles per wave between two
// barriers and the fixed barrier + LDS latency shows, MfmaUtil 56 %); 4 waves as 2 x 2, each wave (BM/2) x (BN/2) as
// 16 x 16 MFMA tiles (the 64 x 64 and 128 x 128 launches run 8 waves as 2 x 4: four waves per SIMD at two workgroups
// per CU, +3 % on the conv step).  Both operand tiles sit in LDS as [row][k] with a row stride of BK + 8 floats: a lane fetches
// ONE 16-byte vector per 16 x 16 x 16 sub-product (k is consumed in the permuted order {kk*4 + j}, the same for A and
// B); strides 24 / 40 are conflict-free for ds_read_b128 under its 16-lane service groups ({0-3,12-15,20-27}, ...
// over 64 banks).  Global -> register prefetch of the next K step overlaps the MFMAs of the current one.
template <int BM, int BN, int BK, int NW, bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * NW) void k_gemm_tiled(const float* __restrict__ A, int64_t sai, int64_t sak,
                                                    const float* __restrict__ Bm, int64_t sbk, int64_t sbj,
                                                    float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                                    const float* __restrict__ mask, int relu, int M, int N, int K,
                                                    int k_per_slice, int64_t slice_stride) {
  constexpr int kGT_BK = BK, kGT_LD = BK + 8, KQ = BK / 4;
  __shared__ __attribute__((aligned(16))) float As[BM * kGT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * kGT_LD];
  constexpr int NT = 64 * NW, WCOLS = NW / 2;  // waves as 2 x WCOLS
  constexpr int WM = BM / 2, WN = BN / WCOLS, TM = WM / 16, TN = WN / 16;
  constexpr int LA = BM * KQ / NT, LB = BN * KQ / NT;  // 16-byte vectors per thread per K step
  static_assert(LA >= 1 && LB >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kb = blockIdx.z * k_per_slice;
  const int ke = (kb + k_per_slice < K) ? kb + k_per_slice : K;
  C += (size_t)blockIdx.z * slice_stride;

  float4 ra[LA], rb[LB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < LA; ++r) {
      const int f = tid + NT * r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (A_KC) {  // 4 consecutive k of row i
        const int i = f / KQ, k = k0 + ((f % KQ) << 2);
        if (m0 + i < M && k < ke) v = *reinterpret_cast<const float4*>(A + (size_t)(m0 + i) * sai + k);
      } else {  // 4 consecutive i of column k
        const int k = k0 + (f % BK), i = (f / BK) << 2;
        if (m0 + i < M && k < ke) v = *reinterpret_cast<const float4*>(A + (size_t)k * sak + (m0 + i));
      }
      ra[r] = v;
    }
#pragma unroll
    for (int r = 0; r < LB; ++r) {
      const int f = tid + NT * r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B_KC) {
        const int j = f / KQ, k = k0 + ((f % KQ) << 2);
        if (n0 + j < N && k < ke) v = *reinterpret_cast<const float4*>(Bm + (size_t)(n0 + j) * sbj + k);
      } else {
        const int k = k0 + (f % BK), j = (f / BK) << 2;
        if (n0 + j < N && k < ke) v = *reinterpret_cast<const float4*>(Bm + (size_t)k * sbk + (n0 + j));
      }
      rb[r] = v;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int r = 0; r < LA; ++r) {
      const int f = tid + NT * r;
      if (A_KC) {
        *reinterpret_cast<float4*>(As + (f / KQ) * kGT_LD + ((f % KQ) << 2)) = ra[r];
      } else {  // rows i and i+4 share banks at this stride: the odd 16-lane halves store their rows rotated by 2
        const int k = f % BK, i = (f / BK) << 2;
        const bool rot = BK == 16 && ((f >> 4) & 1);
        As[(i + (rot ? 2 : 0)) * kGT_LD + k] = rot ? ra[r].z : ra[r].x;
        As[(i + (rot ? 3 : 1)) * kGT_LD + k] = rot ? ra[r].w : ra[r].y;
        As[(i + (rot ? 0 : 2)) * kGT_LD + k] = rot ? ra[r].x : ra[r].z;
        As[(i + (rot ? 1 : 3)) * kGT_LD + k] = rot ? ra[r].y : ra[r].w;
      }
    }
#pragma unroll
    for (int r = 0; r < LB; ++r) {
      const int f = tid + NT * r;
      if (B_KC) {
        *reinterpret_cast<float4*>(Bs + (f / KQ) * kGT_LD + ((f % KQ) << 2)) = rb[r];
      } else {
        const int k = f % BK, j = (f / BK) << 2;
        const bool rot = BK == 16 && ((f >> 4) & 1);
        Bs[(j + (rot ? 2 : 0)) * kGT_LD + k] = rot ? rb[r].z : rb[r].x;
        Bs[(j + (rot ? 3 : 1)) * kGT_LD + k] = rot ? rb[r].w : rb[r].y;
        Bs[(j + (rot ? 0 : 2)) * kGT_LD + k] = rot ? rb[r].x : rb[r].z;
        Bs[(j + (rot ? 1 : 3)) * kGT_LD + k] = rot ? rb[r].y : rb[r].w;
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = (wave / WCOLS) * WM, wn = (wave % WCOLS) * WN;
  const int li = lane & 15, lk = (lane >> 4) << 2;

  fetch(kb);
  for (int k0 = kb; k0 < ke; k0 += kGT_BK) {
    stage();
    __syncthreads();
    if (k0 + kGT_BK < ke) fetch(k0 + kGT_BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const f32x4*>(As + (wm + a * 16 + li) * kGT_LD + kk + lk);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bf[b] = *reinterpret_cast<const f32x4*>(Bs + (wn + b * 16 + li) * kGT_LD + kk + lk);
      // operands swapped (B fragment first): the lane's four accumulator values are four CONSECUTIVE columns of one
      // output row, so the epilogue moves 16 bytes per lane.  Small wave tiles take the k-component outermost so
      // that consecutive MFMAs write different accumulators.
      if constexpr (TM * TN <= 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = mfma16(bf[b][j], af[a][j], acc[a][b]);
      } else {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][b] = mfma16(bf[b][j], af[a][j], acc[a][b]);
      }
    }
    __syncthreads();
  }
  // epilogue: lane holds row lane&15, columns 4*(lane>>4) + r of every 16 x 16 tile
  const bool vec = (((uintptr_t)C | (uintptr_t)bias | (uintptr_t)mask) & 15) == 0 && (ldc & 3) == 0;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int m = m0 + wm + a * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + wn + b * 16 + lk;
      if (n >= N) continue;
      f32x4 v = acc[a][b];
      if (vec && n + 3 < N) {
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (relu)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];  // torch.relu: NaN propagates
        if (mask) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + (size_t)m * ldc + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (mk[r] > 0.f) ? v[r] : 0.f;
        }
        *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= N) continue;
          float w = v[r] + (bias ? bias[n + r] : 0.f);
          if (relu) w = w < 0.f ? 0.f : w;
          if (mask && !(mask[(size_t)m * ldc + n + r] > 0.f)) w = 0.f;
          C[(size_t)m * ldc + n + r] = w;
        }
      }
    }
  }
}

// operand requirements of the 16-byte paths of k_gemm_tiled
static inline bool tiled_ok(const void* p, int64_t ld) { return ((uintptr_t)p & 15) == 0 && (ld & 3) == 0; }

#ifndef MV_BK64
#define MV_BK64 32
#endif
#ifndef MV_BK128
#define MV_BK128 32
#endif
#ifndef MV_NW64
#define MV_NW64 8
#endif
#ifndef MV_NW128
#define MV_NW128 8
#endif
constexpr int kBK64 = MV_BK64, kBK128 = MV_BK128, kNW64 = MV_NW64, kNW128 = MV_NW128;
template <bool A_KC, bool B_KC>
static void launch_gemm_tiled(const float* A, int64_t sai, int64_t sak, const float* Bm, int64_t sbk, int64_t sbj,
                              float* C, int64_t ldc, const float* bias, const float* mask, int relu, int M, int N,
                              int K, int slices, int k_per_slice, int64_t slice_stride, hipStream_t s) {
  // 128 x 128 tiles need >= ~2 workgroups per CU to hide their own latencies; below that 64 x 64 tiles (4x the
  // workgroups, half the LDS reuse) win on every conv layer shape of the reference
  const int64_t wg128 = (int64_t)((N + 127) / 128) * ((M + 127) / 128) * slices;
  if (N > 64 && wg128 < 512) {
    dim3 grid((N + 63) / 64, (M + 63) / 64, slices);
    hipLaunchKernelGGL((k_gemm_tiled<64, 64, kBK64, kNW64, A_KC, B_KC>), grid, dim3(64 * kNW64), 0, s, A, sai, sak, Bm, sbk, sbj, C, ldc,
                       bias, mask, relu, M, N, K, k_per_slice, slice_stride);
  } else if (N > 64) {
    dim3 grid((N + 127) / 128, (M + 127) / 128, slices);
    hipLaunchKernelGGL((k_gemm_tiled<128, 128, kBK128, kNW128, A_KC, B_KC>), grid, dim3(64 * kNW128), 0, s, A, sai, sak, Bm, sbk, sbj, C, ldc,
                       bias, mask, relu, M, N, K, k_per_slice, slice_stride);
  } else {
    dim3 grid((N + 63) / 64, (M + 127) / 128, slices);
    hipLaunchKernelGGL((k_gemm_tiled<128, 64, kBK128, 4, A_KC, B_KC>), grid, dim3(256), 0, s, A, sai, sak, Bm, sbk, sbj, C, ldc,
                       bias, mask, relu, M, N, K, k_per_slice, slice_stride);
  }
}

static bool linear_forward_tiled(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K,
                                 int relu, hipStream_t s) {
  if (!tiled_ok(x, K) || !tiled_ok(W, K) || M > 0x7fffffff) return false;
  launch_gemm_tiled<true, true>(x, K, 1, W, 1, K, y, N, b, nullptr, relu, (int)M, N, K, 1, (K + 15) & ~15, 0, s);
  return true;
}

// Long batch contractions (conv layers: M = B*OH*OW up to 65536 rows): the rows are cut into slices of kTnSlice, one
// workgroup-row of tiles per slice writes its partial [NP, NQ] product, and a second launch adds the slices in index
// order (deterministic; no float atomics).
constexpr int kTnSlice = 256;
__global__ __launch_bounds__(256) void k_gemm_tn_sliced(const float* P, const float* Q, float* part, int M, int NP,
                                                        int NQ, int tiles) {
  const int slice = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int ntQ4 = ((NQ + 15) / 16 + 3) / 4;
  const int m0 = slice * kTnSlice;
  const int rows = (M - m0) < kTnSlice ? (M - m0) : kTnSlice;
  AdamArgs none = {nullptr, nullptr, nullptr, nullptr, 0.0};
  job_tn_wave<false>(P + (size_t)m0 * NP, NP, NP, tile / ntQ4, Q + (size_t)m0 * NQ, NQ, NQ,
                     (tile % ntQ4) * 4 + (threadIdx.x >> 6), rows, part + (size_t)slice * NP * NQ, NQ, none);
}
// out[i] = sum_k part[k][i], fixed order: a workgroup owns 64 outputs, its four waves take the slices k = w, w+4, ...
// (8 loads in flight per lane), the four partial sums meet in LDS and are added in wave order.
__global__ __launch_bounds__(256) void k_sum_slices(const float* part, float* out, int64_t n, int slices,
                                                    const float* bias = nullptr, int ncols = 1, int relu = 0) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < n; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + lane;
    float s = 0.f;
    if (i < n) {
      int k = w;
      for (; k + 28 < slices; k += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + 4 * u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < slices; k += 4) s += part[(size_t)k * n + i];
    }
    sm[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < n) {
      float v = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
      if (bias) v += bias[i % ncols];
      if (relu) v = v < 0.f ? 0.f : v;
      out[i] = v;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_relu_mask(float* dy, const float* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if (!(y[i] > 0.f)) dy[i] = 0.f;
}

extern "C" int64_t mvae_gemm_tn_workspace_floats(int64_t M, int NP, int NQ) {
  if (M <= kTnSlice) return 0;
  return ((M + kTnSlice - 1) / kTnSlice) * (int64_t)NP * NQ;
}

extern "C" int mvae_gemm_tn(const float* P, const float* Q, float* out, int64_t M, int NP, int NQ, float* workspace,
                            void* stream) {
  if (!P || !Q || !out || M < 1 || NP < 1 || NQ < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (M >= kTiledMinRows && tiled_ok(P, NP) && tiled_ok(Q, NQ) && tiled_ok(out, NQ) && M <= 0x7fffffff) {
    // split-K over the batch rows so that >= 256 workgroups exist; the slices are added in index order
    const int wg = ((NP + 127) / 128) * ((NQ + (NQ > 64 ? 127 : 63)) / (NQ > 64 ? 128 : 64));
    int slices = (256 + wg - 1) / wg;
    const int max_slices = (int)((M + kTnSlice - 1) / kTnSlice);  // what mvae_gemm_tn_workspace_floats provides
    if (slices > max_slices) slices = max_slices;
    if (slices > 1 && !workspace) return fail(MVAE_E_BADARG, "mvae_gemm_tn needs a workspace for M > 256%s", "");
    const int kps = (int)((((M + slices - 1) / slices) + 15) & ~(int64_t)15);
    slices = (int)((M + kps - 1) / kps);
    const int64_t n = (int64_t)NP * NQ;
    launch_gemm_tiled<false, false>(P, 1, NP, Q, NQ, 1, slices > 1 ? workspace : out, NQ, nullptr, nullptr, 0, NP, NQ,
                                    (int)M, slices, kps, n, (hipStream_t)stream);
    if (slices > 1)
      hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * n)), dim3(256), 0, (hipStream_t)stream, workspace, out, n, slices);
    LAUNCH_CHECK("tiled gemm_tn launch");
    return 0;
  }
  const int tiles = ((NP + 15) / 16) * (((NQ + 15) / 16 + 3) / 4);
  if (M <= kTnSlice) {
    hipLaunchKernelGGL(k_gemm_tn, dim3(tiles), dim3(256), 0, (hipStream_t)stream, P, Q, out, (int)M, NP, NQ);
  } else {
    if (!workspace) return fail(MVAE_E_BADARG, "mvae_gemm_tn needs a workspace for M > 256%s", "");
    const int slices = (int)((M + kTnSlice - 1) / kTnSlice);
    hipLaunchKernelGGL(k_gemm_tn_sliced, dim3((unsigned)(tiles * slices)), dim3(256), 0, (hipStream_t)stream, P, Q,
                       workspace, (int)M, NP, NQ, tiles);
    const int64_t n = (int64_t)NP * NQ;
    hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * n)), dim3(256), 0, (hipStream_t)stream, workspace, out, n, slices);
  }
  LAUNCH_CHECK("gemm_tn launch");
  return 0;
}

extern "C" int mvae_relu_mask(float* dy, const float* y, int64_t n, void* stream) {
  if (!dy || !y || n < 0) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_relu_mask, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, n);
  LAUNCH_CHECK("relu mask launch");
  return 0;
}

extern "C" int mvae_gemm_nn(const float* G, const float* W, const float* mask, float* out, int64_t M, int K, int N,
                            void* stream) {
  if (!G || !W || !out || M < 1 || K < 1 || N < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (M >= kTiledMinRows && tiled_ok(G, K) && tiled_ok(W, N) && M <= 0x7fffffff) {
    launch_gemm_tiled<true, false>(G, K, 1, W, N, 1, out, N, nullptr, mask, 0, (int)M, N, K, 1, (K + 15) & ~15, 0,
                                   (hipStream_t)stream);
    LAUNCH_CHECK("tiled gemm_nn launch");
    return 0;
  }
  const int64_t grid = ((M + 15) / 16) * ((N + 15) / 16);
  if (grid > 0x7fffffff) return fail(MVAE_E_UNSUPPORTED, "grid too large%s", "");
  hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, G, W, mask, out, (int)M, K, N);
  LAUNCH_CHECK("gemm_nn launch");
  return 0;
}

// y = act(x W^T + b) for FEW rows and a LONG contraction (the conv architecture's heads: M = B, N = 12, K = 8192: 16
// output tiles).  K is cut into slices of kSplitK so that >= ~256 workgroups exist; the slices' partial products are
// added in index order, with the bias and the activation, by k_sum_slices.
constexpr int kSplitK = 128;
extern "C" int64_t mvae_linear_forward_splitk_workspace_floats(int64_t M, int N, int K) {
  const int64_t slices = (K + kSplitK - 1) / kSplitK;
  return slices > 1 ? slices * M * N : 0;
}
extern "C" int mvae_linear_forward_splitk(const float* x, const float* W, const float* b, float* y, int64_t M, int N,
                                          int K, int relu, float* workspace, void* stream) {
  if (!x || !W || !y || M < 1 || N < 1 || K < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  const int slices = (K + kSplitK - 1) / kSplitK;
  if (slices == 1 || !tiled_ok(x, K) || !tiled_ok(W, K) || M > 0x7fffffff)
    return mvae_linear_forward(x, W, b, y, M, N, K, relu, stream);
  if (!workspace) return fail(MVAE_E_BADARG, "mvae_linear_forward_splitk needs its workspace%s", "");
  const int64_t n = M * N;
  launch_gemm_tiled<true, true>(x, K, 1, W, 1, K, workspace, N, nullptr, nullptr, 0, (int)M, N, K, slices, kSplitK, n,
                                (hipStream_t)stream);
  hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * n)), dim3(256), 0, (hipStream_t)stream, workspace, y, n, slices, b,
                     N, relu);
  LAUNCH_CHECK("split-K linear forward launch");
  return 0;
}

// tall matrices (conv activations: up to 65536 rows): row slices of kColSlice are summed by separate workgroups, the
// slice totals are then added in index order
constexpr int kColSlice = 512;
__global__ __launch_bounds__(256) void k_colsum_sliced(const float* G, float* part, int M, int N, int ncb) {
  __shared__ float lds[32 * 17 + 2];
  const int slice = blockIdx.x / ncb, cb = blockIdx.x % ncb;
  const int m0 = slice * kColSlice;
  const int rows = (M - m0) < kColSlice ? (M - m0) : kColSlice;
  AdamArgs none = {nullptr, nullptr, nullptr, nullptr, 0.0};
  job_colsum_opt<false>(lds, G + (size_t)m0 * N, N, rows, N, cb * kColsPerBlock, part + (size_t)slice * N, none);
}

extern "C" int64_t mvae_colsum_workspace_floats(int64_t M, int N) {
  return M <= kColSlice ? 0 : ((M + kColSlice - 1) / kColSlice) * (int64_t)N;
}

extern "C" int mvae_colsum(const float* G, float* out, int64_t M, int N, float* workspace, void* stream) {
  if (!G || !out || M < 1 || N < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  const int ncb = (N + kColsPerBlock - 1) / kColsPerBlock;
  if (M <= kColSlice) {
    hipLaunchKernelGGL(k_colsum, dim3(ncb), dim3(256), 0, (hipStream_t)stream, G, out, (int)M, N);
  } else {
    if (!workspace) return fail(MVAE_E_BADARG, "mvae_colsum needs a workspace for M > 512%s", "");
    const int slices = (int)((M + kColSlice - 1) / kColSlice);
    hipLaunchKernelGGL(k_colsum_sliced, dim3((unsigned)(ncb * slices)), dim3(256), 0, (hipStream_t)stream, G,
                       workspace, (int)M, N, ncb);
    hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * (int64_t)N)), dim3(256), 0, (hipStream_t)stream, workspace, out, (int64_t)N,
                       slices);
  }
  LAUNCH_CHECK("colsum launch");
  return 0;
}

extern "C" int mvae_bce_forward_backward(const float* logits, const float* x, float* bce, float* g, int64_t rows,
                                         int D, void* stream) {
  if (!logits || !x || !bce || !g || rows < 1 || D < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  hipLaunchKernelGGL(k_bce_fwd_bwd, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, x,
                     bce, g, rows, D);
  LAUNCH_CHECK("bce launch");
  return 0;
}

extern "C" int mvae_batch_stats(const float* bce, const float* kl, float* stats, float beta, int B, int ncomp,
                                void* stream) {
  if (!bce || !kl || !stats || B < 1 || ncomp < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  hipLaunchKernelGGL(k_batch_stats, dim3(1), dim3(256), 0, (hipStream_t)stream, bce, kl, stats, beta, B, ncomp);
  LAUNCH_CHECK("batch stats launch");
  return 0;
}


// ------------------------------------------------------------------------------------------------ device-side input pipeline
// Row f-2 of the scope table: the reference feeds the step from 8 DataLoader worker processes that binarise every image
// on the CPU (mt/data/image_reconstruction.py:44-53,70-74) plus a host->device copy, and draws eps with the torch RNG
// inside the step.  Here the data set lives in HBM as uint8, and ONE small launch per step gathers the next batch by
// a device-resident permutation, binarises it dynamically (x = pixel/255 > U(0,1)) and draws eps ~ N(0,1), both from
// a counter-based Philox4x32-10 stream keyed by (seed, batch cursor) -- no host work, so a whole epoch can be
// replayed as HIP graphs.  The cursor lives in counters[8] and is advanced by launch 1 of the step.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void k_prepare_batch(const unsigned char* images, const int* perm, int n_images,
                                                       int D, int B, int E, unsigned long long seed,
                                                       const int* counters, int batches_per_epoch, int train,
                                                       float* x, float* eps) {
  const unsigned cursor = (unsigned)counters[8];
  const int bi = (int)(cursor % (unsigned)batches_per_epoch);
  const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  const int nx4 = (B * D + 3) / 4, ne4 = (B * E + 3) / 4;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nx4 + ne4; i += gridDim.x * 256) {
    unsigned r[4];
    if (i < nx4) {
      philox4x32_10((unsigned)i, cursor, 0u, 0u, k0, k1, r);  // stream 0: binarisation
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int e = i * 4 + t;
        if (e < B * D) {
          const int b = e / D, j = e - b * D;
          const int src = perm ? perm[(size_t)bi * B + b] : (bi * B + b);
          const float pix = (float)images[(size_t)(src < n_images ? src : n_images - 1) * D + j] / 255.0f;
          const float u = (float)(r[t] >> 8) * (1.0f / 16777216.0f);  // [0,1)
          x[e] = (train ? (pix > u) : (pix > 0.5f)) ? 1.0f : 0.0f;
        }
      }
    } else {
      const int q = i - nx4;
      philox4x32_10((unsigned)q, cursor, 1u, 0u, k0, k1, r);  // stream 1: eps
      // Box-Muller on two pairs
      float n[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float u1 = ((float)(r[2 * t] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
        const float u2 = (float)(r[2 * t + 1] >> 8) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        n[2 * t] = rad * cs;
        n[2 * t + 1] = rad * sn;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (q * 4 + t < B * E) eps[q * 4 + t] = n[t];
    }
  }
}

extern "C" int mvae_prepare_batch(const uint8_t* images, const int32_t* perm, int n_images, int D, int B, int E,
                                  uint64_t seed, const int32_t* counters, int batches_per_epoch, int train, float* x,
                                  float* eps, void* stream) {
  if (!images || !counters || !x || !eps || n_images < 1 || D < 1 || B < 1 || E < 1 || batches_per_epoch < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  const int work = (B * D + 3) / 4 + (B * E + 3) / 4;
  hipLaunchKernelGGL(k_prepare_batch, dim3((work + 255) / 256), dim3(256), 0, (hipStream_t)stream, images, perm,
                     n_images, D, B, E, (unsigned long long)seed, counters, batches_per_epoch, train, x, eps);
  LAUNCH_CHECK("prepare batch launch");
  return 0;
}
