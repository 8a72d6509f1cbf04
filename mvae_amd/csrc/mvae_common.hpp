// mvae_common.hpp -- what the translation units of libmvae_hip.so share: error reporting, the component table, the
// per-(row, component) device code over float / dual numbers, the wave-level tile jobs and small reductions.
// The library is built from three translation units compiled in parallel (mvae_amd/build.py):
//   mvae_api.hip   manifold primitives (+ their backward), scalar functions, component operators, generic dense layers,
//                  log-likelihood helpers
//   mvae_step.hip  the fused ELBO step (six launches) and the flat optimizer
//   mvae_conv.hip  building blocks of the conv architecture, the LDS-tiled contraction, the device-side input pipeline
#pragma once
#include <hip/hip_runtime.h>
#include "mvae_p3.hpp"
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mvae_hip.h"
#include "mvae_gemm.hpp"
#include "mvae_math.hpp"

using namespace mv;

// ------------------------------------------------------------------------------------------------ dev timing hooks
// -DMV_DBG_TIMING: workgroup 0 of the latent kernels stamps wall_clock64() (100 MHz) at phase boundaries.
#ifdef MV_DBG_TIMING
static __device__ unsigned long long g_dbg[64];
#define MV_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)
#define MV_STAMP_B(i, blk) do { if (blockIdx.x == (blk) && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)
#ifndef MV_STAMP_BLK
#define MV_STAMP_BLK 200  // which workgroup of launches 5 / 6 stamps its wave-tile phases
#endif
// start / end time and kind of every workgroup of launch L (plain stores to per-workgroup slots: no contention)
static __device__ unsigned long long g_span[6][3][2048];
// the start time stays in a register until the workgroup's end (a store at kernel entry would sit in front of the first
// s_waitcnt vmcnt(0) of the kernel and add its acknowledgement to the first phase)
#define MV_SPAN_BEGIN(L) const unsigned long long mv_span0_ = wall_clock64()
#define MV_SPAN_END(L, kind) do { if (threadIdx.x == 0 && blockIdx.x < 2048) { g_span[L][0][blockIdx.x] = mv_span0_; g_span[L][1][blockIdx.x] = wall_clock64(); g_span[L][2][blockIdx.x] = (kind); } } while (0)
// the same for another thread of the workgroup, recorded `off` slots further (e.g. the dual waves of launch 2)
#define MV_SPAN_END_T(L, kind, thr, off) do { if (threadIdx.x == (thr) && blockIdx.x + (off) < 2048) { g_span[L][0][blockIdx.x + (off)] = mv_span0_; g_span[L][1][blockIdx.x + (off)] = wall_clock64(); g_span[L][2][blockIdx.x + (off)] = (kind); } } while (0)
// register-resident stamps: MV_T(i) reads the clock into a local (no store between the phases, so a phase is not charged
// the acknowledgement of a stamp's own store); MV_TFLUSH(base, n, blk) writes them out at the end of the kernel
#define MV_TDECL unsigned long long mv_t_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MV_T(i) do { mv_t_[i] = wall_clock64(); } while (0)
#define MV_TFLUSH(base, n, blk) do { if (blockIdx.x == (blk) && threadIdx.x == 0) { for (int q_ = 0; q_ < (n); ++q_) g_dbg[(base) + q_] = mv_t_[q_]; } } while (0)
#else
#define MV_STAMP(i) do {} while (0)
#define MV_STAMP_B(i, blk) do {} while (0)
#define MV_TDECL do {} while (0)
#define MV_T(i) do {} while (0)
#define MV_TFLUSH(base, n, blk) do {} while (0)
#define MV_SPAN_BEGIN(L) do {} while (0)
#define MV_SPAN_END(L, kind) do {} while (0)
#define MV_SPAN_END_T(L, kind, thr, off) do {} while (0)
#endif


// ------------------------------------------------------------------------------------------------ errors
// one message buffer per calling thread, defined in mvae_api.hip
int fail(int code, const char* fmt, const char* a = "", long long b = 0);
int hip_fail(hipError_t e, const char* where);
#define LAUNCH_CHECK(where)                         \
  do {                                              \
    hipError_t e_ = hipGetLastError();              \
    if (e_ != hipSuccess) return hip_fail(e_, where); \
  } while (0)

// large, 16-byte aligned problems of mvae_linear_forward go to the LDS-tiled kernel (mvae_conv.hip)
constexpr int64_t kTiledMinRows = 512;
bool linear_forward_tiled(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K, int relu,
                          hipStream_t s);

// Running sums of the epoch statistics (stats.py:120-127 accumulates Python doubles): float32 with Kahan compensation,
// sum block at stats[i], compensation at stats[comp_off + i].  A plain float32 sum drifts by ~1e-3 .. 4e-2 nats per
// sample over a CIFAR-scale epoch (sum ~1e8, ulp 8); the compensated one stays within an ulp of the true total.
__device__ __forceinline__ void kahan_add(float* stats, int i, int comp_off, float old_sum, float old_c, float x) {
  const float y = x - old_c;
  const float t = old_sum + y;
  stats[comp_off + i] = (t - old_sum) - y;
  stats[i] = t;
}

// ------------------------------------------------------------------------------------------------ peer exchange
// State of the one-shot peer-read gradient exchange (mvae_peer.hip; read by the optimizer launch in mvae_step.hip).
constexpr int kPeerMaxWorld = MVAE_PEER_MAX_WORLD;
struct PeerSrc {  // kernel argument: where rank r's published gradients of the CURRENT sequence number live
  const float* slot[kPeerMaxWorld];  // base of rank r's [2][n] slot pair, mapped into this process
  const int* seq;                    // device word: sequence number of the last publish (parity selects the slot)
  long long n;                       // floats per slot
  int world;
  long long slice4;                  // two-shot: float4 per owned slice (rank r owns [r slice4, (r+1) slice4)); 0 = one-shot
};
struct mvae_peer {
  int world = 0, rank = 0;
  int64_t n = 0;
  float* slots = nullptr;                    // own [2][n] floats (hipMalloc: exportable as ONE hipIpc handle)
  float* peer_slots[kPeerMaxWorld] = {};     // rank r's slots in this process's address space (own pointer for r == rank)
  bool imported[kPeerMaxWorld] = {};
  int* seq = nullptr;                        // device: [0] sequence number, [1] spare
  unsigned int* flags_host = nullptr;        // shared host page: [r] = last sequence number published by rank r,
  unsigned int* flags_dev = nullptr;         //                   [32 + r] = wait time-outs seen by rank r
  int shm_fd = -1;
  char shm_name[128] = {};
  unsigned long long timeout_ticks = 0;      // wall_clock64 ticks (100 MHz) a rank waits for a peer before giving up
  int mode = 0;  // 0: one-shot sum in the optimizer launch; 1: two-shot (reduce-scatter + all-gather of GRADIENTS by direct
                 // reads); 2: sharded optimizer (reduce-scatter + Adam on the owned slice + all-gather of PARAMETERS)
};
// (mvae_peer.hip) sharded form, after the owner's optimizer launch: raise the second flag, wait for every peer's, copy
// every other rank's slice of the updated parameters out of its slot
int peer_gather_params(mvae_peer* p, float* params, hipStream_t s);

// float4 per owned slice of the two-shot exchange (the radii region, 16 float4, stays inside slice 0)
inline long long peer_slice4(const mvae_peer* p) {
  const long long n4 = p->n / 4;
  long long s4 = (n4 + p->world - 1) / p->world;
  return s4 < 16 ? 16 : s4;
}

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
  // the inverse for the first four slots of a wave: component at (wave, slot), or -1 (uniform -- scalar -- lookups in the
  // 16-row latent kernels, where lane = slot * 16 + row)
  signed char slot_ci[4][4];
  mvae_component_desc slot_desc[4][4];  // the descriptor itself (kind = -1: empty slot): ONE uniform load per slot
};

inline int bucket_of(int dmax) {
  if (dmax <= 2) return 2;
  if (dmax <= 4) return 4;
  if (dmax <= 8) return 8;
  if (dmax <= 16) return 16;
  if (dmax <= 32) return 32;
  return 64;
}

inline int fill_table(CompTable* t, const mvae_component_desc* comps, int ncomp, const unsigned char* trainable,
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
  memset(t->slot_ci, -1, sizeof(t->slot_ci));
  for (int w = 0; w < 4; ++w)
    for (int sl = 0; sl < 4; ++sl) t->slot_desc[w][sl].kind = -1;
  for (int i = 0; i < ncomp; ++i)
    if (t->lane_of[i] < 4) {
      t->slot_ci[t->wave_of[i]][t->lane_of[i]] = (signed char)i;
      t->slot_desc[t->wave_of[i]][t->lane_of[i]] = comps[i];
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

// ---- the same two in float64 (the latent chain between float32 dense layers; mvae_math.hpp "float64 number types"): inputs
// and outputs stay float32 tensors, every intermediate of the chain is a double
template <int DMAX>
__device__ __forceinline__ void comp_fwd_row64(const mvae_component_desc& c, const float* heads_row, const float* eps_row,
                                               const float* radii, float* z_row, float* kl, float* lq, float* lp,
                                               float* mu_row, float* std_row) {
  MV_BOUNDS(DMAX + 1);
  double m[kN], l[kN], z[kN], mu[kN], sg[kN];
  float e[kN];
  const int d = c.true_dim, lvd = c.logvar_dim;
  MV_FOR(i, 0, d) {
    m[i] = (double)heads_row[c.mean_col + i];
    e[i] = eps_row[c.eps_col + i];
  }
  MV_FOR(i, 0, lvd) l[i] = (double)heads_row[c.logvar_col + i];
  double rp = (c.kind == kEuclidean) ? 0.0 : (double)radii[c.radius_idx];
  double klv = 0.0, lqv = 0.0, lpv = 0.0;
  comp_eval<DMAX, double>(c.kind, m, l, lvd, e, d, rp, z, kl ? &klv : nullptr, lq ? &lqv : nullptr, lq ? &lpv : nullptr,
                          mu_row ? mu : nullptr, std_row ? sg : nullptr);
  const int A = ambient_dim(c.kind, d);
  MV_FOR(i, 0, A) z_row[c.z_col + i] = (float)z[i];
  if (kl) *kl = (float)klv;
  if (lq) {
    *lq = (float)lqv;
    *lp = (float)lpv;
  }
  if (mu_row) {
    MV_FOR(i, 0, A) mu_row[c.z_col + i] = (float)mu[i];
  }
  if (std_row) {
    MV_FOR(i, 0, lvd) std_row[c.eps_col + i] = (float)sg[i];
  }
}
template <int DMAX>
__device__ __forceinline__ float comp_bwd_dir64(const mvae_component_desc& c, const float* heads_row, const float* eps_row,
                                                const float* radii, const float* dz_row, float dkl, int dir) {
  MV_BOUNDS(DMAX + 1);
  DualD m[kN], l[kN], z[kN];
  float e[kN];
  const int d = c.true_dim, lvd = c.logvar_dim;
  MV_FOR(i, 0, d) {
    m[i] = DualD{(double)heads_row[c.mean_col + i], (dir == i) ? 1.0 : 0.0};
    e[i] = eps_row[c.eps_col + i];
  }
  MV_FOR(i, 0, lvd) l[i] = DualD{(double)heads_row[c.logvar_col + i], (dir == d + i) ? 1.0 : 0.0};
  DualD rp = DualD{(c.kind == kEuclidean) ? 0.0 : (double)radii[c.radius_idx], (dir == d + lvd) ? 1.0 : 0.0};
  DualD kl;
  comp_eval<DMAX, DualD>(c.kind, m, l, lvd, e, d, rp, z, &kl, nullptr, nullptr, nullptr, nullptr);
  const int A = ambient_dim(c.kind, d);
  double g = (double)dkl * kl.d;
  MV_FOR(i, 0, A) g += (double)dz_row[c.z_col + i] * z[i].d;
  return (float)g;
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

// sum over the 16 lanes of a DPP row (every lane of the row ends up with the row's total): four DPP adds
__device__ __forceinline__ float row16_sum(float v) {
  int x = __float_as_int(v);
#define MV_DPP_ADD(CTRL)                                                                                \
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true)));
  MV_DPP_ADD(0xB1)   // quad_perm [1,0,3,2]
  MV_DPP_ADD(0x4E)   // quad_perm [2,3,0,1]
  MV_DPP_ADD(0x141)  // row_half_mirror
  MV_DPP_ADD(0x140)  // row_mirror
#undef MV_DPP_ADD
  return __int_as_float(x);
}

// the value of the neighbouring lane (lane ^ 1): one DPP quad_perm [1,0,3,2], no LDS crossbar
__device__ __forceinline__ float lane_swap1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
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

// the 4-byte form (results a LATER-arriving workgroup of the same launch reads after its acquire: they must have left
// this XCD's L2, and a device-scope release fence would write back every dirty line of it instead)
__device__ __forceinline__ void store4_wt(float* base, size_t idx, float v) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rsrc, (int)(idx << 2), 0, /*aux: sc1*/ 16);
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

// ---- FRAGMENT-ORDER copies of the batch-contraction operands (g, hd, dh, x: the dW tiles of launches 5 and 6).
// A row-major operand T[B][N] makes a wave-level request of the dW tile (4 rows x 16 columns of 4 bytes) touch four 64-byte
// HALVES of 128-byte lines, 64 such requests per tile.  The fragment-order copy TF holds, for column tile nt and row block mb,
// one contiguous 1 KB block in the order the MFMA wants it:
//     TF[((nt * MB + mb) * 64 + q * 16 + i) * 4 + t] = T[16 mb + 4 q + t][16 nt + i]        (MB = B / 16)
// so that lane (i, q) reads ONE 16-byte vector per row block -- four consecutive batch rows of its column, i.e. the A / B
// values of four MFMA steps -- and a wave-level request is 1 KB of whole lines: 16 requests per tile instead of 64.  The
// copies are written by the producers INSTEAD of the row-major tensor where every consumer can read fragment order (hd, dhd),
// next to it where one cannot (h), or by spare workgroups (x: the padding workgroups of launch 1's grid).
__host__ __device__ inline size_t frag_off(int m, int n, int MB) {
  return ((size_t)((n >> 4) * MB + (m >> 4)) << 8) + (size_t)(((((m & 15) >> 2) << 4) + (n & 15)) << 2) + (size_t)(m & 3);
}

// dW tile per WAVE from fragment-order operands (+ optional Adam): out[p][q] = sum_m P[m][p] Q[m][q], the output mapping of
// job_tn_wave (lane l: out[p0 + (l & 15)][q0 + 4 (l >> 4) + 0..3], one 16-byte access per lane for g / p / m / v).
// Lanes whose output row is >= NP or whose columns reach past NQ load nothing of p / m / v and store nothing: a ragged last
// tile (dW_heads: NP = heads_dim; dW_d0: NQ = z_dim) costs no branch in the contraction.  ldo % 4 != 0 (z_dim 6, 2): the
// lane's outputs move as scalars.  PROW != NULL: P is read ROW-MAJOR (ld = ldp) in the fragment order's row sequence -- four
// 4-byte requests per row block instead of one 16-byte one -- for an operand that has no fragment-order copy.
// COLSUM: the wave also delivers cs_out[16 pt + i] = sum_m P[m][16 pt + i] (+ Adam on cs_aa) -- the bias gradient that goes
// with the weight gradient's P operand -- from the fragments it holds anyway.
template <bool ADAM, bool COLSUM = false>
__device__ __forceinline__ void job_tn_frag_any(const float* PF, const float* PROW, int ldp, int pt, int NP, const float* QF,
                                                int qt, int NQ, int MB, float* out, int ldo, const AdamArgs& aa,
                                                float* cs_out = nullptr, const AdamArgs cs_aa = AdamArgs{},
                                                const float* scal = nullptr /* {-lr/bc1, sqrt(bc2)} already in registers */) {
  if (qt * 16 >= NQ) return;
  MV_STAMP_B(16, MV_STAMP_BLK);
  const int lane = threadIdx.x & 63;
  const int pr = pt * 16 + (lane & 15), qc0 = qt * 16 + ((lane >> 4) << 2);
  const bool vec = (ldo & 3) == 0;  // uniform
  const bool ok = pr < NP && (vec ? qc0 + 3 < NQ : qc0 < NQ);
  const size_t idx = ok ? (size_t)pr * ldo + qc0 : 0;
  const f32x4* qa = reinterpret_cast<const f32x4*>(QF) + ((size_t)qt * MB << 6) + lane;
  const f32x4* pb = reinterpret_cast<const f32x4*>(PF) + ((size_t)pt * MB << 6) + lane;
  const float* prow = PROW + (size_t)(4 * (lane >> 4)) * ldp + (pr < NP ? pr : 0);
  f32x4 av[8], bv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {  // (clamped: MB >= 1; blocks past MB are masked below)
    const int cc = c < MB ? c : 0;
    av[c] = qa[(size_t)cc << 6];
    if (PROW) {
      const float* r0 = prow + (size_t)(16 * cc) * ldp;
      bv[c] = f32x4{r0[0], r0[(size_t)ldp], r0[2 * (size_t)ldp], r0[3 * (size_t)ldp]};
    } else {
      bv[c] = pb[(size_t)cc << 6];
    }
  }
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, m0 = p0, v0 = p0;
  float neg_step = 0.f, bc2s = 1.f;
  if (ADAM) {
    if (vec) {
      p0 = *reinterpret_cast<const f32x4*>(aa.p + idx);
      m0 = *reinterpret_cast<const f32x4*>(aa.m + idx);
      v0 = *reinterpret_cast<const f32x4*>(aa.v + idx);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t ir = (ok && qc0 + r < NQ) ? idx + r : 0;
        p0[r] = aa.p[ir];
        m0[r] = aa.m[ir];
        v0[r] = aa.v[ir];
      }
    }
    if (scal) {
      neg_step = scal[0];
      bc2s = scal[1];
    } else {
      neg_step = reinterpret_cast<const float*>(aa.counters)[2];  // {-lr/bc1, sqrt(bc2)} of this step (launch 1)
      bc2s = reinterpret_cast<const float*>(aa.counters)[3];
    }
  }
  float cp = 0.f, cm = 0.f, cv = 0.f;  // COLSUM: the bias entry of this lane's P column (lanes 0..15 finish it)
  if (COLSUM && ADAM) {
    const int col = pr < NP ? pr : 0;
    cp = cs_aa.p[col];
    cm = cs_aa.m[col];
    cv = cs_aa.v[col];
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc, cs4 = acc;
  for (int c0 = 0;;) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c0 + c < MB) {  // uniform
        if (COLSUM) cs4 += bv[c];
        acc = mfma16(av[c][0], bv[c][0], acc);
        acc2 = mfma16(av[c][1], bv[c][1], acc2);
        acc = mfma16(av[c][2], bv[c][2], acc);
        acc2 = mfma16(av[c][3], bv[c][3], acc2);
      }
    }
    c0 += 8;
    if (c0 >= MB) break;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int cc = c0 + c < MB ? c0 + c : 0;
      av[c] = qa[(size_t)cc << 6];
      if (PROW) {
        const float* r0 = prow + (size_t)(16 * cc) * ldp;
        bv[c] = f32x4{r0[0], r0[(size_t)ldp], r0[2 * (size_t)ldp], r0[3 * (size_t)ldp]};
      } else {
        bv[c] = pb[(size_t)cc << 6];
      }
    }
  }
  acc += acc2;
  MV_STAMP_B(17, MV_STAMP_BLK);
  if (COLSUM) {
    float cs = (cs4[0] + cs4[1]) + (cs4[2] + cs4[3]);  // rows 4 q + t of every block; then the four row quads q
    cs += __shfl_xor(cs, 16);
    cs += __shfl_xor(cs, 32);
    if (lane < 16 && pr < NP) {
      cs_out[pr] = cs;
      if (ADAM) {
        adam1(cp, cs, cm, cv, neg_step, bc2s);
        cs_aa.p[pr] = cp;
        cs_aa.m[pr] = cm;
        cs_aa.v[pr] = cv;
      }
    }
  }
  if (ADAM) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pp = p0[r], mm = m0[r], vv = v0[r];
      adam1(pp, acc[r], mm, vv, neg_step, bc2s);
      p0[r] = pp;
      m0[r] = mm;
      v0[r] = vv;
    }
  }
  if (ok && vec) {
    store16_wt(out, idx, acc);  // write-through, see job_tn_wave
    if (ADAM) {
      store16_wt(aa.p, idx, p0);
      store16_wt(aa.m, idx, m0);
      store16_wt(aa.v, idx, v0);
    }
  } else if (ok) {
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
template <bool ADAM>
__device__ __forceinline__ void job_tn_frag(const float* PF, int pt, int NP, const float* QF, int qt, int NQ, int MB,
                                            float* out, int ldo, const AdamArgs& aa, const float* scal = nullptr) {
  job_tn_frag_any<ADAM>(PF, nullptr, 0, pt, NP, QF, qt, NQ, MB, out, ldo, aa, nullptr, AdamArgs{}, scal);
}
template <bool ADAM>
__device__ __forceinline__ void job_tn_halffrag(const float* Prow, int ldp, int pt, int NP, const float* QF, int qt, int NQ,
                                                int MB, float* out, int ldo, const AdamArgs& aa, const float* scal = nullptr) {
  job_tn_frag_any<ADAM>(Prow, Prow, ldp, pt, NP, QF, qt, NQ, MB, out, ldo, aa, nullptr, AdamArgs{}, scal);
}

// x [B][N] -> its fragment-order copy, one column tile per workgroup (any block size that is a multiple of 64): thread
// (block c, lane (i, q)) gathers its four batch rows and stores one 16-byte vector.
// ncols > 0: columns past ncols (a ragged last tile) repeat the last column; they only reach outputs nobody stores.
// nrows > 0: rows past nrows (padding rows, mvae_set_valid_rows) are stored as zeros.
__device__ __forceinline__ void job_frag_copy(const float* T, int ld, int nt, int MB, float* TF, int ncols = 0, int nrows = 0) {
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const int col = (ncols > 0 && 16 * nt + i >= ncols) ? ncols - 1 : 16 * nt + i;
  for (int c = threadIdx.x >> 6; c < MB; c += (int)(blockDim.x >> 6)) {
    const float* src = T + (size_t)(16 * c + 4 * q) * ld + col;
    f32x4 v;
    v[0] = src[0];
    v[1] = src[(size_t)ld];
    v[2] = src[2 * (size_t)ld];
    v[3] = src[3 * (size_t)ld];
    if (nrows > 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = 16 * c + 4 * q + t < nrows ? v[t] : 0.f;
    }
    store16_wt(TF, (((size_t)(nt * MB + c) << 6) + lane) << 2, v);  // read by a later launch only
  }
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
    // 8 rows per thread in flight, clamped requests and masks after the batch: with 20 row groups (320 threads) and 128
    // rows the former 4-deep batch left 2-3 rows per thread to a serial tail loop -- one memory round trip each
    for (int m = g; m < Mrows; m += ng * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int mm = m + ng * u;
        v[u] = col[(size_t)(mm < Mrows ? mm : 0) * ld];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (m + ng * u < Mrows) ? v[u] : 0.f;
    }
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

// XCD-aware tile assignment.  Workgroup L is observed to run on XCD L % 8 (MI355X_MICROARCH.md, "Workgroup dispatch");
// each XCD has a private L2 that does not survive a kernel boundary, so a weight block fetched by workgroups on all 8
// XCDs crosses the fabric 8 times per launch.  Weight blocks (= tiles nt of the weight dimension) are therefore dealt
// to XCDs in groups of G adjacent tiles: XCD k owns the groups k, k+8, ... and runs every row tile mt of those; the
// activation rows are the only operand every XCD fetches.  G = 1 for K-contiguous weights (NT layers: a tile's weight
// rows are whole cache lines); G = 2 when a 16-column tile covers only half of each 128-byte line of the weight (NN:
// dx = dy W), so that both halves of a line are wanted by the same XCD.  Placement only changes speed, never results.
// Launch with xcd_grid(NT, MT, G) workgroups; returns false for the padding workgroups.
__host__ __device__ inline int xcd_grid(int NT, int MT, int G = 1) {
  const int groups = (NT + G - 1) / G;
  return 8 * ((groups + 7) / 8) * MT * G;
}
// a / b for 0 <= a < 2^20, b > 0 through the float reciprocal (5 instructions; hipcc's exact 32-bit division is ~35
// dependent ones, and the tile index of a workgroup is computed BEFORE its first load can be issued).  Exact: (a + 0.5) / b
// is at least 0.5 / b away from an integer, far more than the rounding of the product.
__device__ __forceinline__ int fast_div(int a, int b) {
  return (int)(((float)a + 0.5f) * __frcp_rn((float)b));
}
__device__ __forceinline__ bool xcd_tile_g(int NT, int MT, int G, int* nt, int* mt, int L = blockIdx.x) {
  const int k = L & 7, s = L >> 3;
  const int per = MT * G;
  const int gi = fast_div(s, per), rem = s - gi * per;
  *mt = fast_div(rem, G);
  *nt = (k + 8 * gi) * G + (rem - (*mt) * G);
  return *nt < NT;
}
__device__ __forceinline__ bool xcd_tile(int NT, int MT, int* nt, int* mt, int L = blockIdx.x) {
  const int k = L & 7, s = L >> 3;
  const int gi = fast_div(s, MT);
  *nt = k + 8 * gi;
  *mt = s - gi * MT;
  return *nt < NT;
}

// Linear backward with FEW outputs and a WIDE input (the conv architecture's heads: dy [B, NH <= 16], x = the 8192-wide
// flatten): one pass over x.  Workgroup = 32 columns of K (256 workgroups for K = 8192); thread (row group g = tid >> 3,
// column quad c = tid & 7) walks rows g, g + 32, ... of every 256-row chunk: one 16-byte load of x feeds
// dW[:, quad] += dy[m][:] x[m][quad] (registers) and dx[m][quad] = (dy[m][:] W[:, quad]) [x > 0] (stored at once); the 32
// row groups' dW partials meet in LDS and are added in group order.  The last workgroup adds up the bias gradient.
// (As 16 x 16 tiles this was 8192 + 512 workgroups.)
// CL (the conv architecture's heads, conv_vae.py:65): x / dx are the CHANNEL-LAST flatten (column p * C + c of the
// [P = 16 pixels, C = K / 16 channels] activation) while W / dW keep the reference's NCHW-flatten column order c * 16 + p:
// the workgroup's 32 columns lie inside one pixel, a thread's quad is 4 consecutive channels, i.e. 4 entries of a row of W
// that are 16 floats apart.  (Re-ordering the 0.4 MB matrix per step instead cost two extra launches.)
constexpr int kSknN = 16;
// LDS of job_linear_bwd_skn (declared by the calling kernel, so that a kernel with several job classes can overlay them)
template <int NN>
struct SknLds {
  __attribute__((aligned(16))) float dy_s[256][NN];
  f32x4 smn[NN][32][9];
  __attribute__((aligned(16))) float w_s[NN][32];
};

template <int NN, bool CL>  // NN = N rounded up to a multiple of 4
__device__ __forceinline__ void job_linear_bwd_skn(SknLds<NN>& L, int blk, const float* x, const float* W, const float* dy,
                                                   float* dW, float* db, float* dx, int M, int N, int K, int relu_in,
                                                   unsigned short* dx_planes = nullptr, long long dx_ps = 0,
                                                   float* dx_colsum = nullptr) {
  auto& dy_s = L.dy_s;
  auto& smn = L.smn;
  const int tid = threadIdx.x, c = tid & 7, g = tid >> 3;
  if (blk == K / 32) {  // db[n] = sum_m dy[m][n]
    const int cn = tid & 15, gn = tid >> 4;
    float s = 0.f;
    if (cn < N)
      for (int m = gn; m < M; m += 16) s += dy[(size_t)m * N + cn];
    float* sf = reinterpret_cast<float*>(&smn[0][0][0]);
    sf[gn * 17 + cn] = s;
    __syncthreads();
    if (gn == 0 && cn < N) {
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += sf[q * 17 + cn];
      db[cn] = t;
    }
    return;
  }
  const int col = blk * 32 + c * 4;
  const int Cch = K >> 4, wp = CL ? col / Cch : 0, wc = CL ? col - wp * Cch : 0;  // CL: pixel and first channel of the quad
  // Latency, not bandwidth, bounds this kernel (one wave per SIMD, 16 MB moved): EVERY request of a 256-row chunk -- the
  // dy block, the W columns, the 8 rows of x -- is issued before the first use, so a chunk costs one memory round trip.
  // The workgroup's NN x 32 weights: in the channel-last form they sit 64 bytes apart in W (the reference's column order
  // c * 16 + p); fetched per thread (4 NN scalar loads, 8 lines per wave-level request: 1536 line accesses per workgroup,
  // ~1 us of the CU's address path in front of everything else) -- staged once through LDS, 2 requests per wave.
  f32x4 acc[NN], wr[NN];
  f32x4 dxs = {0.f, 0.f, 0.f, 0.f};  // column sums of dx over this thread's rows (dx_colsum)
  constexpr int kWPer = (NN * 32 + 255) / 256;
  float wst[kWPer];
#pragma unroll
  for (int q = 0; q < kWPer; ++q) {
    const int e = tid + 256 * q, n = e >> 5, j = e & 31;  // j: column blk * 32 + j
    wst[q] = 0.f;
    if (e < NN * 32) {
      const int cj = blk * 32 + j;
      const float* wrow = W + (size_t)(n < N ? n : 0) * K;
      wst[q] = CL ? wrow[(cj - (cj / Cch) * Cch) * 16 + cj / Cch] : wrow[cj];
    }
  }
#pragma unroll
  for (int n = 0; n < NN; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  (void)wp; (void)wc;
  constexpr int kDyPer = NN;  // 256 rows x NN entries / 256 threads
  for (int m0 = 0; m0 < M; m0 += 256) {
    const int rows = (M - m0) < 256 ? (M - m0) : 256;
    float dyv[kDyPer];
#pragma unroll
    for (int q = 0; q < kDyPer; ++q) {
      const int e = tid + 256 * q, r = e / NN, n = e - r * NN;
      dyv[q] = dy[(size_t)(m0 + (r < rows ? r : 0)) * N + (n < N ? n : 0)];
    }
    f32x4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = g + 32 * u;
      xv[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + (r < rows ? r : 0)) * K + col);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // the previous chunk's readers of dy_s are done
#pragma unroll
    for (int q = 0; q < kDyPer; ++q) {
      const int e = tid + 256 * q, r = e / NN, n = e - r * NN;
      dy_s[r][n] = (r < rows && n < N) ? dyv[q] : 0.f;
    }
    if (m0 == 0) {
#pragma unroll
      for (int q = 0; q < kWPer; ++q) {
        const int e = tid + 256 * q;
        if (e < NN * 32) L.w_s[e >> 5][e & 31] = wst[q];
      }
    }
    __syncthreads();
    if (m0 == 0) {
#pragma unroll
      for (int n = 0; n < NN; ++n) wr[n] = *reinterpret_cast<const f32x4*>(&L.w_s[n][c * 4]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = g + 32 * u;
      if (r >= rows) continue;
      f32x4 dxv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n4 = 0; n4 < NN; n4 += 4) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dy_s[r][n4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[n4 + j] += d[j] * xv[u];
          if (n4 + j < N) dxv += d[j] * wr[n4 + j];  // (uniform; rows of W past N were clamped to row 0)
        }
      }
      if (dx || dx_planes || dx_colsum) {
        if (relu_in) {
#pragma unroll
          for (int j = 0; j < 4; ++j) dxv[j] = (xv[u][j] > 0.f) ? dxv[j] : 0.f;
        }
        if (dx) *reinterpret_cast<f32x4*>(dx + (size_t)(m0 + r) * K + col) = dxv;
        if (dx_planes)  // the bf16 planes of dx for a consumer on pre-split operands (mvae_p3.hpp)
          store_planes4(dx_planes, dx_ps, (size_t)(m0 + r) * K + col, dxv[0], dxv[1], dxv[2], dxv[3]);
        dxs += dxv;  // (this thread's rows in order; the 32 row groups meet below)
      }
    }
  }
  // the 32 row groups' partial dW rows meet in LDS: all NN outputs at once (one barrier); two threads per (output, column
  // quad) add 16 row groups each in group order and meet by a lane swap (96 threads adding 32 groups each were a 1.3 us tail)
  __syncthreads();  // (the last chunk's readers of dy_s are done: smn may alias nothing, but keep the phases apart)
#pragma unroll
  for (int n = 0; n < NN; ++n) smn[n][g][c] = acc[n];
  __syncthreads();
  for (int o0 = 0; o0 < NN; o0 += 16) {
    const int n = o0 + (tid >> 4), c2 = (tid >> 1) & 7, hsel = tid & 1;
    if (n < NN) {  // (pairs of lanes share n)
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < 16; ++q) t += smn[n][hsel * 16 + q][c2];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += lane_swap1(t[e]);
      if (hsel == 0 && n < N) {
        const int col2 = blk * 32 + c2 * 4;
        if (CL) {
          const int wp2 = col2 / Cch, wc2 = col2 - wp2 * Cch;
#pragma unroll
          for (int j = 0; j < 4; ++j) dW[(size_t)n * K + (wc2 + j) * 16 + wp2] = t[j];
        } else {
          *reinterpret_cast<f32x4*>(dW + (size_t)n * K + col2) = t;
        }
      }
    }
  }
  if (dx_colsum) {  // sum_m dx[m][col .. col + 3]: this workgroup owns its 32 columns over ALL rows, so this is the whole sum
    __syncthreads();
    smn[0][g][c] = dxs;
    __syncthreads();
    if (tid < 16) {
      const int c2 = tid >> 1, hsel = tid & 1;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < 16; ++q) t += smn[0][hsel * 16 + q][c2];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += lane_swap1(t[e]);
      if (hsel == 0) *reinterpret_cast<f32x4*>(dx_colsum + blk * 32 + c2 * 4) = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ device-side input pipeline
// Row f-2 of the scope table: the reference feeds the step from 8 DataLoader worker processes that binarise every image
// on the CPU (mt/data/image_reconstruction.py:44-53,70-74) plus a host->device copy, and draws eps with the torch RNG
// inside the step.  Here the data set lives in HBM as uint8; a batch is gathered by a device-resident permutation,
// binarised dynamically (x = pixel/255 > U(0,1)) and eps ~ N(0,1) is drawn, both from a counter-based Philox4x32-10
// stream keyed by (seed, batch cursor) -- no host work, so a whole epoch can be replayed as HIP graphs.  The work is
// cut into ITEMS of four consecutive values (one Philox call each); who runs the items is the caller's choice: the
// stand-alone launch of mvae_prepare_batch (mvae_conv.hip) or spare workgroups of launch 4 of the PREVIOUS step
// (mvae_set_next_batch_feed, mvae_step.hip), which write the same bits.
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

// Four N(0, 1) draws from four 32-bit words: Box-Muller on two pairs.
// Both uniforms on the OPEN interval, (k + 1/2) 2^-23: with u1 in (0, 1] a pair is exactly (0, 0) whenever u1 = 1 --
// probability 2^-24 per pair, i.e. about once per 50 MNIST epochs -- and a sphere component whose eps is (0, 0) is
// 0 / 0 in the reference's formula (spherical.py:87-88 divides by |u| unclamped): the float32 CLI run went non-finite
// at a random epoch in 2 of 16 seeds, at exactly the batches tools/eps_zero_scan.py finds such a pair in.  (The reference draws
// float64 normals by default, where the same event has probability 2^-53.)  The half-step offset also keeps sin / cos
// off their exact zeros, so no single draw is exactly 0 either.
// (23 random bits: k + 1/2 is then exact in float32 -- with 24 bits the largest k + 1/2 rounds up to 2^24 and u1 is 1 again)
__device__ __forceinline__ void box_muller4(const unsigned r[4], float n[4]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float u1 = ((float)(r[2 * t] >> 9) + 0.5f) * (1.0f / 8388608.0f);      // (0, 1)
    const float u2 = ((float)(r[2 * t + 1] >> 9) + 0.5f) * (1.0f / 8388608.0f);  // (0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    n[2 * t] = rad * cs;
    n[2 * t + 1] = rad * sn;
  }
}

struct FeedArgs {
  const unsigned char* images;  // [n_images][D] uint8; NULL: no feed
  const int* perm;              // device permutation or NULL
  const int* counters;          // counters[8] = batch cursor
  float* x;                     // [B][D]
  float* eps;                   // [B][E]
  unsigned long long seed;
  int n_images, D, B, E, batches_per_epoch, mode, n_wg;
};

__device__ __forceinline__ int feed_items(const FeedArgs& f) { return (f.B * f.D + 3) / 4 + (f.B * f.E + 3) / 4; }

// ToTensor's x / 255 as the correctly rounded float32 quotient: through double (the build's -freciprocal-math may turn a
// float division into x * (1 / 255), one ulp off for some pixel values; the double product's error is 2^-29 of a float
// ulp and no pixel value lies that close to a rounding boundary -- tests/test_input_pipeline_gpu.py checks all 256)
__device__ __forceinline__ float feed_pixel(unsigned char v) { return (float)((double)v * (1.0 / 255.0)); }

__device__ __forceinline__ float feed_value(int mode, float pix, unsigned r) {
  const float u = (float)(r >> 8) * (1.0f / 16777216.0f);  // [0,1)
  return mode == 2 ? pix : ((mode ? (pix > u) : (pix > 0.5f)) ? 1.0f : 0.0f);
}

__device__ __forceinline__ void feed_item(const FeedArgs& f, unsigned cursor, int i) {
  const int bi = (int)(cursor % (unsigned)f.batches_per_epoch);
  const unsigned k0 = (unsigned)f.seed, k1 = (unsigned)(f.seed >> 32);
  const int nx4 = (f.B * f.D + 3) / 4, D = f.D;
  unsigned r[4];
  if (i < nx4) {
    philox4x32_10((unsigned)i, cursor, 0u, 0u, k0, k1, r);  // stream 0: binarisation
    const int e0 = i * 4;
    if ((D & 3) == 0 && (((size_t)f.images | (size_t)f.x) & 3) == 0) {  // the four values lie in one row: one request each way
      const int b = e0 / D, j = e0 - b * D;
      const int src = f.perm ? f.perm[(size_t)bi * f.B + b] : (bi * f.B + b);
      const unsigned pk =
          *reinterpret_cast<const unsigned*>(f.images + (size_t)(src < f.n_images ? src : f.n_images - 1) * D + j);
      float v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = feed_value(f.mode, feed_pixel((unsigned char)(pk >> (8 * t))), r[t]);
      if ((((size_t)f.x) & 15) == 0) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(f.x + e0) = o;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) f.x[e0 + t] = v[t];
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = e0 + t;
      if (e < f.B * D) {
        const int b = e / D, j = e - b * D;
        const int src = f.perm ? f.perm[(size_t)bi * f.B + b] : (bi * f.B + b);
        const float pix = feed_pixel(f.images[(size_t)(src < f.n_images ? src : f.n_images - 1) * D + j]);
        f.x[e] = feed_value(f.mode, pix, r[t]);
      }
    }
  } else {
    const int q = i - nx4;
    philox4x32_10((unsigned)q, cursor, 1u, 0u, k0, k1, r);  // stream 1: eps
    float n[4];
    box_muller4(r, n);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (q * 4 + t < f.B * f.E) f.eps[q * 4 + t] = n[t];
  }
}
