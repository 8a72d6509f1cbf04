// mvae_step.hip -- the fused ELBO step of mvae on gfx950 (C ABI: include/mvae_hip.h, "The whole step").
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
// The BASELINE MLP shapes (heads_dim <= 16, z_dim <= 8, B a multiple of 128, B <= 256) take FOUR launches instead:
// 1, k_fwd23 (2 + 3 in one), 4 (which also sums dz), and k_bwd56 (5 + 6 in one) -- see those kernels.
// In the single-GPU step the optimizer runs in the gradient epilogues, each weight one launch after its last read;
// the two-call path (mvae_step_forward_backward -> all-reduce -> mvae_step_optimizer) uses k_optim instead.
// Nothing here synchronises or allocates, so the host layer can capture any number of steps into one HIP graph.
// Further down: manifold primitives, component operators, generic dense layers, log-likelihood helpers and the
// patch-matrix gathers of the conv architecture -- the rest of the C ABI.
#ifndef MV_PREFETCH_WGS
#define MV_PREFETCH_WGS 0  // L2 prefetch workgroups of k_fwd23 (an experiment, see the kernel)
#endif
#include "mvae_common.hpp"
#include "mvae_coop.hpp"
#include <atomic>
constexpr int kMaxDevices = 64;  // per-device "attribute set" marks of kernels that need more than 64 KB of dynamic LDS
struct StatsArgs {  // job_step_stats outside launch 4; bce_part == NULL: not here
  const float* bce_part;
  const float* kl;
  float* bce_user;
  float* stats;
  int ncomp;
  int valid_rows;  // mvae_set_valid_rows: rows past it are padding (k_latent_bwd_blk zeroes their derivative terms)
};
__device__ __forceinline__ void job_step_stats(float* sm, const float* bce_part, const float* kl, float* bce_user, float* stats,
                                               float beta, int B, int ntD, int ncomp);
#include "mvae_step_blk.hpp"

// ---- the four-launch step (k_bwd56 below): what the forward launch and launch 4 hand to it
constexpr int kRecVecMax = 3;   // 16-byte vectors per dual record: {d kl, d z_0 .. d z_{A-1}}, A <= 9 on the fused-forward path
constexpr int kRecRad = 8;      // radius-direction records per row (ncomp <= 8 on that path)
constexpr double kDzScale = 274877906944.0;  // 2^38: dz partial sums as 64-bit fixed point (|partial| < 2^19, 25 .. 32 of them)
constexpr float kDzLimit = 524288.f;         // 2^19
struct Rec4Args {
  float* recH;        // head-direction records in the consumer's lane order (NULL: the records go to `duals`)
  float* recR;        // radius-direction records [B][kRecRad][NV] vectors
  long long* dzfix;   // [B][8] fixed-point sums of dz, then one unsigned: the overflow / non-finite mark
  int NV;             // vectors per record, (max ambient dim + 1 + 3) / 4
  int Bv;             // valid rows (mvae_set_valid_rows): rows past it are padding -- no loss, no KL, no gradient
};

// ================================================================================================ the fused step
struct mvae_ctx {
  mvae_model_desc d;
  CompTable t;
  int dmax;
  int ldh;    // heads row stride (NH rounded up to 4)
  int ldz;    // z row stride
  // workspace carve (floats)
  int64_t o_h, o_heads, o_z, o_hd, o_g, o_bce_part, o_kl, o_dhd, o_dz, o_dheads, o_dh, o_drpart, o_duals, o_dirtab, o_total;
  int64_t o_hdF, o_xF, o_hF, o_dhdF, o_zF, o_dheadsF, o_dhF;  // fragment-order operands of the lite backward (mvae_common.hpp: frag_off)
  int64_t o_dzp, o_whF;  // partial dz products of launch 4's tiles (block backward: [B/16][H/16][z tiles <= 4][64][4]); W_heads snapshot
  int64_t o_recH, o_recR, o_dzfix;   // the four-launch step: dual records per head column / radius, fixed-point dz sums
  int rec_nv;                        // 16-byte vectors per dual record there: (max ambient dimension + 1 + 3) / 4
  bool no_lite;                      // MVAE_NO_LITE=1: the fused-forward shapes keep the round-4 backward launches (A/B measurements)
  int nt_d, nt_h, nt_b;  // 16-wide tile counts of D, H, B
  bool no_fwd23;         // MVAE_NO_FWD23=1: keep launches 2 and 3 separate (A/B measurements)
  GroupTable gt;         // component groups of the 16-row block kernels (mvae_step_blk.hpp)
  bool groups_ok;        // every component fits one 16-column head tile
  bool no_blk;           // MVAE_NO_BLK=1: per-row latent kernels for many-component models too (A/B measurements)
  bool blk_small;        // MVAE_BLK_SMALL=1: the block backward kernel also for z_dim <= 16 (the fused-forward configs)
  bool blk_fwd;          // block kernels in the forward launches as well (MVAE_BLK_FWD=0: per-row forward, A/B measurements)
  bool coop;             // large components (true dim >= 9): wave-cooperative kernels (MVAE_NO_COOP=1: off)
  FeedArgs feed;         // mvae_set_next_batch_feed: the batch launch 4 of the NEXT step prepares (images == NULL: none)
  int valid_rows;        // mvae_set_valid_rows: rows [valid_rows, batch) of x / eps are PADDING (four-launch step only)
};

static int latent_path(const mvae_ctx* c, bool x_aligned);
static bool uses_blk_bwd(const mvae_ctx* c, bool x_aligned);

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
  c->o_dirtab = take(4 * (int64_t)kBlkDirs);  // int4 per active input direction (k_latent_bwd_blk)
  c->o_hdF = take(B * H);
  c->o_xF = take(B * D);
  c->o_hF = take(B * H);
  c->o_dhdF = take(B * H);
  c->o_dhF = take(B * H);
  c->o_zF = take(B * 16 * (((int64_t)d.z_dim + 15) / 16));  // whole 16-column tiles
  c->o_dheadsF = take(B * 16 * (((int64_t)d.heads_dim + 15) / 16));  // whole 16-column tiles (zero past heads_dim)
  c->o_dzp = take(B * (int64_t)c->nt_h * 64);  // [B/16][H/16][z tiles <= 4][64][4] (block backward)
  c->o_whF = take((int64_t)c->nt_h * 256);
  c->o_recH = take(B * 64 * kRecVecMax);
  c->o_recR = take(B * kRecRad * 4 * kRecVecMax);
  c->o_dzfix = take(B * 16 + 64);  // [B][8] 64-bit sums + the overflow mark
  c->o_total = o;
}

// the active-direction table of the block kernels lives in the workspace; (re)written whenever the component table is
static int upload_dirtab(mvae_ctx* c) {
  int4 tab[kBlkDirs];
  memset(tab, 0, sizeof(tab));
  fill_dirtab(c->t, tab);
  hipError_t e = hipMemcpy(c->d.workspace + c->o_dirtab, tab, sizeof(tab), hipMemcpyHostToDevice);
  return e == hipSuccess ? 0 : hip_fail(e, "hipMemcpy(direction table)");
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
  const char* nf = getenv("MVAE_NO_FWD23");
  c->no_fwd23 = nf && nf[0] && nf[0] != '0';
  const char* nb = getenv("MVAE_NO_BLK");
  c->no_blk = nb && nb[0] && nb[0] != '0';
  const char* bs = getenv("MVAE_BLK_SMALL");
  c->blk_small = bs && bs[0] && bs[0] != '0';
  const char* bf = getenv("MVAE_BLK_FWD");
  c->blk_fwd = !(bf && bf[0] == '0');
  const char* nl = getenv("MVAE_NO_LITE");
  c->no_lite = nl && nl[0] && nl[0] != '0';
  {
    int amax = 1;
    for (int i = 0; i < desc->ncomp; ++i) {
      const int A = ambient_dim(desc->comps[i].kind, desc->comps[i].true_dim);
      amax = A > amax ? A : amax;
    }
    c->rec_nv = (amax + 1 + 3) / 4;
  }
  c->groups_ok = build_groups(c->t, &c->gt);
  const char* nc = getenv("MVAE_NO_COOP");
  c->coop = bucket_of(c->dmax) > 8 && coop_eligible(c->t) && !(nc && nc[0] && nc[0] != '0');
  carve(c, bucket_of(c->dmax));
  c->valid_rows = desc->batch;
  // the only device access of create, and only for models that take the block kernels
  if (uses_blk_bwd(c, true) && (rc = upload_dirtab(c)) != 0) {
    delete c;
    return rc;
  }
  *out = c;
  return 0;
}

extern "C" void mvae_destroy(mvae_ctx* ctx) { delete ctx; }

// Rows [valid_rows, batch) of x / eps become PADDING: they contribute no reconstruction term, no KL term and no gradient, and
// the device-side input pipeline prepares valid_rows rows per batch.  For batch sizes that are not a multiple of 16 (the
// reference CLI's default is 100, mt/examples/run.py:32): the caller rounds the batch up, keeps the padding rows of its x / eps
// buffers finite (zeros), and the step runs on the fused kernels instead of the one-row-per-workgroup ones (B = 100: 28.5
// against 39.4 us per step).  Only the four-launch step masks; everything else declines.
extern "C" int mvae_set_valid_rows(mvae_ctx* c, int valid_rows) {
  if (!c) return fail(MVAE_E_BADARG, "null ctx%s", "");
  const mvae_model_desc& d = c->d;
  if (valid_rows < 1 || valid_rows > d.batch) return fail(MVAE_E_BADARG, "valid_rows outside [1, batch]%s (%lld)", "", valid_rows);
  if (valid_rows < d.batch) {
    const bool four = latent_path(c, true) == MVAE_PATH_FUSED && !uses_blk_bwd(c, true) && !c->no_lite && d.batch <= 256 &&
                      d.ncomp <= kRecRad && c->rec_nv <= kRecVecMax;
    // ... or the fragment-order block kernels (many small components, z_dim 17 .. 64, block forward on)
    const bool blk = !c->no_lite && uses_blk_bwd(c, true) && d.z_dim > 16 && d.z_dim <= 64 &&
                     latent_path(c, true) == MVAE_PATH_BLOCK && c->blk_fwd && (d.batch % 16 == 0) && (d.h_dim % 16 == 0) &&
                     (d.in_dim % 16 == 0);
    if (!four && !blk) return MVAE_E_UNSUPPORTED;  // (quietly: the caller then steps on exactly valid_rows rows)
  }
  c->valid_rows = valid_rows;
  c->feed = FeedArgs{};
  return 0;
}

extern "C" int mvae_set_radius_trainable(mvae_ctx* c, const uint8_t* trainable) {
  if (!c || !trainable) return fail(MVAE_E_BADARG, "null ctx / trainable%s", "");
  mvae_component_desc comps[kMaxComp];
  const int n = c->t.n;
  for (int i = 0; i < n; ++i) comps[i] = c->t.c[i];
  const int rc = fill_table(&c->t, comps, n, trainable, &c->dmax);
  if (rc) return rc;
  return uses_blk_bwd(c, true) ? upload_dirtab(c) : 0;
}

#ifdef MV_DBG_TIMING
extern "C" int mvae_debug_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * n);
}
extern "C" int mvae_debug_read_spans(unsigned long long* out /* [6][3][2048] */) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 6 * 3 * 2048);
}
#endif

// ---- 1: encoder layer (512 threads).  In the fused single-GPU step, workgroup (0,0) also advances the step counter.
template <bool FULL>
__global__ __launch_bounds__(512) void k_enc_fwd(const float* x, const float* W, const float* b, float* h, int B, int H,
                                                 int D, int* counters, int bump_step, double lr, float* xF, float* hF) {
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
  if (!real) {
    // The padding workgroups of the XCD-aware grid have a CU each and nothing to do: they write the fragment-order copy of x
    // that launch 6's dW_e0 tiles contract with (xF != NULL only when the grid has padding workgroups).
    if (xF) {
      const int NT = (H + 15) / 16, MT = (B + 15) / 16;
      const int npad = (8 * ((NT + 7) / 8) - NT) * MT;
      for (int j = (nt - NT) * MT + mt; j < (D >> 4); j += npad) job_frag_copy(x, D, j, B >> 4, xF);
    }
    return;
  }
  const bool vx = aligned16(x) && (D & 3) == 0, vw = aligned16(W) && (D & 3) == 0;
  // the epilogue's operand is requested with the tile operands (clamped address, no branch): asked for after the
  // contraction it costs the epilogue a memory round trip of its own
  const int n_ep = nt * 16 + (threadIdx.x & 15);
  const float bias = b[n_ep < H ? n_ep : 0];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_nt<7, FULL>(x, D, B, mt * 16, W, D, H, nt * 16, D, wave, kW8, vx, vw, acc);
  const float s = reduce_tiles8(red, acc);
  if (threadIdx.x < 256) {
    const int m = mt * 16 + (threadIdx.x >> 4), n = n_ep;
    if (FULL || (m < B && n < H)) {
      const float v = s + bias;
      h[(size_t)m * H + n] = v < 0.f ? 0.f : v;  // torch.relu: NaN propagates
      if (hF) hF[frag_off(m, n, B >> 4)] = v < 0.f ? 0.f : v;  // fragment order: launch 6 (dh's ReLU mask, dW_heads tiles)
    }
  }
  MV_SPAN_END(0, 1);
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
// COOP: large components on the wave-cooperative form (mvae_coop.hpp); the per-lane component code (whose vectors live in
// scratch memory beyond d = 8) is not instantiated then, so the kernel needs no scratch at all.
template <int DMAX, bool FAST, bool COOP = false>
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
  if (COOP && tid >= 256) return;  // large components: the records come from job_duals_coop (one WAVE per (row, direction), launch 3)
  if (!COOP && tid >= 256) {
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
        float kld = 0.f;
        if constexpr (!COOP) kld = comp_dual_dir<DMAX>(c, heads_s, eps_s, rad_s, rem, zd);
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

  // ---- latent components.  Large true dimensions (coop): one WAVE per component, lane i = entry i of the ambient vectors
  // (mvae_coop.hpp); otherwise one lane per component, placed by fill_table (kinds on different waves)
  if constexpr (COOP) {
    for (int ci = wave; ci < t.n; ci += 4) {
      const mvae_component_desc c = desc_s[ci];
      const int d = c.true_dim, j = lane - 1;
      const bool act = lane >= 1 && lane <= d;
      const float m = act ? heads_s[c.mean_col + j] : 0.f;
      const float l = act ? heads_s[c.logvar_col + (c.logvar_dim == 1 ? 0 : j)] : 0.f;
      const float e = act ? eps_s[c.eps_col + j] : 0.f;
      const float rp = c.kind == kEuclidean ? 0.f : rad_s[c.radius_idx];
      float zl = 0.f, klv = 0.f;
      (void)coop_eval<float>(c.kind, m, l, e, rp, d, lane, &zl, &klv);
      const int idx = coop_z_shifted(c.kind) ? j : lane;
      if (idx >= 0 && idx < ambient_dim(c.kind, d)) {
        z_s[c.z_col + idx] = zl;
        z[row * ldz + c.z_col + idx] = zl;
      }
      if (lane == 0) {
        kl[(size_t)ci * B + row] = klv;
        if (kl_user) kl_user[(size_t)ci * B + row] = klv;
      }
    }
  } else {
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

// ---- forward-mode dual records of LARGE components: one WAVE per (row, active input direction) evaluates the component
// over dual numbers in the lane-distributed form of mvae_coop.hpp and writes {d kl, d z_0 .. d z_{A-1}} -- what the dual
// waves of k_latent_fwd produce with one lane per record, whose 41-entry vectors (h40) live in scratch memory there.
// They have no consumer before launch 5, so they ride as extra workgroups BEHIND the tiles of launch 3 (k_dec1_fwd_duals):
// one launch less (h40: 64.8 -> 61.3 us per step).  The records are throughput-bound (10 368 waves for h40, 13.8 us of the
// whole chip): spreading them over launches 3 and 4 hides nothing (50 / 50: 62.7 us, 35 / 65: 61.7).
struct CoopDualArgs {
  const float* heads;
  const float* eps;
  const float* radii;
  float* duals;
  int ldh, eps_ld, NH, DS, ngroups, n_tile_wg;
};
__device__ __forceinline__ void job_duals_coop(const CompTable& t, const float* heads, int ldh, const float* eps, int eps_ld,
                                               const float* radii, float* duals, int NH, int DS, int ngroups, int wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = wg / ngroups, grp = wg - row * ngroups;
  const int gd = __builtin_amdgcn_readfirstlane(grp * 8 + wave);
  if (gd >= t.total_dirs) return;
  int ci = 0;
  while (gd >= t.dir_off[ci + 1]) ++ci;
  const int dir = gd - t.dir_off[ci];
  const mvae_component_desc c = t.c[ci];
  const int d = c.true_dim, lvd = c.logvar_dim, j = lane - 1;
  const bool act = lane >= 1 && lane <= d;
  const float* hrow = heads + (size_t)row * ldh;
  const float mv_ = hrow[c.mean_col + (act ? j : 0)];
  const float lv_ = hrow[c.logvar_col + ((act && lvd != 1) ? j : 0)];
  const float ev_ = eps[(size_t)row * eps_ld + c.eps_col + (act ? j : 0)];
  const float rv_ = c.kind == kEuclidean ? 0.f : radii[c.radius_idx];
  const Dual m{act ? mv_ : 0.f, (act && dir == j) ? 1.f : 0.f};
  const Dual l{act ? lv_ : 0.f, (act && (lvd == 1 ? dir == d : dir == d + j)) ? 1.f : 0.f};
  const Dual rp{rv_, dir == d + lvd ? 1.f : 0.f};
  Dual zl{0.f, 0.f}, klv{0.f, 0.f};
  (void)coop_eval<Dual>(c.kind, m, l, act ? ev_ : 0.f, rp, d, lane, &zl, &klv);
  float* rec = duals + ((size_t)row * (NH + t.n) + t.first_dir[ci] + dir) * DS;
  const int idx = coop_z_shifted(c.kind) ? j : lane;
  if (idx >= 0 && idx < ambient_dim(c.kind, d)) rec[1 + idx] = zl.d;
  if (lane == 0) rec[0] = klv.d;
}

// ---- 2+3 fused (the BASELINE MLP shapes): heads -> latent components -> first decoder layer -> output layer + BCE in ONE
// launch, by RECOMPUTING the per-row latent chain in every output tile's workgroup instead of handing it over through
// memory.  Workgroup (mt, nt) owns the 16 x 16 logits tile of row block mt; everything between `h` and that tile depends
// only on the 16 rows of the block:
//     heads[16, NH] = h[16, H] W_heads^T + b      one MFMA tile, K = H split over the 8 waves
//     z[16, Z], kl                                one lane per (row, component), the same device code as launch 2
//     hd[16, H]     = relu(z W_d0^T + b)          K = Z <= 8: one column per thread, 16 rows each, result kept in LDS
//     logits tile   = hd[16, H] W_logits[nt]^T    MFMA with the A operand read from LDS, B prefetched at kernel entry
// The 49 column tiles of a row block repeat the (tiny) chain 49 times -- the chip is idle otherwise -- and in exchange the
// step loses a kernel boundary (1.2-1.7 us), a cold first-load phase (~1.7 us) and the round trip of hd through memory;
// the W_logits / x / bias requests of the output tile are in flight while the chain runs.  The "lead" workgroup of a
// row block (nt == mt mod ntD: spread over the XCDs) writes what later launches need: heads, z, kl, hd.
// Forward-mode dual records for launch 5 are produced by extra workgroups of launch 4 (job_duals).
// Preconditions (checked on the host): NH <= 16, Z <= 8, eps_dim <= 8, ncomp <= 8 with at most 4 components per wave,
// H <= 416 (the staging of W_logits: kWl), B, H, D multiples of 16, 16-byte aligned operands.
template <int DMAX>
__global__ __launch_bounds__(512) void k_fwd23(CompTable t, const float* h, const float* Wh, const float* bh,
                                               const float* eps, int eps_ld, const float* radii, const float* Wd0,
                                               const float* bd0, const float* Wl, const float* bl, const float* x,
                                               float* heads, int ldh, float* z, int ldz, float* z_user, float* kl,
                                               float* kl_user, float* hd, float* g, float* bce_part,
                                               float* logits_user, int B, int H, int D, int NH, int Z, float* duals, float* zF,
                                               float* hdF, Rec4Args r4) {
  // dynamic LDS: hd_s[16][ld] | wl_s[32][ld] | wd_s[H][8] | bd_s[H]     (ld = H + 4: conflict-free ds_read_b128 of the
  // 16 rows an MFMA operand fetch touches)
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ float red[kW8][16][17];
  __shared__ float red2[kW8][16][17];  // second column tile of the pair
  __shared__ __attribute__((aligned(16))) float heads_s[16][16];
  __shared__ __attribute__((aligned(16))) float z_s[16][8];
  __shared__ __attribute__((aligned(16))) float eps_s[16][8];
  __shared__ float rad_s[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int mt, nt;
  MV_SPAN_BEGIN(2);
  // A workgroup owns TWO adjacent column tiles (16 x 32 logits): the per-row-block chain and its operands (h rows,
  // W_heads: 51 KB) are shared by twice the output, and the grid (8 row blocks x 25 pairs = 200 workgroups) puts exactly
  // one workgroup on a CU.
  const int ntD = D >> 4, ntP = (ntD + 1) >> 1;
  int pt = 0;
  // The first B/16 workgroups of the grid are DUAL workgroups, one per row block: they compute the heads of their 16
  // rows like everybody else and then the forward-mode dual records of the latent components for launch 5 (d kl and
  // d z along every input direction: a ~3.5 us dependent chain per lane, 16 rows x directions lanes).  They have a CU
  // to themselves (208 workgroups on 256 CUs) and finish well inside the launch.
  const int n_dual = B >> 4;  // dispatched FIRST (the longest job); a multiple of 8 keeps workgroup L on XCD L % 8
  const bool is_dual = (int)blockIdx.x < n_dual;
#ifndef MV_PREFETCH_WGS
#define MV_PREFETCH_WGS 0
#endif
  constexpr int n_pref = MV_PREFETCH_WGS;  // 0 or 48 (a multiple of 8): see below
  if (n_pref > 0 && !is_dual && (int)blockIdx.x < n_dual + n_pref) {
    // L2 PREFETCH workgroups (an experiment: MV_PREFETCH_WGS=48, the CUs the tile workgroups leave free).  The weights were
    // written write-through by the previous step's optimizer epilogues, so the first reader of a line in this launch
    // fetches it over the fabric.  Prefetcher j runs on XCD j % 8 and touches one 16-row block of W_logits that THIS
    // XCD's tile workgroups stage ~2.5 us into the launch (pairs x, x + 8, x + 16 of XCD x), plus W_d0.
    const int j = (int)blockIdx.x - n_dual, xk = j & 7, part = j >> 3;  // part 0..5
    const int tile = (xk + 8 * (part >> 1)) * 2 + (part & 1);
    const f32x4* src = reinterpret_cast<const f32x4*>(Wl + (size_t)tile * 16 * H);
    const int n4 = 4 * H;  // 16 rows x H / 4 vectors
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < n4; e += 512) sink += src[e];
    if (part == 0) {
      const f32x4* wd = reinterpret_cast<const f32x4*>(Wd0);
      for (int e = tid; e < (H * Z) >> 2; e += 512) sink += wd[e];
    }
    asm volatile("" ::"v"(sink));
    return;
  }
  if (is_dual) mt = (int)blockIdx.x;
  else {
    // XCD-aware without padding workgroups (a padding workgroup holds its CU's LDS allocation long enough to push a
    // real one into a second round): the first 8 * floor(ntP / 8) pairs are dealt to the XCDs as in xcd_tile, the
    // remaining pairs' workgroups follow in plain order (their W_logits rows are fetched by several XCDs: 50 KB each).
    const int L = (int)blockIdx.x - n_dual - n_pref, MT = B >> 4;
    const int full = (ntP >> 3) << 3;
    if (L < full * MT) (void)xcd_tile(full, MT, &pt, &mt, L);
    else {
      const int idx = L - full * MT;
      pt = full + fast_div(idx, MT);
      mt = idx - (pt - full) * MT;
    }
  }
  nt = pt * 2;
  const bool two = nt + 1 < ntD;  // the last pair of an odd tile count has one tile
  const bool lead = !is_dual && pt == mt - fast_div(mt, ntP) * ntP;
  MV_TDECL;
  MV_T(0);
  const int ld = H + 4;
  float* hd_s = dyn;
  float* wl_s = dyn + 16 * ld;
  float* wd_s = wl_s + 32 * ld;
  float* bd_s = wd_s + H * 8;
  const int i = lane & 15, q = lane >> 4;
  const int nchunks = H >> 4;

  // ---- requests of the FIRST phase only (branch-free, clamped addresses).  A CU's load path serves requests in issue
  // order at ~30 cycles per wave-level request: everything a later phase needs is fetched later, by waves that would
  // otherwise idle (see the component phase), so that nothing queues in front of these.
  const float* hrow = h + (size_t)(mt * 16 + i) * H;
  const float* whrow = Wh + (size_t)(i < NH ? i : 0) * H;
  float4 ha[4], hb[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave + 8 * gq;
    ha[gq] = make_float4(0.f, 0.f, 0.f, 0.f);
    hb[gq] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nchunks) {  // wave-uniform (scalar branch): a chunk this wave does not have is not requested
      const int k = (c << 4) + (q << 2);
      ha[gq] = *reinterpret_cast<const float4*>(hrow + k);
      hb[gq] = *reinterpret_cast<const float4*>(whrow + k);  // rows past NH re-read row 0: their columns are never used
    }
  }
  const float bhv = bh[(tid & 15) < NH ? (tid & 15) : 0];
  float epsv = 0.f;
  {
    const int r = (tid >> 3) & 15, j = tid & 7;
    epsv = eps[(size_t)(mt * 16 + r) * eps_ld + (j < eps_ld ? j : 0)];
  }
  const float rad_r = radii[tid < t.n ? tid : 0];
  // epilogue operands: threads 0..255 take the first tile of the pair, 256..511 the second
  const int nt_ep = (tid < 256 || !two) ? nt : nt + 1;
  const int m_ep = mt * 16 + ((tid & 255) >> 4), n_ep = nt_ep * 16 + (tid & 15);
  __builtin_amdgcn_sched_barrier(0);
  // this lane's component: lane = slot * 16 + row; the four slot descriptors of this wave are fetched with UNIFORM
  // (scalar) loads, all four in flight at once while the vector requests above travel, and selected per lane without a
  // branch.  (A per-lane index into the kernarg table would be a vector load that hipcc sinks to its first use.)
  mvae_component_desc my_desc;
  {
    const int wv = __builtin_amdgcn_readfirstlane(wave) & 3;
    const mvae_component_desc d0 = t.slot_desc[wv][0], d1 = t.slot_desc[wv][1], d2 = t.slot_desc[wv][2],
                              d3 = t.slot_desc[wv][3];
    const int sl = lane >> 4;
#define MV_SEL(f) my_desc.f = (sl == 0 ? d0.f : (sl == 1 ? d1.f : (sl == 2 ? d2.f : d3.f)))
    MV_SEL(kind); MV_SEL(true_dim); MV_SEL(mean_col); MV_SEL(logvar_col); MV_SEL(logvar_dim); MV_SEL(eps_col);
    MV_SEL(z_col); MV_SEL(radius_idx);
#undef MV_SEL
  }
  const int my_ci = (wave < 4 && my_desc.kind >= 0) ? my_desc.radius_idx : -1;  // radius_idx == component index here
  if (tid < 8) rad_s[tid] = rad_r;
  if (tid < 128) {
    eps_s[tid >> 3][tid & 7] = (tid & 7) < eps_ld ? epsv : 0.f;
    z_s[tid >> 3][tid & 7] = 0.f;  // columns past Z stay zero (K of the hd tiles is padded to 8)
  }
  // ---- staging state of waves 4..7 (operands of the hd and logits phases, requested by MV_STAGE_REQUESTS, written to LDS
  // after the heads barrier).  MV_EARLY_STAGE=1 (an experiment, not the default): the requests are issued right after the
  // wave's heads MFMAs, i.e. once its own phase-1 operands have landed, instead of after the heads reduction + barrier.
  // Measured slower (heads reduce 0.8 -> 1.5 us, components 2.2 -> 1.7 us, kernel 8.9 -> 9.4 us): the other waves'
  // phase-1 operands are still travelling and the CU's load path is the shared resource.
  const int lt = tid - 256;  // 0..255 on waves 4..7
  const int H4 = H >> 2;     // float4 per row of W_logits
  constexpr int kWl = 13;    // ceil(32 * 128 / 256): rows of up to 512 floats
  f32x4 wv[kWl];             // native vectors: whole-struct copies of HIP's float4 keep the array in scratch
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x4 dv[4];
  f32x2 dp[6];
  const int nd4 = (H * Z) >> 2, nd2 = (H * Z) >> 1;
  const bool quads = Z == 8 || Z == 4;  // uniform
  float bdv2[2];
#define MV_STAGE_REQUESTS()                                                                                           \
  do {                                                                                                                \
    /* W_logits rows nt*16 .. nt*16+31 (the second tile repeats the first when the pair is incomplete) */             \
    _Pragma("unroll") for (int u = 0; u < kWl; ++u) {                                                                 \
      const int e4 = lt + 256 * u;                                                                                    \
      const int r = e4 / H4, c4 = e4 - r * H4;                                                                        \
      const int rr = r < 32 ? r : 0;                                                                                  \
      const int grow = (rr < 16 || two) ? nt * 16 + rr : nt * 16 + rr - 16;                                           \
      wv[u] = *reinterpret_cast<const f32x4*>(Wl + (size_t)grow * H + 4 * (r < 32 ? c4 : 0));                         \
    }                                                                                                                 \
    /* W_d0 [H][Z] -> wd_s[H][8] (zero-padded columns), b_d0 -> bd_s.  Z == 8 / 4: whole 16-byte vectors; other even  \
       Z (Z == 6: BASELINE config [0], `e6`): 8-byte pairs, 6 x 256 x 2 floats cover H Z <= 3072 */                   \
    if (quads) {                                                                                                      \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                 \
        const int e4 = lt + 256 * u;                                                                                  \
        dv[u] = *reinterpret_cast<const f32x4*>(Wd0 + 4 * (size_t)(e4 < nd4 ? e4 : 0));                               \
      }                                                                                                               \
    } else {                                                                                                          \
      _Pragma("unroll") for (int u = 0; u < 6; ++u) {                                                                 \
        const int e2 = lt + 256 * u;                                                                                  \
        dp[u] = *reinterpret_cast<const f32x2*>(Wd0 + 2 * (size_t)(e2 < nd2 ? e2 : 0));                               \
      }                                                                                                               \
    }                                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                   \
      const int c = lt + 256 * u;                                                                                     \
      bdv2[u] = bd0[c < H ? c : 0];                                                                                   \
    }                                                                                                                 \
  } while (0)
#ifndef MV_EARLY_STAGE
#define MV_EARLY_STAGE 0  // measured (r03): 9.4 us against 8.9 us -- the early requests still delay the heads phase
#endif
  // ---- heads = h W_heads^T + b
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      if (wave + 8 * gq >= nchunks) ha[gq] = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = mfma16(ha[gq].x, hb[gq].x, acc);
      acc2 = mfma16(ha[gq].y, hb[gq].y, acc2);
      acc = mfma16(ha[gq].z, hb[gq].z, acc);
      acc2 = mfma16(ha[gq].w, hb[gq].w, acc2);
    }
#if MV_EARLY_STAGE
    if (wave >= 4 && !is_dual) {  // wave-uniform
      __builtin_amdgcn_sched_barrier(0);
      MV_STAGE_REQUESTS();
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    MV_T(1);
    const float sv = reduce_tiles8(red, acc + acc2);
    MV_T(2);
    if (tid < 256) {
      const int r = tid >> 4, n = tid & 15;
      const float v = n < NH ? sv + bhv : 0.f;
      heads_s[r][n] = v;
      if (lead && n < NH) heads[(size_t)(mt * 16 + r) * ldh + n] = v;
    }
  }
  lds_barrier();
  MV_T(3);

  if (is_dual) {
    // wave w and wave w + 4 share the components placed on wave w & 3 (one manifold kind: one instruction stream);
    // item = (direction index within those components) * 16 + row
    constexpr int AM = DMAX + 1, DS = DMAX + 2;
    const int wk = __builtin_amdgcn_readfirstlane(wave) & 3;
    int total = 0;
    for (int ci = 0; ci < t.n; ++ci)
      if (t.wave_of[ci] == wk) total += t.dir_off[ci + 1] - t.dir_off[ci];
    for (int item = (wave >> 2) * 64 + lane; item < total * 16; item += 128) {
      const int r = item & 15, dd = item >> 4;
      int mine = 0, mydir = 0, base = 0;
      for (int ci = 0; ci < t.n; ++ci) {  // uniform loop, per-lane select
        if (t.wave_of[ci] != wk) continue;
        const int nd = t.dir_off[ci + 1] - t.dir_off[ci];
        if (dd >= base && dd < base + nd) {
          mine = ci;
          mydir = dd - base;
        }
        base += nd;
      }
      const mvae_component_desc c = t.c[mine];
      float zd[AM];
      const float kld = comp_dual_dir<DMAX>(c, heads_s[r], eps_s[r], rad_s, mydir, zd);
      const int A = ambient_dim(c.kind, c.true_dim);
      if (r4.recH) {
        // the four-launch step (k_bwd56): a record = NV 16-byte vectors {d kl, d z_0 .. d z_{A-1}, 0 ...}.  Head directions
        // are indexed by the HEAD COLUMN hc that receives the derivative and stored in the order the consumer's lanes read
        // them -- lane (i = row & 15, q = hc >> 2) of row block mt takes vector v of column 4 q + tt at
        // ((mt * 4 + tt) * NV + v) * 64 + q * 16 + i: one 1 KB wave-level request per (tt, v); radius directions: recR[row][ci]
        float rv[4 * kRecVecMax];
        rv[0] = kld;
#pragma unroll
        for (int q2 = 0; q2 < 4 * kRecVecMax - 1; ++q2) rv[1 + q2] = (q2 < AM && q2 < A) ? zd[q2 < AM ? q2 : 0] : 0.f;
        const bool head_dir = mydir < c.true_dim + c.logvar_dim;
        const int hc = mydir < c.true_dim ? c.mean_col + mydir : c.logvar_col + (mydir - c.true_dim);
        f32x4* dst = head_dir
                         ? reinterpret_cast<f32x4*>(r4.recH) + ((size_t)(mt * 4 + (hc & 3)) * r4.NV << 6) + ((hc >> 2) << 4) + r
                         : reinterpret_cast<f32x4*>(r4.recR) + ((size_t)(mt * 16 + r) * kRecRad + mine) * r4.NV;
        const int vstride = head_dir ? 64 : 1;
#pragma unroll
        for (int v = 0; v < kRecVecMax; ++v)
          if (v < r4.NV) dst[(size_t)v * vstride] = f32x4{rv[4 * v], rv[4 * v + 1], rv[4 * v + 2], rv[4 * v + 3]};
      } else {
        float* rec = duals + (((size_t)mt * 16 + r) * (NH + t.n) + t.first_dir[mine] + mydir) * DS;
        rec[0] = kld;
#pragma unroll
        for (int q2 = 0; q2 < AM; ++q2)
          if (q2 < A) rec[1 + q2] = zd[q2];
      }
    }
    if (r4.dzfix && tid < 64) {
      // launch 4's tiles ADD their shares of dz to these fixed-point sums (k_dec1_bwd, LITE 1): zeroed here, one launch after
      // the previous step's last reader (k_bwd56) and one before the adds; this row block's 16 rows x 8 sums = 64 x 16 bytes
      typedef long long i64x2 __attribute__((ext_vector_type(2)));
      reinterpret_cast<i64x2*>(r4.dzfix + (size_t)mt * 128)[tid] = i64x2{0, 0};
      if (mt == 0 && tid == 0) *reinterpret_cast<unsigned*>(r4.dzfix + (size_t)B * 8) = 0u;  // the overflow / NaN mark
    }
    MV_SPAN_END(2, 2);
    return;
  }

  // the epilogue operands are requested only now: at the top of the kernel they were 16 of the ~100 wave-level requests
  // the heads phase waits behind (the CU's load path serves ~1 request per 36 cycles whatever its size)
  const float tv = x[(size_t)m_ep * D + n_ep];
  const float bias = bl[n_ep];
  __builtin_amdgcn_sched_barrier(0);
  // ---- waves 0..3: the latent components, one lane per (slot, row) -- a ~2 us dependent chain on a handful of lanes.
  // ---- waves 4..7 meanwhile stage the operands of the two remaining phases in LDS: W_d0 / b_d0 (first decoder layer) and
  // the two 16-row blocks of W_logits (B operands of the output tiles), 16-byte coalesced requests.
  if (wave < 4) {
    const int r = lane & 15;
    if (my_ci >= 0) {
      float klv;
      const size_t row = (size_t)mt * 16 + r;
      // (tried in round 3: the tile workgroups stop at z and the dual workgroup writes the KL terms from the value part
      // of its dual evaluation -- no gain, the phase is bound by the staging loads of waves 4..7, not by this chain; and
      // the value part of the dual evaluation differs from this one by rounding, which an ill-conditioned KL term of a
      // projected-sphere component turned into 4e-4)
      comp_fwd_row<DMAX>(my_desc, heads_s[r], eps_s[r], rad_s, z_s[r], lead ? z + row * ldz : nullptr, &klv, nullptr,
                         nullptr, nullptr, nullptr);
      if (lead) {
        const float klm = (int)row < r4.Bv ? klv : 0.f;  // (padding rows carry no KL term)
        kl[(size_t)my_ci * B + row] = klm;
        if (kl_user) kl_user[(size_t)my_ci * B + row] = klm;
      }
    }
  } else {
#if !MV_EARLY_STAGE
    MV_STAGE_REQUESTS();
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < kWl; ++u) {
      const int e4 = lt + 256 * u;
      const int r = e4 / H4, c4 = e4 - r * H4;
      if (r < 32) *reinterpret_cast<f32x4*>(wl_s + r * ld + 4 * c4) = wv[u];
    }
    if (quads) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e4 = lt + 256 * u;
        if (e4 < nd4) {
          if (Z == 8) {
            *reinterpret_cast<f32x4*>(wd_s + 4 * e4) = dv[u];
          } else {  // Z == 4: row c = e4, columns 0..3; columns 4..7 are zero
            *reinterpret_cast<f32x4*>(wd_s + 8 * e4) = dv[u];
            *reinterpret_cast<f32x4*>(wd_s + 8 * e4 + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else {
      const int zh = Z >> 1;  // pairs per row
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int e2 = lt + 256 * u;
        if (e2 < nd2) {
          const int c = e2 / zh, j2 = e2 - c * zh;
          *reinterpret_cast<f32x2*>(wd_s + 8 * c + 2 * j2) = dp[u];
        }
      }
      for (int e = lt; e < H * (8 - Z); e += 256) {  // zero padding of the columns past Z
        const int c = e / (8 - Z), j = Z + e - c * (8 - Z);
        wd_s[8 * c + j] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = lt + 256 * u;
      if (c < H) bd_s[c] = bdv2[u];
    }
  }
  lds_barrier();
  MV_T(4);
  if (lead && z_user && tid < 16 * Z) z_user[((size_t)mt * 16 + tid / Z) * Z + tid % Z] = z_s[tid / Z][tid % Z];
  if (lead && zF && tid < 64) {  // z in fragment order (one 16-column tile, zero past Z): launch 6's dW_d0 tiles
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < 8) v = f32x4{z_s[4 * q][i], z_s[4 * q + 1][i], z_s[4 * q + 2][i], z_s[4 * q + 3][i]};  // z_s is zero past Z
    // (padding rows: zero -- their z need not be finite (the sphere's sample at eps = 0 is 0 / 0, spherical.py:87-88), and
    // the batch contractions multiply it by a zero gradient)
#pragma unroll
    for (int r4i = 0; r4i < 4; ++r4i) v[r4i] = mt * 16 + 4 * q + r4i < r4.Bv ? v[r4i] : 0.f;
    store16_wt(zF, ((size_t)mt * 64 + tid) << 2, v);
  }

  // ---- hd = relu(z W_d0^T + b) as MFMA tiles (K = 8: two 16x16x4 steps per 16 columns), kept in LDS as the A operand
  // of the output layer.  A[i][k] = z[i][k] (zero past Z): lane (i, q) supplies k = q and q + 4.
  {
    const float za0 = z_s[i][q];
    const float za1 = z_s[i][q + 4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int c = wave + 8 * gq;
      if (c < nchunks) {  // wave-uniform
        const int col = (c << 4) + i;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = mfma16(za0, wd_s[col * 8 + q], a);
        a = mfma16(za1, wd_s[col * 8 + q + 4], a);
        const float bv = bd_s[col];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          float v = a[r4] + bv;
          v = v < 0.f ? 0.f : v;  // torch.relu: NaN propagates
          hd_s[(q * 4 + r4) * ld + col] = v;
        }
      }
    }
  }
  lds_barrier();
  MV_T(5);
  if (lead && !hdF) {  // launches 4 and 5 read hd (ReLU mask, operand of dW_logits): coalesced 16-byte rows
    for (int e4 = tid; e4 < 4 * H; e4 += 512) {
      const int r = e4 / (H >> 2), c4 = e4 - r * (H >> 2);
      *reinterpret_cast<float4*>(hd + ((size_t)mt * 16 + r) * H + 4 * c4) =
          *reinterpret_cast<const float4*>(hd_s + r * ld + 4 * c4);
    }
  }

  // ---- the two logits tiles = hd W_logits[nt, nt+1]^T (both operands from LDS), BCE-with-logits and its gradient
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, bcc = {0.f, 0.f, 0.f, 0.f}, bcc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave + 8 * gq;
    if (c < nchunks) {  // wave-uniform
      const int k = (c << 4) + (q << 2);
      const float4 av = *reinterpret_cast<const float4*>(hd_s + i * ld + k);
      const float4 b1 = *reinterpret_cast<const float4*>(wl_s + i * ld + k);
      const float4 b2 = *reinterpret_cast<const float4*>(wl_s + (16 + i) * ld + k);
      acc = mfma16(av.x, b1.x, acc);
      bcc = mfma16(av.x, b2.x, bcc);
      acc2 = mfma16(av.y, b1.y, acc2);
      bcc2 = mfma16(av.y, b2.y, bcc2);
      acc = mfma16(av.z, b1.z, acc);
      bcc = mfma16(av.z, b2.z, bcc);
      acc2 = mfma16(av.w, b1.w, acc2);
      bcc2 = mfma16(av.w, b2.w, bcc2);
    }
  }
  MV_T(6);
  if (hdF) {
    // hd leaves in FRAGMENT ORDER only (launch 4 reads its ReLU mask element-wise, launch 5's dW_logits tiles read whole
    // fragments): position (q', i') of the block of (column tile, this row block) = rows 4 q' .. 4 q' + 3 of column i'.
    // EVERY pair workgroup of the row block holds all of hd in LDS: each stores the column tiles t = pt, pt + ntP, ... (one
    // 1 KB block for the BASELINE shapes) instead of the lead storing all H / 16 of them on the launch's critical path.
    const int MB = B >> 4;
    if (wave == 0)
      for (int tile = pt; tile < (H >> 4); tile += ntP) {
        const int ii = lane & 15, qq = lane >> 4;
        const float* src = hd_s + (4 * qq) * ld + tile * 16 + ii;
        f32x4 v = {src[0], src[ld], src[2 * ld], src[3 * ld]};
#pragma unroll
        for (int r4i = 0; r4i < 4; ++r4i) v[r4i] = mt * 16 + 4 * qq + r4i < r4.Bv ? v[r4i] : 0.f;  // (padding rows: zero)
        reinterpret_cast<f32x4*>(hdF)[((size_t)(tile * MB + mt) << 6) + lane] = v;
      }
  }
  // both partial tiles go to LDS under one barrier; threads 0..255 add up the first tile, 256..511 the second
  float sv;
  {
    const f32x4 pa = acc + acc2, pb = bcc + bcc2;
    const int col = lane & 15, rbase = (lane >> 4) << 2;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      red[wave][rbase + r4][col] = pa[r4];
      red2[wave][rbase + r4][col] = pb[r4];
    }
    lds_barrier();
    const int r = (tid & 255) >> 4, c = tid & 15;
    float (*rr)[16][17] = tid < 256 ? red : red2;
    sv = ((rr[0][r][c] + rr[1][r][c]) + (rr[2][r][c] + rr[3][r][c])) +
         ((rr[4][r][c] + rr[5][r][c]) + (rr[6][r][c] + rr[7][r][c]));
  }
  MV_T(7);
  if (tid >= 256 && !two) return;
  const float y = sv + bias;
  // F.binary_cross_entropy_with_logits (image_reconstruction.py:81-82): (1-t)*y - log_sigmoid(y)
  const float e = expf(-fabsf(y));
  const float log_sig = fminf(y, 0.f) - mvf::log1p_pos(e);
  float loss = (1.f - tv) * y - log_sig;
  const float sig = (y >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
  loss += __shfl_xor(loss, 8, 16);  // sum over the tile's 16 columns (the 16 lanes of one row are contiguous)
  loss += __shfl_xor(loss, 4, 16);
  loss += __shfl_xor(loss, 2, 16);
  loss += __shfl_xor(loss, 1, 16);
  const bool vrow = m_ep < r4.Bv;  // (a padding row: no reconstruction term, and through g = 0 no gradient behind it)
  g[(size_t)m_ep * D + n_ep] = vrow ? sig - tv : 0.f;  // d(sum bce)/d(logit)
  if (logits_user) logits_user[(size_t)m_ep * D + n_ep] = y;
  if ((tid & 15) == 0) bce_part[(size_t)nt_ep * B + m_ep] = vrow ? loss : 0.f;
  MV_TFLUSH(24, 8, 96);
  MV_SPAN_END(2, 1);
}
#undef MV_STAGE_REQUESTS

// Forward-mode dual records of the latent components (d kl / d dir and d z / d dir for every input direction of every
// (row, component)), one thread per record: what waves 4..7 of k_latent_fwd produce in the six-launch step.  In the
// fused step they are computed by extra workgroups of launch 4, off every critical path (they need heads / eps / radii
// from the forward launch and are consumed by launch 5).
template <int DMAX>
__device__ __forceinline__ void job_duals(const CompTable& t, const float* heads, int ldh, const float* eps, int eps_ld,
                                          const float* radii, float* duals, int B, int NH, int item) {
  // item = direction-major: the 64 lanes of a wave hold 64 rows of ONE (component, direction), i.e. one manifold kind:
  // one instruction stream per wave, no divergence
  constexpr int AM = DMAX + 1, DS = DMAX + 2;
  const int total = t.total_dirs;
  if (item >= B * total) return;
  const int gd = item / B, row = item - gd * B;
  int ci = 0;
  while (gd >= t.dir_off[ci + 1]) ++ci;
  const int dir = gd - t.dir_off[ci];
  const mvae_component_desc c = t.c[ci];
  float zd[AM];
  const float kld = comp_dual_dir<DMAX>(c, heads + (size_t)row * ldh, eps + (size_t)row * eps_ld, radii, dir, zd);
  float* rec = duals + ((size_t)row * (NH + t.n) + t.first_dir[ci] + dir) * DS;
  const int A = ambient_dim(c.kind, c.true_dim);
  rec[0] = kld;
#pragma unroll
  for (int k = 0; k < AM; ++k)
    if (k < A) rec[1 + k] = zd[k];
}

// ---- 3: output layer + BCE-with-logits + its gradient (512 threads)
template <bool FULL>
__device__ __forceinline__ void job_dec1_fwd_tile(float (*red)[16][17], const float* hd, const float* W, const float* b,
                                                  const float* x, float* g, float* bce_part, float* logits_user, int B, int H,
                                                  int D) {
  const int wave = threadIdx.x >> 6;
  int mt, nt;
  MV_SPAN_BEGIN(2);
  if (!xcd_tile((D + 15) / 16, (B + 15) / 16, &nt, &mt)) return;
  MV_TDECL;
  MV_T(0);
  const int m = mt * 16 + ((threadIdx.x & 255) >> 4), n = nt * 16 + (threadIdx.x & 15);
  const bool ok = threadIdx.x < 256 && m < B && n < D;
  // epilogue operands requested up front, branch-free (clamped addresses; `ok` is applied in the epilogue)
  const float tv = x[(size_t)(m < B ? m : 0) * D + (n < D ? n : 0)];
  const float bias = b[n < D ? n : 0];
  const bool v1 = aligned16(hd) && (H & 3) == 0, v2 = aligned16(W) && (H & 3) == 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = tile_nt<4, FULL>(hd, H, B, mt * 16, W, H, D, nt * 16, H, wave, kW8, v1, v2, acc);
  MV_T(1);
  const float s = reduce_tiles8(red, acc);
  MV_T(2);
  if (threadIdx.x >= 256) return;
  const float y = s + bias;
  // F.binary_cross_entropy_with_logits (image_reconstruction.py:81-82): (1-t)*y - log_sigmoid(y)
  const float e = expf(-fabsf(y));
  const float log_sig = fminf(y, 0.f) - mvf::log1p_pos(e);
  float loss = ok ? (1.f - tv) * y - log_sig : 0.f;
  const float sig = (y >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
  // sum over the tile's 16 columns (the 16 lanes of one row are contiguous) BEFORE the stores: hipcc drains vmcnt in
  // front of the cross-lane operations, which would otherwise wait for the store acknowledgements
  loss += __shfl_xor(loss, 8, 16);
  loss += __shfl_xor(loss, 4, 16);
  loss += __shfl_xor(loss, 2, 16);
  loss += __shfl_xor(loss, 1, 16);
  if (ok) {
    g[(size_t)m * D + n] = sig - tv;  // d(sum bce)/d(logit)
    if (logits_user) logits_user[(size_t)m * D + n] = y;
  }
  MV_T(3);
  if ((threadIdx.x & 15) == 0 && m < B) bce_part[(size_t)nt * B + m] = loss;
  MV_TFLUSH(20, 4, 96);
  MV_SPAN_END(2, 1);
}

template <bool FULL>
__global__ __launch_bounds__(512) void k_dec1_fwd(const float* hd, const float* W, const float* b, const float* x,
                                                  float* g, float* bce_part, float* logits_user, int B, int H, int D) {
  __shared__ float red[kW8][16][17];
  job_dec1_fwd_tile<FULL>(red, hd, W, b, x, g, bce_part, logits_user, B, H, D);
}

// Large components on the wave-cooperative kernels: the workgroups PAST the tiles compute the forward-mode dual records
// (job_duals_coop) -- dispatched after the tiles, which are the short job with a consumer in the next launch.
template <bool FULL>
__global__ __launch_bounds__(512) void k_dec1_fwd_duals(const float* hd, const float* W, const float* b, const float* x,
                                                        float* g, float* bce_part, float* logits_user, int B, int H, int D,
                                                        CompTable t, CoopDualArgs cd) {
  __shared__ float red[kW8][16][17];
  if ((int)blockIdx.x >= cd.n_tile_wg) {
    job_duals_coop(t, cd.heads, cd.ldh, cd.eps, cd.eps_ld, cd.radii, cd.duals, cd.NH, cd.DS, cd.ngroups,
                   (int)blockIdx.x - cd.n_tile_wg);
    return;
  }
  job_dec1_fwd_tile<FULL>(red, hd, W, b, x, g, bce_part, logits_user, B, H, D);
}

// ---- 4: dhd = (g W_logits) * [hd > 0] ; db_logits (+Adam) ; step statistics   (512 threads)
#ifndef MV_DHD_GROUP
#define MV_DHD_GROUP 2
#endif
constexpr int kDhdGroup = MV_DHD_GROUP;  // adjacent 16-column tiles of W_logits owned by one XCD (see xcd_tile_g)
// DUAL > 0 (fused step): the first n_dual workgroups compute the forward-mode dual records of the latent components
// (job_duals<DUAL>), one thread per (row, input direction).
// fragment-order copies produced inside launch 4: hd (by the dhd tile that reads it as its ReLU mask) and x (n_xf short jobs)
struct FragArgs {
  float* hdF;
  float* dhdF;
  const float* x;
  float* xF;
  int n_xf;  // > 0 only when launch 1's grid has no padding workgroups to do it
  // LITE 1 (the four-launch step): hd arrives in fragment order (hdF is an INPUT), dhd leaves in fragment order only, and every
  // dhd tile adds its share of dz = dhd W_d0 to the fixed-point sums `dzfix`.  LITE 2 (block backward): partial tiles in dzp
  float* dzp;
  const float* Wd0;
  int Z;
  // block backward: z's fragment-order copy for launch 6'' (n_zf = z tiles, one spare workgroup each)
  const float* z;
  float* zF;
  int ldz, n_zf;
  int stats_later;  // the statistics job runs in launch 5 (k_latent_bwd_blk, StatsArgs)
  // the four-launch step: dz = sum over the tiles as 64-bit fixed-point ATOMIC adds (order-independent, hence deterministic),
  // and the snapshot of W_heads for k_bwd56's dh product by n_snap short jobs of this launch (W_heads changes in k_bwd56)
  long long* dzfix;
  const float* Wh;
  float* whF;
  int n_snap, NH;
  int Bv;  // valid rows: z's fragment copy is zero past them (a padding row's z need not be finite)
};
struct DualArgs {
  const float* heads;
  const float* eps;
  const float* radii;
  float* duals;
  int ldh, eps_ld, NH, n_dual;
};
// LITE: 0 the generic launch; 1 the four-launch step (z_dim <= 8: hd read / dhd written in fragment order, the tile's share of
// dz from per-thread products and DPP row sums, added to dzfix); 2 the generic launch PLUS dz partials for the block backward
// (z_dim <= 64): the partial of a tile is itself an MFMA product, stored in the MFMA's output order
// statistics job of the step (one workgroup; launch 4, or launch 5 of the fragment-order block backward where launch 4's
// tiles end before it would).  sm: kW8 * 16 * 17 = 2176 floats of LDS: [0, 64) block sums / component sums, [64, ...) part sums
__device__ __forceinline__ void job_step_stats(float* sm, const float* bce_part, const float* kl, float* bce_user, float* stats,
                                               float beta, int B, int ntD, int ncomp) {
  // statistics block (BatchStats, stats.py:144-212): sums over the batch of bce, kl_i, elbo.
  // Thread (row r, part p) adds its quarter of the column tiles of the row's BCE partials with every load in flight at
  // once (one memory round trip instead of one per 8 tiles: this workgroup used to be the tail of the launch); the P
  // part sums of a row meet in LDS and are added in part order, so the result is deterministic.
  __shared__ float kl_s[4096];   // kl[i][r] of this step when ncomp * B fits: the per-component sums then read LDS
  __shared__ float old_s[8 + kMaxComp];  // the running sums this step is added to, requested with the first loads
  const int tid = threadIdx.x, nthr = blockDim.x;
  const bool kl_in_lds = (size_t)ncomp * B <= 4096;
  __shared__ float old_c[8 + kMaxComp];  // their Kahan compensation terms (third block of `stats`)
  if (tid < 4 + ncomp) {
    old_s[tid] = stats[tid];
    old_c[tid] = stats[2 * (4 + ncomp) + tid];
  }
  const int P = (4 * B <= nthr) ? 4 : ((2 * B <= nthr) ? 2 : 1);
  const int rows_pass = nthr / P;  // rows handled per pass
  const int chunk = (ntD + P - 1) / P;
  float bce_acc = 0.f, elbo_acc = 0.f;
  __shared__ float cs_s[kMaxComp];  // per-component batch sums of this step
  // (tried: 16-byte loads of four rows' partials per thread, 3 requests instead of 13, the group sums added from LDS:
  // 4.5 us against 4.2 us for this form)
  for (int r0 = 0; r0 < B; r0 += rows_pass) {
    const int rl = tid % rows_pass, p = tid / rows_pass;
    const int r = r0 + rl;
    const bool act = p < P && r < B;
    const int rr = act ? r : 0;
    float part = 0.f;
    for (int nt0 = p * chunk; nt0 < (p + 1) * chunk && nt0 < ntD; nt0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {  // pure requests (clamped); the masks come after the batch
        const int nt = nt0 + u;
        const bool ok = nt < (p + 1) * chunk && nt < ntD;
        v[u] = bce_part[(size_t)(ok ? nt : 0) * B + rr];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int nt = nt0 + u;
        part += (nt < (p + 1) * chunk && nt < ntD) ? v[u] : 0.f;
      }
    }
    // the row's KL terms: part p takes the components p, p + P, ... (one batch of requests instead of one round trip per
    // 8 components on a quarter of the threads); the P partial sums meet in LDS next to the BCE partials
    float klp = 0.f;
    if (act) {
      for (int i0 = p; i0 < ncomp; i0 += 8 * P) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * P;
          v[u] = kl[(size_t)(i < ncomp ? i : 0) * B + r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * P;
          if (i < ncomp) {
            klp += v[u];
            if (kl_in_lds) kl_s[(size_t)i * B + r] = v[u];
          }
        }
      }
    }
    if (P > 1) {
      if (p < P) {
        sm[64 + p * rows_pass + rl] = part;
        sm[64 + 1024 + p * rows_pass + rl] = klp;
      }
      __syncthreads();
    }
    if (p == 0 && act) {
      float bce = part, klr = klp;
      if (P == 2) {
        bce = sm[64 + rl] + sm[64 + rows_pass + rl];
        klr = sm[64 + 1024 + rl] + sm[64 + 1024 + rows_pass + rl];
      }
      if (P == 4) {
        bce = (sm[64 + rl] + sm[64 + rows_pass + rl]) + (sm[64 + 2 * rows_pass + rl] + sm[64 + 3 * rows_pass + rl]);
        klr = (sm[64 + 1024 + rl] + sm[64 + 1024 + rows_pass + rl]) +
              (sm[64 + 1024 + 2 * rows_pass + rl] + sm[64 + 1024 + 3 * rows_pass + rl]);
      }
      if (bce_user) bce_user[r] = bce;
      bce_acc += bce;
      elbo_acc += (-bce - beta * klr);
    }
    if (P > 1) __syncthreads();
  }
  // One phase, one barrier: every wave leaves its partial sums of bce / elbo in LDS and, round-robin, the batch sum of a
  // component's KL (rows in lane order); thread 0 then adds the wave partials in wave order (fixed order: deterministic).
  // (As two block-wide reductions followed by the component sums this tail cost six barriers, and the statistics
  // workgroup -- 4.6 us -- outlasted the dhd tiles of the launch.)
  const int last = 4 + ncomp;
  if (P == 1) __syncthreads();  // kl_s complete (the loop's own barriers cover P > 1)
  {
    const int wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;
    const float wb = wave_sum(bce_acc), we = wave_sum(elbo_acc);
    if (lane == 0) {
      sm[wave] = wb;
      sm[8 + wave] = we;
    }
    for (int i = wave; i < ncomp; i += nw) {
      float a = 0.f;
      if (kl_in_lds) {
        for (int r = lane; r < B; r += 64) a += kl_s[(size_t)i * B + r];
      } else {
        for (int r = lane; r < B; r += 64) a += kl[(size_t)i * B + r];
      }
      a = wave_sum(a);
      if (lane == 0) {
        kahan_add(stats, 4 + i, 2 * last, old_s[4 + i], old_c[4 + i], a);
        stats[last + 4 + i] = a;
        cs_s[i] = a;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    float bce_sum = 0.f, elbo_sum = 0.f, kl_total = 0.f;
    for (int w = 0; w < (nthr >> 6); ++w) {
      bce_sum += sm[w];
      elbo_sum += sm[8 + w];
    }
    for (int i = 0; i < ncomp; ++i) kl_total += cs_s[i];
    kahan_add(stats, 0, 2 * last, old_s[0], old_c[0], bce_sum);
    kahan_add(stats, 1, 2 * last, old_s[1], old_c[1], kl_total);
    kahan_add(stats, 2, 2 * last, old_s[2], old_c[2], elbo_sum);
    stats[3] = old_s[3] + 1.f;
    stats[last + 0] = bce_sum;
    stats[last + 1] = kl_total;
    stats[last + 2] = elbo_sum;
    stats[last + 3] = 1.f;
  }
}

template <bool ADAM, bool FULL, int DUAL, int LITE = 0>
__global__ __launch_bounds__(512) void k_dec1_bwd(CompTable t, const float* g, const float* hd, const float* W, float* db,
                                                  float* dhd, const float* bce_part, const float* kl, float* bce_user,
                                                  float* stats, float beta, int B, int H, int D, int ncomp, int n_dhd,
                                                  int n_db, AdamArgs ab, DualArgs da, FeedArgs fd, FragArgs fr) {
  __shared__ float red[kW8][16][17];
  // Workgroup order: the short jobs first (statistics, bias column sums, padded to a multiple of 8 so that the tile
  // workgroups keep L % 8 == XCD), then the dhd tiles.  The grid is larger than the chip: workgroups dispatched last
  // share a CU, which costs a tile workgroup little when the other one is short, but made the statistics workgroup the
  // tail of the launch when it came last.
  int b = blockIdx.x;

  const int ntH = (H + 15) / 16, ntD = (D + 15) / 16;
  const int n_dual = DUAL > 0 ? da.n_dual : 0;
  const int n_short = (n_dual + 1 + n_db + fd.n_wg + fr.n_xf + fr.n_zf + fr.n_snap + 7) & ~7;
  MV_SPAN_BEGIN(3);
  if (DUAL > 0 && b < n_dual) {  // the longest chains of the launch: dispatched first, ONE wave per workgroup (= per CU)
    if (threadIdx.x < 64)
      job_duals<(DUAL > 0 ? DUAL : 2)>(t, da.heads, da.ldh, da.eps, da.eps_ld, da.radii, da.duals, B, da.NH,
                                       b * 64 + (int)threadIdx.x);
    MV_SPAN_END(3, 4);
    return;
  }
  b -= n_dual;
  if (b >= n_short - n_dual) {  // n_dhd = xcd_grid(ntH, ntB, kDhdGroup): W_logits column blocks are dealt to XCDs
    b -= n_short - n_dual;
    int mt, nt;
    if (!xcd_tile_g(ntH, (B + 15) / 16, kDhdGroup, &nt, &mt, b)) return;
    const int wave = threadIdx.x >> 6;
    const int m = mt * 16 + ((threadIdx.x & 255) >> 4), n = nt * 16 + (threadIdx.x & 15);
    const bool ok = threadIdx.x < 256 && m < B && n < H;
    constexpr bool lite = LITE == 1;  // FULL shapes only
    // branch-free request, used in the epilogue
    const float mask = *((LITE && fr.hdF) ? fr.hdF + frag_off(m, n, B >> 4) : hd + (size_t)(m < B ? m : 0) * H + (n < H ? n : 0));
    f32x4 wz0 = {0.f, 0.f, 0.f, 0.f}, wz1 = wz0;  // lite: W_d0[n][0..7] (zero past Z), the thread's share of dz
    // LITE 2: wave w < ZT multiplies the masked dhd tile by W_d0[16 tile rows][16 z columns w]: B[k = 4 q + t][j = i]
    float wzt[4] = {0.f, 0.f, 0.f, 0.f};
    if (LITE == 2) {
      const int lane = threadIdx.x & 63, zc = 16 * wave + (lane & 15);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const float v = fr.Wd0[(size_t)(nt * 16 + 4 * (lane >> 4) + t4) * fr.Z + (zc < fr.Z ? zc : 0)];
        wzt[t4] = zc < fr.Z ? v : 0.f;
      }
    }
    if (lite) {
      const float* wr = fr.Wd0 + (size_t)n * fr.Z;
      if (fr.Z == 8) {
        wz0 = *reinterpret_cast<const f32x4*>(wr);
        wz1 = *reinterpret_cast<const f32x4*>(wr + 4);
      } else if (fr.Z == 4) {
        wz0 = *reinterpret_cast<const f32x4*>(wr);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = wr[j < fr.Z ? j : 0], b2 = wr[j + 4 < fr.Z ? j + 4 : 0];
          wz0[j] = j < fr.Z ? a : 0.f;
          wz1[j] = j + 4 < fr.Z ? b2 : 0.f;
        }
      }
    }
    const bool vg = aligned16(g) && (D & 3) == 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_nn<7, FULL>(g, D, B, mt * 16, W, H, H, nt * 16, D, wave, kW8, vg, acc);
    const float s = reduce_tiles8(red, acc);
    const float dv = (mask > 0.f) ? s : 0.f;
    if (lite) {
      if (threadIdx.x < 256) {
        fr.dhdF[frag_off(m, n, B >> 4)] = dv;  // launch 6: dW_d0 tiles and the b_d0 column sums
        // this tile's share of dz[m][j] = sum_n dhd[m][n] W_d0[n][j]: the 16 lanes of a row add up over n (DPP row sums)
        f32x4 p0 = dv * wz0, p1 = dv * wz1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          p0[j] = row16_sum(p0[j]);
          p1[j] = row16_sum(p1[j]);
        }
        {
          // lane j < Z of the row adds entry j to the row's fixed-point sum (integer adds commute: any arrival order gives the
          // same bits); a partial that is not finite or too large for the format marks the step instead (k_bwd56: dz = NaN)
          const int j = threadIdx.x & 15;
          float pv = p0[0];
          pv = j == 1 ? p0[1] : pv;
          pv = j == 2 ? p0[2] : pv;
          pv = j == 3 ? p0[3] : pv;
          pv = j == 4 ? p1[0] : pv;
          pv = j == 5 ? p1[1] : pv;
          pv = j == 6 ? p1[2] : pv;
          pv = j == 7 ? p1[3] : pv;
          if (j < fr.Z) {
            if (fabsf(pv) < kDzLimit) {
              const long long fx = (long long)rint((double)pv * kDzScale);
              atomicAdd(reinterpret_cast<unsigned long long*>(fr.dzfix) + (size_t)m * 8 + j, (unsigned long long)fx);
            } else {
              atomicOr(reinterpret_cast<unsigned*>(fr.dzfix + (size_t)B * 8), 1u);
            }
          }
        }
      }
      MV_SPAN_END(3, 1);
      return;
    }
    if (LITE == 2 && fr.dhdF) {  // fragment order only: launch 5 takes dz from the partial tiles, launch 6'' contracts fragments
      if (threadIdx.x < 256) fr.dhdF[frag_off(m, n, B >> 4)] = dv;
    } else if (ok) {
      dhd[(size_t)m * H + n] = dv;
    }
    if (LITE == 2) {
      __shared__ float dvs[16][17];
      if (threadIdx.x < 256) dvs[threadIdx.x >> 4][threadIdx.x & 15] = dv;
      lds_barrier();
      const int ZT = (fr.Z + 15) >> 4;
      if (wave < ZT) {  // uniform
        const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
        f32x4 d = {0.f, 0.f, 0.f, 0.f}, d2 = d;
        d = mfma16(dvs[i][4 * q + 0], wzt[0], d);
        d2 = mfma16(dvs[i][4 * q + 1], wzt[1], d2);
        d = mfma16(dvs[i][4 * q + 2], wzt[2], d);
        d2 = mfma16(dvs[i][4 * q + 3], wzt[3], d2);
        // lane (z column j = i of tile w, rows 4 q + r): one 16-byte store per lane, [row block][tile][z tile][64][4]
        reinterpret_cast<f32x4*>(fr.dzp)[((size_t)((mt * ntH + nt) * ZT + wave) << 6) + lane] = d + d2;
      }
    }
    MV_SPAN_END(3, 1);
    return;
  }
  if (b > n_db) {
    // Input pipeline (mvae_set_next_batch_feed): spare workgroups gather, binarise and draw the NEXT step's batch while the
    // dhd tiles run -- launch 1 of this step has already advanced the cursor, so counters[8] names that batch.  Nothing in
    // this step reads the buffers written here (the caller alternates between two).
    const int fb = b - n_db - 1;
    if (fb < fd.n_wg) {
      const unsigned cursor = (unsigned)fd.counters[8];
      const int n = feed_items(fd);
      for (int i = fb * (int)blockDim.x + (int)threadIdx.x; i < n; i += fd.n_wg * (int)blockDim.x) feed_item(fd, cursor, i);
    } else if (fb - fd.n_wg < fr.n_xf) {
      job_frag_copy(fr.x, D, fb - fd.n_wg, B >> 4, fr.xF);  // x in fragment order for launch 6's dW_e0 tiles
    } else if (fb - fd.n_wg - fr.n_xf < fr.n_zf) {
      job_frag_copy(fr.z, fr.ldz, fb - fd.n_wg - fr.n_xf, B >> 4, fr.zF, fr.Z, fr.Bv);
    } else if (fb - fd.n_wg - fr.n_xf - fr.n_zf < fr.n_snap) {
      // whF[(pt * 64 + q * 16 + i) * 4 + t] = W_heads[4 q + t][16 pt + i] (0 past NH): the B fragments of k_bwd56's dh product
      const int sb = fb - fd.n_wg - fr.n_xf - fr.n_zf;
      for (int e = sb * (int)blockDim.x + (int)threadIdx.x; e < (H >> 4) * 256; e += fr.n_snap * (int)blockDim.x) {
        const int pt = e >> 8, ln = (e & 255) >> 2, n = 4 * (ln >> 4) + (e & 3);
        const float v = fr.Wh[(size_t)(n < fr.NH ? n : 0) * H + pt * 16 + (ln & 15)];
        fr.whF[e] = n < fr.NH ? v : 0.f;
      }
    }
    return;  // (the rest: padding)
  }
  if (b >= 1) {
    b -= 1;
    job_colsum_opt<ADAM>(&red[0][0][0], g, D, B, D, b * kColsPerBlock, db, ab);
    MV_SPAN_END(3, 2);
    return;
  }
  if (fr.stats_later) return;  // (block backward, fragment-order form: the job runs in launch 5)
  job_step_stats(&red[0][0][0], bce_part, kl, bce_user, stats, beta, B, ntD, ncomp);
  MV_SPAN_END(3, 3);
}

// ---- 5: backward through the first decoder layer, the latent components and the heads (one batch row per
// workgroup) ; dW_logits = g^T hd (+Adam: W_logits was last read by launch 4)
template <int DMAX, bool FAST, bool ADAM>  // FAST also implies tile-aligned B, H, D (checked on the host)
__global__ __launch_bounds__(64 * kTileWaves5) void k_latent_bwd(CompTable t, const float* dhd, const float* Wd0, int ldh,
                                                    const float* h, const float* Wh, float* dheads, float* dh,
                                                    float* drpart, const float* g, const float* hd, float* dWl,
                                                    float beta, int B, int H, int D, int NH, int Z, int n_rows,
                                                    AdamArgs awl, const float* duals, const float* dzp, float* dheadsF,
                                                    float* dhF) {
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
#ifdef MV_TILE_SLEEP
    // let the row workgroups' requests (the critical path of this launch) enter the memory system first: the tile
    // workgroups move ~16 MB and would otherwise queue ahead of them
    __builtin_amdgcn_s_sleep(MV_TILE_SLEEP);
#endif
    const int ntHg = ((H + 15) / 16 + kTileWaves5 - 1) / kTileWaves5;
    const int pt = fast_div(b, ntHg), qg = b - pt * ntHg;
    // (tried: XCD parity classes of the g column blocks as in launch 6, and g-blocks-fastest order: 5.95 / 5.78 us against
    // 5.77 us in this plain order)
    if (FAST) job_tn_wave<ADAM, true>(g, D, D, pt, hd, H, H, qg * kTileWaves5 + wave, B, dWl, H, awl);
    else job_tn_wave<ADAM, false>(g, D, D, pt, hd, H, H, qg * kTileWaves5 + wave, B, dWl, H, awl);
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
  // FAST with z_dim 8 or 4 (every BASELINE MLP config but e6): the two small weight matrices are requested as 16-byte
  // vectors -- W_d0 [H][Z] flat (H Z / 1024 requests per thread), W_heads as four consecutive columns per thread (one
  // request per head row, on the first H/4 threads) -- ~70 wave-level requests per workgroup instead of ~220 (4-byte
  // ones): at ~36 cycles per request on the CU's load path that was ~3 us of this kernel's 5.
  const bool zv = FAST && (Z == 8 || Z == 4);  // uniform
  f32x4 wz4[4], wh4[16];
  float4 hm4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (zv) {
    const int nz4 = (H * Z) >> 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e4 = tid + 256 * u;
      wz4[u] = *reinterpret_cast<const f32x4*>(Wd0 + 4 * (size_t)(e4 < nz4 ? e4 : 0));
    }
    const unsigned c4 = 4u * (unsigned)(tid < (H >> 2) ? tid : 0);
#pragma unroll
    for (int n = 0; n < 16; ++n) wh4[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((wave << 6) < (H >> 2)) {  // wave-uniform: only the waves that own columns of dh request W_heads (and h)
      hm4 = *reinterpret_cast<const float4*>(h + rowH + c4);
#pragma unroll
      for (int n = 0; n < 16; ++n)
        if (n < NH) wh4[n] = *reinterpret_cast<const f32x4*>(Wh + (unsigned)n * (unsigned)H + c4);  // uniform
    }
  } else
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
  } else if (!dzp) {
    for (int k = tid; k < H; k += 256) dhd_s[k] = dhd[row * H + k];
  }
  // generic 16-byte path of dz: thread (slice s2, column quad j4) owns rows c = s2, s2 + nslv, ... of W_d0; the
  // first 24 of them are requested here, ahead of the barrier (one round trip instead of one per 4 rows)
  constexpr int kDzB = 24;
  // dzp (generic shapes on tile-aligned sizes: the large components): dz arrives as the partial products of launch 4's dhd
  // tiles -- [row block][tile][z tile][64][4], as for the block backward -- and the row adds its 25 partials in tile order
  // instead of contracting its dhd row with the whole of W_d0 (H x Z floats per ROW: 65 KB of 4-byte requests for `h40`)
  const bool use_dzp = !FAST && dzp != nullptr;
  const bool dz_vec = !FAST && !use_dzp && (Z & 3) == 0 && aligned16(Wd0);
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
  if (use_dzp) {
    const int ntH = (H + 15) >> 4, ZT = (Z + 15) >> 4;
    const int mt = (int)(row >> 4), rr = (int)(row & 15);
    for (int j = tid; j < Z; j += 256) {
      const float* src = dzp + ((((size_t)mt * ntH) * ZT + (j >> 4)) * 64 + (rr >> 2) * 16 + (j & 15)) * 4 + (rr & 3);
      const size_t stride = (size_t)ZT * 256;  // floats from one tile's record to the next tile's
      float tot = 0.f;
      int nt = 0;
      for (; nt + 8 <= ntH; nt += 8) {  // eight requests in flight, added in tile order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(nt + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += v[u];
      }
      for (; nt < ntH; ++nt) tot += src[(size_t)nt * stride];
      dz_s[j] = tot;
    }
    lds_barrier();
  } else {
    float p = 0.f;
    if (zv) {
      // thread's flat vectors e4 = tid + 256 u: row c = e4 / (Z/4), columns 4 (e4 % (Z/4)) .. +3; the column half is the
      // parity of the lane (Z == 8) or none (Z == 4)
      const int nz4 = (H * Z) >> 2, sh = Z == 8 ? 1 : 0;
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e4 = tid + 256 * u;
        const float dv = e4 < nz4 ? dhd_s[(e4 < nz4 ? e4 : 0) >> sh] : 0.f;
        a4 += dv * wz4[u];
      }
      // lanes of equal column half: butterfly over the remaining lane bits, then the four waves meet in LDS
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        if (off >= (1 << sh)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) a4[k] += __shfl_xor(a4[k], off);
        }
      if (lane < (1 << sh)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) part[wave * ZP + lane * 4 + k] = a4[k];
      }
      lds_barrier();
      if (tid < Z) dz_s[tid] = (part[tid] + part[ZP + tid]) + (part[2 * ZP + tid] + part[3 * ZP + tid]);
    } else
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
    if (zv) {
    } else if (FAST) {
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
  if (dheadsF) {  // (uniform; with dzp) fragment order, whole 16-column tiles, zero past NH: launch 6 is k_enc_bwd3
    for (int c = tid; c < ((NH + 15) & ~15); c += 256) dheadsF[frag_off((int)row, c, B >> 4)] = c < NH ? dheads_s[c] : 0.f;
  }
  // ---- dh = (dheads W_heads) * [h > 0]   (K = NH is small)
  if (zv) {
    if (tid < (H >> 2)) {
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < 16; ++n) a4 += dheads_s[n] * wh4[n];  // entries [NH, 16) of dheads_s are zero
      float4 o;
      o.x = hm4.x > 0.f ? a4[0] : 0.f;
      o.y = hm4.y > 0.f ? a4[1] : 0.f;
      o.z = hm4.z > 0.f ? a4[2] : 0.f;
      o.w = hm4.w > 0.f ? a4[3] : 0.f;
      *reinterpret_cast<float4*>(dh + row * H + 4 * tid) = o;
    }
  } else if (FAST) {
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
      if (dhF) {  // (uniform) fragment order only
        const size_t o = frag_off((int)row, 4 * tid, B >> 4);  // columns 4 tid .. + 3 lie in one tile: 4 floats apart
        dhF[o] = tot.x;
        dhF[o + 4] = tot.y;
        dhF[o + 8] = tot.z;
        dhF[o + 12] = tot.w;
      } else {
        *reinterpret_cast<float4*>(dh + row * H + 4 * tid) = tot;
      }
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

// radius gradients: sum over the batch rows of the per-row terms of launch 5 (fixed order: deterministic), and in the
// fused step torch.optim.SGD(lr=curv_lr) on the trainable radii: param.add_(grad, alpha=-lr).  One workgroup; gsh: LDS,
// kRadiiRegion floats.
template <bool ADAM>
__device__ __forceinline__ void job_radii(const CompTable& t, float* gsh, const float* drpart, float* G, float* P, int B,
                                          double curv_lr, int do_curv) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < kRadiiRegion) {
    G[tid] = 0.f;
    gsh[tid] = 0.f;
  }
  __syncthreads();
  const int nw = (int)(blockDim.x >> 6);
  if (B <= 256) {  // four components per wave and round, every request of the round in flight at once (rows in the same order)
    for (int c0 = wave; c0 < t.n; c0 += 4 * nw) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ci = c0 + u * nw;
        const bool on = ci < t.n && t.trainable[ci];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = lane + 64 * k;
          v[u][k] = on && r < B ? drpart[(size_t)ci * B + r] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ci = c0 + u * nw;
        const float s = wave_sum(((v[u][0] + v[u][1]) + v[u][2]) + v[u][3]);
        if (lane == 0 && ci < t.n && t.trainable[ci]) gsh[ci] = s;
      }
    }
  } else {
    for (int ci = wave; ci < t.n; ci += nw) {
      if (!t.trainable[ci]) continue;
      float s = 0.f;
      for (int r = lane; r < B; r += 64) s += drpart[(size_t)ci * B + r];
      s = wave_sum(s);
      if (lane == 0) gsh[ci] = s;
    }
  }
  __syncthreads();
  if (tid < t.n && t.trainable[tid]) {
    float s = gsh[tid];
    if (ADAM && (t.trainable[tid] & 2)) s *= clip_coef(t, gsh);  // vae.py:161-163 (fused step; else k_optim clips)
    G[tid] = s;
    if (ADAM && do_curv) P[tid] = P[tid] + (float)(-curv_lr) * s;
  }
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
    job_radii<ADAM>(t, &red[0][0][0], drpart, G, P, B, curv_lr, do_curv);
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
    // (tried: the five waves splitting the contraction of the single 16-column tile -- the job itself 4.6 -> 3.5 us, the
    // launch 5.45 -> 5.65 us; not kept)
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
    // (tried: XCD parity classes of the x column groups, and dh-blocks-fastest order: no faster than this plain order)
    const int pt = fast_div(b, ntDg), qg = b - pt * ntDg;
    job_tn_wave<ADAM, FULL>(dh, H, H, pt, x, D, D, qg * kTileWaves + (threadIdx.x >> 6), B, G + off_w_e0, D,
                            at(off_w_e0));
    MV_SPAN_END(5, 1);
  }
}


// ===================================================== the backward of the fused-forward shapes: launch 4 + ONE more (k_bwd56)
// Models on the fused forward (NH <= 16, Z <= 8, B <= 256: BASELINE configs [0], [1], [2]) run the step in FOUR launches:
// k_enc_fwd, k_fwd23, k_dec1_bwd<LITE 1>, k_bwd56.  What the generic launches 5 and 6 (k_latent_bwd, k_enc_bwd) do per batch
// row -- load a row of dhd and W_d0 for dz, load W_heads and the row of h for dh = (dheads W_heads)[h > 0] -- and the launch
// boundary between them (a launch costs ~4 us whatever it does) are replaced by:
//   * dz arrives SUMMED: every dhd tile of launch 4 adds its share of dz = dhd W_d0 into 64-bit fixed-point accumulators with
//     atomic adds (integer adds commute: the sum has the same bits in any arrival order; the forward launch zeroes them);
//   * every workgroup of k_bwd56 rebuilds the dheads rows it needs from dz and the dual records -- which the forward launch
//     stores per HEAD COLUMN in the order these lanes read them (Rec4Args): lane (i, q) of the wave that owns row block c
//     computes dheads[16 c + i][4 q .. 4 q + 3], which IS the A fragment of the dh product -- 4 NV + 1 wave-level requests per
//     row block, no cross-wave dependency in front of the dh MFMAs;
//   * dh is never written: each dW_e0 workgroup rebuilds the fragments of dh it contracts with -- the MFMA D = dheads[16 rows][16]
//     W_heads[16][16 columns], whose output lane layout IS the B-operand fragment of the weight-gradient MFMA -- masks them
//     with h's fragment-order copy and shares them through LDS (their column sums are b_e0's gradient); W_heads is read from
//     the snapshot short jobs of launch 4 write (this launch's dW_heads tiles update W_heads in place);
//   * every batch contraction reads fragment-order operands (mvae_common.hpp: frag_off); only g is read row-major (a second,
//     fragment-order copy of g cost the forward launch more than the tiles gained: 29.8 against 29.2 us per step);
//   * the dW_logits tiles (+ Adam; W_logits was last read by launch 4) ride on FIVE MORE WAVES of the same workgroups.  They
//     issue their requests only after the five main waves have issued theirs (an LDS counter; a CU's load path serves
//     requests in issue order and the main waves carry the dependent chain), and the main waves meet on an LDS counter, not on
//     s_barrier, which would also wait for the tile waves.
// Workgroup 0: radii (+ SGD, clip) and b_heads; workgroups 1 .. n_small: dW_d0 (+ b_d0) tiles (and dW_heads where the rows of
// tiles have no idle wave); the rest: 16 columns of dh x 5 x tiles each, dW_heads on the idle wave of the last tile group of
// a row.  1 + 5 + 250 workgroups of ten waves for the BASELINE shapes: one per CU.
// Measured (interleaved 2000-step runs, two boxes): 29.0-29.2 us per step against 31.0-31.2 with launches 5' + 6' (rounds 5's
// k_latent_bwd2 / k_enc_bwd2, removed): launches 4.7 / 9.4 / 4.9 / 8.5 us against 4.7 / 9.4 / 4.8 / 4.75 + 5.2.
constexpr int kW56 = 2 * kTileWaves;
struct L56Args {
  const long long* dzfix;  // [B][8] fixed-point dz, then the overflow mark
  const float* recH;       // head-direction records (Rec4Args)
  const float* recR;       // radius-direction records
  const float* g;          // [B][D] row-major
  const float* hdF;        // hd in fragment order
  float* dheads;           // [B][ldh] (for observers; written by workgroup 0)
  float* drpart;           // [ncomp][B]     "
  int ldh;
  float beta;
  int64_t off_w_logits;
  int Bv;                  // valid rows: dheads / radius terms of the padding rows are zero
};

// the five main waves of a k_bwd56 workgroup meet here (s_barrier would wait for the tile waves as well)
__device__ __forceinline__ void group_sync(int* ctr, int target) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// dheads of the row blocks c = wave, wave + 5, ... of this wave (waves 0 .. 4): da[u] = dheads[16 c + i][4 q .. 4 q + 3] of lane
// (i, q); also left in dh_s (every row of the batch once the five waves have met) and dz_s (dz as floats).  The blocks are
// taken CH at a time (request(u0) ... finish(u0)): a pass holds CH x 4 NV record vectors per lane, and a ten-wave workgroup
// has 168 registers -- one pass for the BASELINE shapes (B = 128, NV <= 2), more for B = 256 or records of three vectors.
template <int NV, int MBT>
struct DheadsJob {
  static constexpr int kPer = (MBT + kTileWaves - 1) / kTileWaves;
  static constexpr int CH = NV >= 3 ? 1 : (kPer < 2 ? kPer : 2);
  static constexpr int AM = 4 * NV - 1;
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  i64x2 zq[CH];
  f32x4 rv[CH][4][NV];
  unsigned mark;
  int zc[4], Aa[4];
  __device__ __forceinline__ void request(const L56Args& a, int MB, int B, int wave, int lane, int u0 = 0,
                                          const unsigned* mark_p = nullptr) {
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = wave + kTileWaves * (u0 + u);
      const int cc = c < MB ? c : 0;
      zq[u] = *reinterpret_cast<const i64x2*>(a.dzfix + ((size_t)(16 * cc + i) << 3) + (q << 1));
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int v = 0; v < NV; ++v)
          rv[u][tt][v] = reinterpret_cast<const f32x4*>(a.recH)[((size_t)((cc * 4 + tt) * NV + v) << 6) + lane];
    }
    if (u0 == 0) mark = mark_p ? *mark_p : *reinterpret_cast<const unsigned*>(a.dzfix + (size_t)B * 8);
  }
  // this lane's four head columns 4 q + tt: z column and ambient dimension of the component that owns each (A = 0: none)
  __device__ __forceinline__ void tables(const CompTable& t, int lane) {
    const int q = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) zc[tt] = Aa[tt] = 0;
    for (int ci = 0; ci < t.n; ++ci) {  // uniform loop, per-lane selects
      const mvae_component_desc c = t.c[ci];
      const int A = ambient_dim(c.kind, c.true_dim);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int col = 4 * q + tt;
        const bool in = (col >= c.mean_col && col < c.mean_col + c.true_dim) ||
                        (col >= c.logvar_col && col < c.logvar_col + c.logvar_dim);
        zc[tt] = in ? c.z_col : zc[tt];
        Aa[tt] = in ? A : Aa[tt];
      }
    }
  }
  __device__ __forceinline__ void finish(const L56Args& a, int MB, int NH, int wave, int lane, f32x4 (&da)[kPer],
                                         float (*dh_s)[16], float (*dz_s)[8], int u0 = 0) {
    const int i = lane & 15, q = lane >> 4;
    const float nanv = __int_as_float(0x7fc00000);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (u0 + u >= kPer) break;
      const int c = wave + kTileWaves * (u0 + u);
      if (c < MB) {  // uniform
        const int row = 16 * c + i;
        float d0 = (float)((double)zq[u][0] * (1.0 / kDzScale)), d1 = (float)((double)zq[u][1] * (1.0 / kDzScale));
        d0 = mark ? nanv : d0;
        d1 = mark ? nanv : d1;
        typedef float f32x2l __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2l*>(&dz_s[row][2 * q]) = f32x2l{d0, d1};
        // the four q lanes of a row exchange through LDS: one wave, program order, no barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        f32x4 gv;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          float acc = a.beta * rv[u][tt][0][0];
#pragma unroll
          for (int k = 0; k < AM; ++k) {
            const int zi = zc[tt] + k;
            const float dzv = dz_s[row][zi < 8 ? zi : 7];
            const float rk = rv[u][tt][(1 + k) >> 2][(1 + k) & 3];
            acc += k < Aa[tt] ? dzv * rk : 0.f;
          }
          gv[tt] = (Aa[tt] > 0 && 4 * q + tt < NH && row < a.Bv) ? acc : 0.f;  // (zero for padding rows)
        }
        da[u0 + u] = gv;
        *reinterpret_cast<f32x4*>(&dh_s[row][4 * q]) = gv;
      } else {
        da[u0 + u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  // the passes after the first (their records are requested only now: a memory round trip each)
  __device__ __forceinline__ void rest(const L56Args& a, int MB, int B, int NH, int wave, int lane, f32x4 (&da)[kPer],
                                       float (*dh_s)[16], float (*dz_s)[8]) {
#pragma unroll
    for (int u0 = CH; u0 < kPer; u0 += CH) {
      __builtin_amdgcn_sched_barrier(0);
      request(a, MB, B, wave, lane, u0);
      __builtin_amdgcn_sched_barrier(0);
      finish(a, MB, NH, wave, lane, da, dh_s, dz_s, u0);
    }
  }
};

template <int NV, bool ADAM, int MBT>
__global__ __launch_bounds__(64 * kW56) void k_bwd56(CompTable t, L56Args a, const float* __restrict__ scal_p,
                                                     const unsigned* __restrict__ mark_p, const float* xF, const float* hF, const float* whF,
                                                     const float* dhdF, const float* zF, float* G, float* P, int B, int H,
                                                     int D, int NH, int Z, int n_small, int heads_on_idle, int64_t off_w_e0,
                                                     int64_t off_b_e0, int64_t off_w_heads, int64_t off_b_heads,
                                                     int64_t off_w_d0, int64_t off_b_d0, AdamArgs base, double curv_lr,
                                                     int do_curv) {
  // scal_p = {-lr/bc1, sqrt(bc2)} of this step (launch 1; counters[2..3]) and mark_p = the dz overflow mark: uniform values
  // behind __restrict__ pointers, so that they are SCALAR loads -- as vector loads they were 3 of a wave's ~25 requests
  __shared__ float red[4][16][17];
  __shared__ f32x4 frag_s[MBT][64];  // the masked dh fragments of this workgroup's 16 columns, all row blocks
  __shared__ __attribute__((aligned(16))) float dh_s[MBT * 16][16];  // dheads of every batch row (zero past NH)
  __shared__ __attribute__((aligned(16))) float dz_s[MBT * 16][8];   // dz of every batch row
  __shared__ int meet_s, gate_s;
  const float scal[2] = {ADAM ? scal_p[0] : 0.f, ADAM ? scal_p[1] : 1.f};
  const int b = blockIdx.x;
  const int MB = B >> 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  MV_SPAN_BEGIN(5);
  auto at = [&](int64_t off) {
    AdamArgs r = base;
    r.p += off;
    r.m += off;
    r.v += off;
    return r;
  };
  constexpr int kPer = DheadsJob<NV, MBT>::kPer;
  if (b == 0) {
    // ---- radii (+ SGD, clip) and b_heads: all ten waves.  Waves 0 .. 4 rebuild dheads / dz for every row; meanwhile every
    // thread requests its radius-direction records, item = ci * B + row
    __shared__ float drp_s[kRecRad * MBT * 16];
    constexpr int kRI = (kRecRad * MBT * 16 + 64 * kW56 - 1) / (64 * kW56);
    DheadsJob<NV, MBT> dj;
    f32x4 da[kPer];
    if (wave < kTileWaves) dj.request(a, MB, B, wave, lane, 0, mark_p);
    f32x4 rr[kRI][NV];
    const int n_items = t.n * B;
#pragma unroll
    for (int k = 0; k < kRI; ++k) {
      const int it = (int)threadIdx.x + 64 * kW56 * k;
      const int itc = it < n_items ? it : 0;
      const int ci = itc / B, row = itc - ci * B;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        rr[k][v] = reinterpret_cast<const f32x4*>(a.recR)[((size_t)row * kRecRad + ci) * NV + v];
    }
    float bp = 0.f, bm = 0.f, bv2 = 0.f, neg_step = 0.f, bc2s = 1.f;
    if (ADAM) {
      const int col = (int)threadIdx.x < NH ? (int)threadIdx.x : 0;
      bp = base.p[off_b_heads + col];
      bm = base.m[off_b_heads + col];
      bv2 = base.v[off_b_heads + col];
      neg_step = scal[0];
      bc2s = scal[1];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wave < kTileWaves) {
      dj.tables(t, lane);
      dj.finish(a, MB, NH, wave, lane, da, dh_s, dz_s);
      dj.rest(a, MB, B, NH, wave, lane, da, dh_s, dz_s);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRI; ++k) {
      const int it = (int)threadIdx.x + 64 * kW56 * k;
      if (it < n_items) {
        const int ci = it / B, row = it - ci * B;
        int zc = 0, A = 0, on = 0;
        for (int cj = 0; cj < t.n; ++cj) {  // uniform loop, per-lane selects
          const mvae_component_desc c = t.c[cj];
          const bool me = cj == ci;
          zc = me ? c.z_col : zc;
          A = me ? ambient_dim(c.kind, c.true_dim) : A;
          on = me ? (int)t.trainable[cj] : on;
        }
        float acc = a.beta * rr[k][0][0];
#pragma unroll
        for (int kk = 0; kk < 4 * NV - 1; ++kk) {
          const int zi = zc + kk;
          const float dzv = dz_s[row][zi < 8 ? zi : 7];
          acc += kk < A ? dzv * rr[k][(1 + kk) >> 2][(1 + kk) & 3] : 0.f;
        }
        acc = (on && row < a.Bv) ? acc : 0.f;  // (the record of a fixed radius is not written; padding rows: no term)
        drp_s[it] = acc;
        a.drpart[it] = acc;
      }
    }
    for (int e = threadIdx.x; e < B * 16; e += 64 * kW56)
      if ((e & 15) < NH) a.dheads[(size_t)(e >> 4) * a.ldh + (e & 15)] = dh_s[e >> 4][e & 15];
    __syncthreads();
    job_radii<ADAM>(t, &red[0][0][0], drp_s, G, P, B, curv_lr, do_curv);
    __syncthreads();
    // b_heads[col] = sum over the rows of dheads[:, col]: 16 row groups, then the groups in index order
    if (threadIdx.x < 256) {
      const int c = threadIdx.x & 15, gq = threadIdx.x >> 4;
      float s = 0.f;
      for (int m = gq; m < B; m += 16) s += dh_s[m][c];
      red[0][gq][c] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NH) {
      float tsum = 0.f;
      for (int gq = 0; gq < 16; ++gq) tsum += red[0][gq][threadIdx.x];
      G[off_b_heads + threadIdx.x] = tsum;
      if (ADAM) {
        adam1(bp, tsum, bm, bv2, neg_step, bc2s);
        base.p[off_b_heads + threadIdx.x] = bp;
        base.m[off_b_heads + threadIdx.x] = bm;
        base.v[off_b_heads + threadIdx.x] = bv2;
      }
    }
    MV_SPAN_END(5, 7);
    return;
  }
  if (threadIdx.x == 0) meet_s = gate_s = 0;
  __syncthreads();  // (the only hardware barrier of these workgroups: all ten waves are at their first instructions)
  auto open_gate = [&]() {  // a main wave has issued its requests
    if (lane == 0) __hip_atomic_fetch_add(&gate_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (wave >= kTileWaves) {
    // ---- dW_logits[D, H] tiles (+ Adam), one per wave: tile index over (workgroup - 1, wave - 5)
    const int ntH = H >> 4;
    const int tw = (b - 1) * kTileWaves + (wave - kTileWaves), pt = fast_div(tw, ntH), qt = tw - pt * ntH;
    if (pt * 16 < D) {
      // The main waves' requests first: a CU's load path serves ~1 wave-level request per 36 cycles in issue order, and these
      // tiles have slack -- issued from the first cycle they ended at 4.8 us of a 7.4 us launch while the main waves' operands
      // queued behind their ~200 requests
      // (measured: 29.2 against 29.5 us per step without the wait)
      while (__hip_atomic_load(&gate_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < kTileWaves) __builtin_amdgcn_s_sleep(4);
      const AdamArgs awl = at(a.off_w_logits);
      // P = g read row-major in the fragments' row order, Q = hd in fragment order: 40 wave-level requests per tile
      job_tn_halffrag<ADAM>(a.g, D, pt, D, a.hdF, qt, H, MB, G + a.off_w_logits, H, awl, scal);
    }
    MV_SPAN_END_T(5, 2, 64 * kTileWaves, 1024);
    return;
  }
  const int bs = b - 1;
  if (bs < n_small) {
    // wave w, column tile tl = 5 bs + w of H: the dW_d0 tile [16 rows tl, Z] = dhd^T z with b_d0's 16 entries (the column sums
    // of its dhd fragments); without idle tile waves (D / 16 a multiple of 5) also the dW_heads tile [NH, 16 columns tl]
    const int tl = bs * kTileWaves + wave;
    const bool live = tl * 16 < H;  // uniform
    open_gate();
    if (live)
      job_tn_frag_any<ADAM, true>(dhdF, nullptr, 0, tl, H, zF, 0, Z, MB, G + off_w_d0, Z, at(off_w_d0), G + off_b_d0,
                                  at(off_b_d0), scal);
    if (!heads_on_idle) {  // (uniform; one job after the other: a rare shape, and both at once do not fit the registers)
      DheadsJob<NV, MBT> dj;
      f32x4 da[kPer];
      f32x4 hv[MBT];
      f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, m0 = p0, v0 = p0;
      float neg_step = 0.f, bc2s = 1.f;
      const bool okh = i < NH && live;
      const size_t idxh = (size_t)(i < NH ? i : 0) * H + (live ? tl : 0) * 16 + (q << 2);
      dj.request(a, MB, B, wave, lane, 0, mark_p);
#pragma unroll
      for (int c = 0; c < MBT; ++c) hv[c] = reinterpret_cast<const f32x4*>(hF)[((size_t)((live ? tl : 0) * MB + (c < MB ? c : 0)) << 6) + lane];
      __builtin_amdgcn_sched_barrier(0);
      dj.tables(t, lane);
      dj.finish(a, MB, NH, wave, lane, da, dh_s, dz_s);
      dj.rest(a, MB, B, NH, wave, lane, da, dh_s, dz_s);
      if (ADAM) {  // (requested late: together with the records they do not fit the registers)
        p0 = *reinterpret_cast<const f32x4*>(base.p + off_w_heads + idxh);
        m0 = *reinterpret_cast<const f32x4*>(base.m + off_w_heads + idxh);
        v0 = *reinterpret_cast<const f32x4*>(base.v + off_w_heads + idxh);
        neg_step = scal[0];
        bc2s = scal[1];
      }
      __builtin_amdgcn_sched_barrier(0);
      group_sync(&meet_s, kTileWaves);
      if (live) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int c = 0; c < MBT; ++c)
          if (c < MB) {  // uniform
            const int r0 = 16 * c + 4 * q;
            acc = mfma16(hv[c][0], dh_s[r0 + 0][i], acc);
            acc2 = mfma16(hv[c][1], dh_s[r0 + 1][i], acc2);
            acc = mfma16(hv[c][2], dh_s[r0 + 2][i], acc);
            acc2 = mfma16(hv[c][3], dh_s[r0 + 3][i], acc2);
          }
        acc += acc2;
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
        if (okh) {
          store16_wt(G + off_w_heads, idxh, acc);
          if (ADAM) {
            store16_wt(base.p + off_w_heads, idxh, p0);
            store16_wt(base.m + off_w_heads, idxh, m0);
            store16_wt(base.v + off_w_heads, idxh, v0);
          }
        }
      }
    }
    MV_SPAN_END(5, 3);
    return;
  }
  {  // dW_e0[H, D] = dh^T x ; b_e0 = column sums of dh ; on the idle wave of a tile row: its dW_heads tile
    const int bt = bs - n_small;
    const int ntDg = ((D >> 4) + kTileWaves - 1) / kTileWaves;
    const int pt = fast_div(bt, ntDg), qg = bt - pt * ntDg;
    const int qt = qg * kTileWaves + wave;
    const bool have = qt * 16 < D;  // wave-uniform
    const bool idle_wave = heads_on_idle && !have && qt == (D >> 4);
    const bool bias_wave = qg == 0 && wave == 0;
    // ---- requests.  First what the dheads rows of this wave's row blocks are made of (the head of the dependent chain), then
    // the operands of the dh product (B = W_heads[4 q + t][p0 + i] from launch 4's snapshot: this launch's dW_heads tiles
    // update W_heads in place; mask = h's fragment), then the tile's own
    DheadsJob<NV, MBT> dj;
    MV_STAMP_B(40, MV_STAMP_BLK);
    dj.request(a, MB, B, wave, lane, 0, mark_p);
    f32x4 da[kPer], hm[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int c = wave + kTileWaves * u;
      const int cc = c < MB ? c : 0;
      hm[u] = reinterpret_cast<const f32x4*>(hF)[((size_t)(pt * MB + cc) << 6) + lane];
    }
    const f32x4 wb = reinterpret_cast<const f32x4*>(whF)[((size_t)pt << 6) + lane];
    // Q operand and optimizer state, branch-free: tile wave = x's fragments of column tile qt and W_e0's tile (pt, qt);
    // idle wave = h's fragments of column tile pt and W_heads' tile (0, pt)
    const bool ok = idle_wave ? i < NH : have;
    const size_t idx = idle_wave ? (size_t)(i < NH ? i : 0) * H + pt * 16 + (q << 2)
                                 : (size_t)(pt * 16 + i) * D + (have ? qt : 0) * 16 + (q << 2);
    const int64_t woff = idle_wave ? off_w_heads : off_w_e0;
    const f32x4* qa = reinterpret_cast<const f32x4*>(idle_wave ? hF : xF) + ((size_t)(idle_wave ? pt : (have ? qt : 0)) * MB << 6) + lane;
    // Wide records (NV >= 2) or sixteen row blocks: the records and the tile's Q fragments do not fit the 168 registers of a
    // ten-wave workgroup together -- the tile's own operands are then requested once the dheads rows are done (their round
    // trip overlaps the dh product and the meeting instead of the records' one)
    constexpr bool kLateQ = NV >= 2 || MBT > 8;
    f32x4 av[MBT];
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, m0 = p0, v0 = p0;
    float neg_step = 0.f, bc2s = 1.f;
    const int bcol = pt * 16 + i;
    float bp = 0.f, bm = 0.f, bvv = 0.f;
    auto request_tile = [&]() {
#pragma unroll
      for (int c = 0; c < MBT; ++c) av[c] = qa[(size_t)(c < MB ? c : 0) << 6];
      if (ADAM) {
        p0 = *reinterpret_cast<const f32x4*>(base.p + woff + idx);
        m0 = *reinterpret_cast<const f32x4*>(base.m + woff + idx);
        v0 = *reinterpret_cast<const f32x4*>(base.v + woff + idx);
        neg_step = scal[0];
        bc2s = scal[1];
        // (the bias column: requested by every wave of the FIRST tile group of a row of tiles -- under `if (bias_wave)` the
        // compiler ends the block with a wait for ALL outstanding requests, a full memory round trip in front of the
        // fragment phase the five waves meet on; the other nine tile groups of the row do not request it at all)
        if (qg == 0) {  // (uniform per workgroup)
          bp = base.p[off_b_e0 + bcol];
          bm = base.m[off_b_e0 + bcol];
          bvv = base.v[off_b_e0 + bcol];
        }
      }
    };
    if (!kLateQ) request_tile();
    open_gate();
    __builtin_amdgcn_sched_barrier(0);
    MV_STAMP_B(41, MV_STAMP_BLK);
    dj.tables(t, lane);
      dj.finish(a, MB, NH, wave, lane, da, dh_s, dz_s);
      dj.rest(a, MB, B, NH, wave, lane, da, dh_s, dz_s);
    MV_STAMP_B(42, MV_STAMP_BLK);
    if (kLateQ) {
      __builtin_amdgcn_sched_barrier(0);
      request_tile();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int c = wave + kTileWaves * u;
      if (c < MB) {  // uniform
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d = mfma16(da[u][0], wb[0], d);
        d = mfma16(da[u][1], wb[1], d);
        d = mfma16(da[u][2], wb[2], d);
        d = mfma16(da[u][3], wb[3], d);
        // lane (column i, rows 4 q + r of block c): exactly the fragment position (q, i); ReLU mask from h
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = hm[u][r] > 0.f ? d[r] : 0.f;
        frag_s[c][lane] = d;
      }
    }
    MV_STAMP_B(43, MV_STAMP_BLK);
    group_sync(&meet_s, kTileWaves);
    MV_STAMP_B(44, MV_STAMP_BLK);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc, csum = acc;
#pragma unroll
    for (int c = 0; c < MBT; ++c) {
      if (c < MB) {  // uniform
        f32x4 bv = frag_s[c][lane];
        if (bias_wave) csum += bv;
        if (idle_wave) {  // dheads' fragment instead of dh's: rows 4 q + t of block c, head column i
          const int r0 = 16 * c + 4 * q;
          bv = f32x4{dh_s[r0][i], dh_s[r0 + 1][i], dh_s[r0 + 2][i], dh_s[r0 + 3][i]};
        }
        if (have || idle_wave) {
          acc = mfma16(av[c][0], bv[0], acc);
          acc2 = mfma16(av[c][1], bv[1], acc2);
          acc = mfma16(av[c][2], bv[2], acc);
          acc2 = mfma16(av[c][3], bv[3], acc2);
        }
      }
    }
    acc += acc2;
    MV_STAMP_B(45, MV_STAMP_BLK);
    if (bias_wave) {
      // b_e0[p0 + i] = sum over all rows of dh[:, p0 + i]: the lane's 4 MB values, then the four row quads q
      float tsum = (csum[0] + csum[1]) + (csum[2] + csum[3]);
      tsum += __shfl_xor(tsum, 16);
      tsum += __shfl_xor(tsum, 32);
      if (lane < 16) {
        const int col = pt * 16 + lane;
        G[off_b_e0 + col] = tsum;
        if (ADAM) {  // (lanes 0..15: q == 0, so col == bcol)
          adam1(bp, tsum, bm, bvv, neg_step, bc2s);
          base.p[off_b_e0 + col] = bp;
          base.m[off_b_e0 + col] = bm;
          base.v[off_b_e0 + col] = bvv;
        }
      }
    }
    if (have || idle_wave) {
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
      if (ok) {
        store16_wt(G + woff, idx, acc);
        if (ADAM) {
          store16_wt(base.p + woff, idx, p0);
          store16_wt(base.m + woff, idx, m0);
          store16_wt(base.v + woff, idx, v0);
        }
      }
    }
    MV_STAMP_B(46, MV_STAMP_BLK);
    MV_SPAN_END(5, 1);
  }
}

// ---- 6'' (block backward, z_dim 17 .. 64): every weight gradient from fragment-order operands -- dh and dhd arrive in
// fragment order from launches 5 / 4 (never row-major), dheads from launch 5 too, x and h as the copies launch 1 writes, z as the
// copy spare workgroups of launch 4 write.  One tile per wave; the wave whose tile is the first of its P column block
// also delivers that block's bias gradient (the column sums of the P fragments it holds): no column-sum workgroups.
// Grid: 1 (radii) + n_wd0 + n_wh + nt_h * ceil(nt_d / 5) workgroups, short jobs first.
template <bool ADAM, int W>
__global__ __launch_bounds__(64 * W) void k_enc_bwd3(CompTable t, const float* xF, const float* hF, const float* dhF,
                                                  const float* dheadsF, const float* dhdF, const float* zF,
                                                  const float* drpart, float* G, float* P, int B, int H, int D, int NH,
                                                  int Z, int n_wd0, int n_wh, int64_t off_w_e0, int64_t off_b_e0,
                                                  int64_t off_w_heads, int64_t off_b_heads, int64_t off_w_d0,
                                                  int64_t off_b_d0, AdamArgs base, double curv_lr, int do_curv) {
  __shared__ float gsh[kRadiiRegion];
  int b = blockIdx.x;
  const int MB = B >> 4, ntH = H >> 4;
  MV_SPAN_BEGIN(5);
  auto at = [&](int64_t off) {
    AdamArgs a = base;
    a.p += off;
    a.m += off;
    a.v += off;
    return a;
  };
  if (b == 0) {
    job_radii<ADAM>(t, gsh, drpart, G, P, B, curv_lr, do_curv);
    MV_SPAN_END(5, 7);
    return;
  }
  b -= 1;
  const int wave = threadIdx.x >> 6;
  if (b < n_wd0) {  // dW_d0[H, Z] = dhd^T z (+ b_d0 on the z tile 0 of a row tile)
    const int ZT = (Z + 15) >> 4;
    const int tw = b * W + wave, pt = tw / ZT, zt = tw - pt * ZT;
    if (pt < ntH) {
      if (zt == 0)
        job_tn_frag_any<ADAM, true>(dhdF, nullptr, 0, pt, H, zF, zt, Z, MB, G + off_w_d0, Z, at(off_w_d0), G + off_b_d0,
                                    at(off_b_d0));
      else
        job_tn_frag_any<ADAM, false>(dhdF, nullptr, 0, pt, H, zF, zt, Z, MB, G + off_w_d0, Z, at(off_w_d0));
    }
    MV_SPAN_END(5, 3);
    return;
  }
  b -= n_wd0;
  if (b < n_wh) {  // dW_heads[NH, H] = dheads^T h (+ b_heads on the h tile 0 of a heads tile)
    const int NT = (NH + 15) >> 4;
    const int tw = b * W + wave, qt = tw / NT, pt = tw - qt * NT;
    if (qt < ntH) {
      if (qt == 0)
        job_tn_frag_any<ADAM, true>(dheadsF, nullptr, 0, pt, NH, hF, qt, H, MB, G + off_w_heads, H, at(off_w_heads),
                                    G + off_b_heads, at(off_b_heads));
      else
        job_tn_frag_any<ADAM, false>(dheadsF, nullptr, 0, pt, NH, hF, qt, H, MB, G + off_w_heads, H, at(off_w_heads));
    }
    MV_SPAN_END(5, 2);
    return;
  }
  b -= n_wh;
  {  // dW_e0[H, D] = dh^T x (+ b_e0 on the x tile 0 of a row tile)
    const int ntDg = ((D >> 4) + W - 1) / W;
    const int pt = fast_div(b, ntDg), qt = (b - pt * ntDg) * W + wave;
    if (qt == 0)
      job_tn_frag_any<ADAM, true>(dhF, nullptr, 0, pt, H, xF, qt, D, MB, G + off_w_e0, D, at(off_w_e0), G + off_b_e0,
                                  at(off_b_e0));
    else
      job_tn_frag_any<ADAM, false>(dhF, nullptr, 0, pt, H, xF, qt, D, MB, G + off_w_e0, D, at(off_w_e0));
    MV_SPAN_END(5, 1);
  }
}

// ---- 7 (data-parallel / two-call path only): fused optimizer over the flat buffer after the gradient all-reduce
// PEER: the gradient is the sum of the ranks' published slots (mvae_peer.hip), added in rank order -- the same
// floating-point sum on every rank -- and written to g like an all-reduced .grad.
constexpr int kOptU = 4;  // 16-byte vectors per thread of k_optim (1: 41.4 us, 2: 39.1, 4: 37.8, 8: 38.4 for the world-1 forced-exchange step)
static inline int optim_blocks(int n4) { return (n4 - kRadiiRegion / 4 + 256 * kOptU - 1) / (256 * kOptU); }
template <bool PEER>
__global__ __launch_bounds__(256) void k_optim(CompTable t, float* p, float* g, float* m, float* v, int n4,
                                               int* counters, double lr, double curv_lr, int do_curv, PeerSrc ps) {
  __shared__ float sh[2];
  __shared__ float gsh[kMaxComp];
  const int tid = threadIdx.x;
  adam_consts(sh, counters, lr, 1);
  size_t slot_off = 0;
  if (PEER) slot_off = (size_t)(ps.seq[0] & 1) * (size_t)ps.n;
  if (blockIdx.x == 0 && tid < t.n) {
    float gv;
    if (PEER) {
      gv = ps.slot[0][slot_off + tid];  // two-shot: slice 0 (which holds the radii region) was reduced by rank 0
      if (ps.slice4 == 0)
        for (int r = 1; r < ps.world; ++r) gv += ps.slot[r][slot_off + tid];
      g[tid] = gv;
    } else {
      gv = g[tid];
    }
    gsh[tid] = gv;
  }
  __syncthreads();
  const float neg_step = sh[0], bc2s = sh[1];
  // kOptU 16-byte vectors per thread, a grid-sized stride apart (coalesced), every request of the thread in flight
  // before the first use: 8 (16 in the peer forms) 16-byte loads per lane instead of 4 -- the conv architecture's
  // 8.4 MB buffers streamed at 4.1 TB/s with one vector per thread
  float4 pp[kOptU], gg[kOptU], mm[kOptU], vv[kOptU];
  int i4s[kOptU];
#pragma unroll
  for (int u = 0; u < kOptU; ++u) {
    const int i4 = (u * (int)gridDim.x + (int)blockIdx.x) * 256 + tid + kRadiiRegion / 4;
    i4s[u] = i4;
    const int ic = i4 < n4 ? i4 : kRadiiRegion / 4;  // clamped request, masked at the store
    pp[u] = reinterpret_cast<float4*>(p)[ic];
    if (PEER && ps.slice4 > 0) {  // two-shot: the reduced slice is read from its owner
      int owner = (int)(ic / ps.slice4);
      owner = owner < ps.world ? owner : ps.world - 1;
      gg[u] = reinterpret_cast<const float4*>(ps.slot[owner] + slot_off)[ic];
    } else if (PEER) {
      gg[u] = reinterpret_cast<const float4*>(ps.slot[0] + slot_off)[ic];
      for (int r = 1; r < ps.world; ++r) {
        const float4 o = reinterpret_cast<const float4*>(ps.slot[r] + slot_off)[ic];
        gg[u].x += o.x;
        gg[u].y += o.y;
        gg[u].z += o.z;
        gg[u].w += o.w;
      }
    } else {
      gg[u] = reinterpret_cast<const float4*>(g)[ic];
    }
    mm[u] = reinterpret_cast<float4*>(m)[ic];
    vv[u] = reinterpret_cast<float4*>(v)[ic];
  }
#pragma unroll
  for (int u = 0; u < kOptU; ++u) {
    const int i4 = i4s[u];
    if (i4 >= n4) continue;
    if (PEER) store16_wt(g, (size_t)i4 * 4, f32x4{gg[u].x, gg[u].y, gg[u].z, gg[u].w});
    adam1(pp[u].x, gg[u].x, mm[u].x, vv[u].x, neg_step, bc2s);
    adam1(pp[u].y, gg[u].y, mm[u].y, vv[u].y, neg_step, bc2s);
    adam1(pp[u].z, gg[u].z, mm[u].z, vv[u].z, neg_step, bc2s);
    adam1(pp[u].w, gg[u].w, mm[u].w, vv[u].w, neg_step, bc2s);
    // write-through, as in the tile epilogues: nobody in this launch reads these lines again
    store16_wt(p, (size_t)i4 * 4, f32x4{pp[u].x, pp[u].y, pp[u].z, pp[u].w});
    store16_wt(m, (size_t)i4 * 4, f32x4{mm[u].x, mm[u].y, mm[u].z, mm[u].w});
    store16_wt(v, (size_t)i4 * 4, f32x4{vv[u].x, vv[u].y, vv[u].z, vv[u].w});
  }
  if (blockIdx.x == 0 && tid < t.n && t.trainable[tid]) {
    float gv = gsh[tid];
    if (t.trainable[tid] & 2) {  // universal curvature: clipped (after the all-reduce), written back like .grad
      gv *= clip_coef(t, gsh);
      g[tid] = gv;
    }
    if (do_curv & 1) p[tid] = p[tid] + (float)(-curv_lr) * gv;  // SGD: param.add_(grad, alpha=-lr)
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
        if (do_curv & 2) counters[8] = counters[8] + 1;  // (engines whose first launch does not: the conv step) batch cursor
      }
    }
  }
}

// ---- 7', sharded form of the peer exchange (mvae_peer mode 2): the optimizer shrinks with the world size.  Rank r owns slice
// r of the flat buffer: it adds that slice of EVERY rank's published gradients (rank order: the same sum, bit for bit, as
// the other peer forms and -- at world 2 -- as an all-reduce), applies Adam with ITS slice of m and v, and writes the new
// parameters both to its own buffer and INTO its slot, in place of the gradients it has just read (the only reader of slice
// r of rank r's slot in this phase is rank r).  After the second flag round every rank copies the other slices out of
// their owners' slots (k_peer_gather): the same bytes over the links as the two-shot form, 1 / world of the optimizer's
// reads and writes, and no pass over the whole buffer on any rank.  Adam's moments are valid on the owner only.
// Rank 0's slice holds the radii region: clip + SGD there as in k_optim, the region travels with slice 0.
__global__ __launch_bounds__(256) void k_optim_shard(CompTable t, float* p, float* g, float* m, float* v, int n4,
                                                     int* counters, double lr, double curv_lr, int do_curv, PeerSrc ps,
                                                     int rank, float* own_slots) {
  __shared__ float sh[2];
  __shared__ float gsh[kMaxComp];
  const int tid = threadIdx.x;
  adam_consts(sh, counters, lr, 1);
  const size_t slot_off = ps.seq ? (size_t)(ps.seq[0] & 1) * (size_t)ps.n : 0;  // (no sequence word: one buffer, not a slot pair)
  const bool radii_owner = rank == 0 && blockIdx.x == 0;
  if (radii_owner && tid < t.n) {
    float gv = ps.slot[0][slot_off + tid];
    for (int r = 1; r < ps.world; ++r) gv += ps.slot[r][slot_off + tid];
    g[tid] = gv;
    gsh[tid] = gv;
  }
  __syncthreads();
  const float neg_step = sh[0], bc2s = sh[1];
  const long long lo4 = (long long)rank * ps.slice4 + (rank == 0 ? kRadiiRegion / 4 : 0);
  long long hi4 = (long long)(rank + 1) * ps.slice4;
  hi4 = hi4 < n4 ? hi4 : n4;
  float4 pp[kOptU], gg[kOptU], mm[kOptU], vv[kOptU];
  long long i4s[kOptU];
#pragma unroll
  for (int u = 0; u < kOptU; ++u) {
    const long long i4 = lo4 + ((long long)u * gridDim.x + blockIdx.x) * 256 + tid;
    i4s[u] = i4;
    const long long ic = i4 < hi4 ? i4 : kRadiiRegion / 4;  // clamped request, masked at the store
    pp[u] = reinterpret_cast<float4*>(p)[ic];
    gg[u] = reinterpret_cast<const float4*>(ps.slot[0] + slot_off)[ic];
    for (int r = 1; r < ps.world; ++r) {
      const float4 o = reinterpret_cast<const float4*>(ps.slot[r] + slot_off)[ic];
      gg[u].x += o.x;
      gg[u].y += o.y;
      gg[u].z += o.z;
      gg[u].w += o.w;
    }
    mm[u] = reinterpret_cast<float4*>(m)[ic];
    vv[u] = reinterpret_cast<float4*>(v)[ic];
  }
#pragma unroll
  for (int u = 0; u < kOptU; ++u) {
    const long long i4 = i4s[u];
    if (i4 >= hi4) continue;
    store16_wt(g, (size_t)i4 * 4, f32x4{gg[u].x, gg[u].y, gg[u].z, gg[u].w});  // the summed gradient, on its owner
    adam1(pp[u].x, gg[u].x, mm[u].x, vv[u].x, neg_step, bc2s);
    adam1(pp[u].y, gg[u].y, mm[u].y, vv[u].y, neg_step, bc2s);
    adam1(pp[u].z, gg[u].z, mm[u].z, vv[u].z, neg_step, bc2s);
    adam1(pp[u].w, gg[u].w, mm[u].w, vv[u].w, neg_step, bc2s);
    const f32x4 pn{pp[u].x, pp[u].y, pp[u].z, pp[u].w};
    store16_wt(p, (size_t)i4 * 4, pn);
    if (own_slots) store16_wt(own_slots + slot_off, (size_t)i4 * 4, pn);  // what the peers gather
    store16_wt(m, (size_t)i4 * 4, f32x4{mm[u].x, mm[u].y, mm[u].z, mm[u].w});
    store16_wt(v, (size_t)i4 * 4, f32x4{vv[u].x, vv[u].y, vv[u].z, vv[u].w});
  }
  if (radii_owner && tid < kRadiiRegion) {
    float pv = p[tid];
    if (tid < t.n && t.trainable[tid]) {
      float gv = gsh[tid];
      if (t.trainable[tid] & 2) {  // universal curvature: clipped after the reduction
        gv *= clip_coef(t, gsh);
        g[tid] = gv;
      }
      if (do_curv & 1) {
        pv = pv + (float)(-curv_lr) * gv;  // SGD: param.add_(grad, alpha=-lr)
        p[tid] = pv;
      }
    }
    if (own_slots) own_slots[slot_off + tid] = pv;  // the whole radii region travels with slice 0
  }
  if (tid == 0) {  // the last workgroup to arrive advances the step counter (as in k_optim)
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

// Which latent kernels the step takes (see MVAE_PATH_* in the header)
static int latent_path(const mvae_ctx* c, bool x_aligned) {
  const mvae_model_desc& d = c->d;
  const int B = d.batch, H = d.h_dim, D = d.in_dim, NH = d.heads_dim, Z = d.z_dim;
  const float* P = d.params;
  const bool fast = NH <= 16 && Z <= 8 && H <= 512 && (H & 3) == 0 && aligned16(P + d.off_w_heads);
  const bool full = (B % 16 == 0) && (H % 16 == 0) && (D % 16 == 0) && x_aligned && aligned16(P + d.off_w_e0) &&
                    aligned16(P + d.off_w_logits) && aligned16(d.workspace);
  int max_slot = 0;
  for (int i = 0; i < c->t.n; ++i) max_slot = c->t.lane_of[i] > max_slot ? c->t.lane_of[i] : max_slot;
  // the fused forward (launches 2 + 3 in one, k_fwd23) for the shapes it was written for -- any batch size that is a multiple
  // of 16 (up to round 5: multiples of 128 only; measured in round 6 with the four-launch step, us per step fused / per-row
  // kernels: B = 16 27.1 / 31.5, 64 27.2 / 31.9, 112 28.2 / 33.4, 192 41.9 / 44.2).  H <= 416: its staging of the two
  // W_logits row blocks is sized for 32 x 416 floats (kWl = 13 vectors per thread) -- up to round 4 wider layers (H <= 512)
  // were let through and read LDS rows nobody had written
  if (fast && full && H <= 416 && (Z == 8 || Z == 4 || Z == 6 || Z == 2) && d.eps_dim <= 8 && d.ncomp <= 8 &&
      max_slot < 4 && aligned16(P + d.off_w_d0) &&
      bucket_of(c->dmax) <= 8 && !c->no_fwd23)
    return MVAE_PATH_FUSED;
  // many small components: 16-row block kernels (mvae_step_blk.hpp)
  if (!fast && full && c->groups_ok && !c->no_blk && (Z & 3) == 0 && Z <= 64 && H <= 512 && NH <= kHeadsMax &&
      bucket_of(c->dmax) <= 8 && aligned16(P + d.off_w_d0) && aligned16(P + d.off_w_heads))
    return MVAE_PATH_BLOCK;
  return MVAE_PATH_ROW;
}

// the block form of the latent BACKWARD launch: many-component models, and (MVAE_BLK_SMALL) the small-z ones
static bool uses_blk_bwd(const mvae_ctx* c, bool x_aligned) {
  const int path = latent_path(c, x_aligned);
  if (path == MVAE_PATH_BLOCK) return true;
  const mvae_model_desc& d = c->d;
  return path == MVAE_PATH_FUSED && c->blk_small && c->groups_ok && !c->no_blk && (d.z_dim & 3) == 0 && d.z_dim <= 16 &&
         d.heads_dim + d.ncomp <= 64 && aligned16(d.params + d.off_w_d0);
}

extern "C" int mvae_step_kernel_path(const mvae_ctx* c) { return c ? latent_path(c, true) : MVAE_E_BADARG; }

// fused = single-GPU step (Adam/SGD in the gradient epilogues, no k_optim); otherwise gradients only.
static int step_impl(mvae_ctx* c, const float* x, const float* eps, float beta, bool fused, int do_curv,
                     int want_outputs, float* logits, float* concat_z, float* bce, float* kl, void* stream,
                     hipEvent_t* ev) {
  // profile slot of the next launch (0 enc_fwd, 1 latent_fwd, 2 dec1_fwd | the fused 2+3, 3 dec1_bwd, 4 latent_bwd,
  // 5 enc_bwd); with `ev` != NULL the launch is bracketed by ev[2 ki] (start) / ev[2 ki + 1] (stop)
  int ki = 0;
#define STEP_LAUNCH(KERN, GRID, BLOCK, LDS, ...)                                                              \
  do {                                                                                                         \
    if (ev) hipExtLaunchKernelGGL(KERN, GRID, BLOCK, LDS, s, ev[2 * ki], ev[2 * ki + 1], 0, __VA_ARGS__);      \
    else hipLaunchKernelGGL(KERN, GRID, BLOCK, LDS, s, __VA_ARGS__);                                          \
  } while (0)
  if (!c || !x || !eps) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (c->feed.images && (c->feed.x == x || c->feed.eps == eps)) {
    c->feed = FeedArgs{};  // launch 4 would overwrite what launches 4-6 read
    return fail(MVAE_E_BADARG, "mvae_set_next_batch_feed: x_next / eps_next are the buffers this step reads%s", "");
  }
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
  // The four-launch step (k_dec1_bwd<LITE 1> + k_bwd56, fragment-order operands): fused-forward shapes with B <= 256.
  // MVAE_NO_LITE=1: the generic backward launches (A/B measurements, and the other half of the lite-vs-generic parity test).
  const bool lite = full && !c->no_lite && latent_path(c, aligned16(x)) == MVAE_PATH_FUSED && !uses_blk_bwd(c, aligned16(x)) &&
                    fast_b && NH <= 16 && Z <= 8 && B <= 256 && d.ncomp <= kRecRad && c->rec_nv <= kRecVecMax;
  // the block backward (many small components) takes dz from partial products of launch 4's tiles too (z_dim 17 .. 64),
  // and its weight gradients from fragment-order operands (k_enc_bwd3)
  const bool dzp_blk = full && !c->no_lite && uses_blk_bwd(c, aligned16(x)) && Z > 16 && Z <= 64;
  // ... and so does the generic per-row backward on tile-aligned shapes (the large components: `h40` contracted a dhd row with
  // the 65 KB of W_d0 per batch ROW): k_dec1_bwd<LITE 2> with row-major dhd, k_latent_bwd adds the partials
  const bool dzp_row = full && !c->no_lite && !lite && !dzp_blk && !fast_b && !uses_blk_bwd(c, aligned16(x)) && Z <= 64;
  const bool fr6 = lite || dzp_blk || dzp_row;  // launch 6 reads fragment order
  // ... and with the block FORWARD (k_fwd3m) hd exists in fragment order only
  const bool hdf_blk = dzp_blk && latent_path(c, aligned16(x)) == MVAE_PATH_BLOCK && c->blk_fwd;
  float *hdF = (lite || hdf_blk) ? ws + c->o_hdF : nullptr, *xF = fr6 ? ws + c->o_xF : nullptr, *hF = fr6 ? ws + c->o_hF : nullptr,
        *dhdF = fr6 ? ws + c->o_dhdF : nullptr, *zF = fr6 ? ws + c->o_zF : nullptr,
        *dheadsF = fr6 ? ws + c->o_dheadsF : nullptr, *dhF = (dzp_blk || dzp_row) ? ws + c->o_dhF : nullptr;
  // x's copy is written by the padding workgroups of launch 1's XCD-aware grid when it has any, else by short jobs of launch 4
  const bool xf_in_l1 = fr6 && (c->nt_h & 7) != 0;
  if (c->valid_rows < B && !lite && !(dzp_blk && hdf_blk))
    return fail(MVAE_E_UNSUPPORTED, "padding rows (mvae_set_valid_rows) need the four-launch step or the block kernels%s: this call's shape / alignment takes other kernels", "");
  float *dzp = ws + c->o_dzp, *whF = ws + c->o_whF;
  float *recH = ws + c->o_recH, *recR = ws + c->o_recR;
  long long* dzfix = reinterpret_cast<long long*>(ws + c->o_dzfix);
  {
  ki = 0;
  if (full)
    STEP_LAUNCH(k_enc_fwd<true>, dim3(8 * ((c->nt_h + 7) / 8) * c->nt_b), dim3(512), 0, x, P + d.off_w_e0,
                P + d.off_b_e0, h, B, H, D, d.step_count, fused ? 1 : 0, (double)d.lr, xf_in_l1 ? xF : nullptr, hF);
  else
    STEP_LAUNCH(k_enc_fwd<false>, dim3(8 * ((c->nt_h + 7) / 8) * c->nt_b), dim3(512), 0, x, P + d.off_w_e0,
                P + d.off_b_e0, h, B, H, D, d.step_count, fused ? 1 : 0, (double)d.lr, nullptr, nullptr);
  // the fused forward (launches 2 + 3 in one, k_fwd23) for the shapes it was written for
  const int path = latent_path(c, aligned16(x));
  const bool fwd23 = path == MVAE_PATH_FUSED, blk = path == MVAE_PATH_BLOCK;
  if (fwd23) {
    ki = 2;
    const size_t lds = ((size_t)48 * (H + 4) + (size_t)H * 9) * sizeof(float);
    const int n_main23 = ((c->nt_d + 1) / 2) * c->nt_b;  // no padding workgroups (see the kernel)
#define LF23(DM)                                                                                                     \
  {                                                                                                                  \
    /* more than 64 KB of dynamic LDS has to be allowed once per kernel AND DEVICE (the attribute is per device) */   \
    static std::atomic<size_t> lds_set[kMaxDevices];                                                                 \
    int dev_ = 0;                                                                                                    \
    (void)hipGetDevice(&dev_);                                                                                       \
    const bool tracked_ = dev_ >= 0 && dev_ < kMaxDevices;                                                           \
    if (!tracked_ || lds > lds_set[dev_].load(std::memory_order_acquire)) {                                          \
      auto kfn_ = &k_fwd23<DM>;                                                                                      \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn_),                                       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e_ != hipSuccess) return hip_fail(e_, "hipFuncSetAttribute(k_fwd23)");                                     \
      if (tracked_) lds_set[dev_].store(lds, std::memory_order_release);                                             \
    }                                                                                                                \
  }                                                                                                                  \
  STEP_LAUNCH((k_fwd23<DM>), dim3(n_main23 + c->nt_b + MV_PREFETCH_WGS), dim3(512), lds, c->t, h, P + d.off_w_heads,  \
              P + d.off_b_heads, eps, d.eps_dim, P + d.off_radii, P + d.off_w_d0, P + d.off_b_d0, P + d.off_w_logits, \
              P + d.off_b_logits, x, heads, c->ldh, z, c->ldz, concat_z, klw, kl, hd, g, bce_part, logits, B, H, D,   \
              NH, Z, duals, zF, hdF, r4)
    const Rec4Args r4 = {lite ? recH : nullptr, recR, lite ? dzfix : nullptr, c->rec_nv, c->valid_rows};
    const int bk = bucket_of(c->dmax);
    if (bk == 2) { LF23(2); } else if (bk == 4) { LF23(4); } else { LF23(8); }
#undef LF23
  } else if (blk && c->blk_fwd) {
    ki = 1;
#define LHC(DM)                                                                                                      \
  STEP_LAUNCH((k_heads_comp<DM>), dim3(c->nt_b * c->gt.ng), dim3(512), 0, c->t, c->gt, h, P + d.off_w_heads,          \
              P + d.off_b_heads, eps, d.eps_dim, P + d.off_radii, heads, c->ldh, z, c->ldz, concat_z, klw, kl, B, H,  \
              NH, Z, c->valid_rows)
    const int bk = bucket_of(c->dmax);
    if (bk == 2) { LHC(2); } else if (bk == 4) { LHC(4); } else { LHC(8); }
#undef LHC
    ki = 2;
    const size_t lds = (size_t)16 * (H + 4) * sizeof(float);
    const int n_dual3 = c->nt_b * c->gt.ng;  // a multiple of 8 for B % 128 == 0: the tile workgroups keep L % 8 == XCD
#define LF3(DM)                                                                                                      \
  STEP_LAUNCH((k_fwd3m<DM>), dim3(n_dual3 + ((c->nt_d + 1) / 2) * c->nt_b), dim3(512), lds, c->t, c->gt, heads,       \
              c->ldh, eps, d.eps_dim, P + d.off_radii, NH, duals, n_dual3, z, c->ldz, P + d.off_w_d0, P + d.off_b_d0, \
              P + d.off_w_logits, P + d.off_b_logits, x, hd, g, bce_part, logits, B, H, D, Z,  \
              hdF, c->valid_rows)
    if (bk == 2) { LF3(2); } else if (bk == 4) { LF3(4); } else { LF3(8); }
#undef LF3
  } else {
  {
    ki = 1;
    const size_t lds = (((size_t)H + 3) & ~(size_t)3) * sizeof(float) + ((size_t)d.eps_dim + 4) * sizeof(float);
    // large components (d >= 9): the components of launch 2 one WAVE each, the dual records one wave per (row, direction) as
    // extra workgroups of launch 3 -- instead of one lane each over scratch-resident vectors
    const bool coop = c->coop;
#define LF(DM, FA, CO)                                                                                               \
  STEP_LAUNCH((k_latent_fwd<DM, FA, CO>), dim3(B), dim3(CO ? 256 : 512), lds, c->t, h, P + d.off_w_heads,            \
                     P + d.off_b_heads, eps, d.eps_dim, P + d.off_radii, P + d.off_w_d0, P + d.off_b_d0, heads,      \
                     c->ldh, z, c->ldz, concat_z, klw, kl, hd, B, H, NH, Z, duals)
    if (coop) { if (fast) LF(2, true, true); else LF(2, false, true); }
    else if (fast) { DMAX_SWITCH(c->dmax, LF(DM, true, false)); } else { DMAX_SWITCH(c->dmax, LF(DM, false, false)); }
#undef LF
  }
  ki = 2;
  const int n_tile3 = 8 * ((c->nt_d + 7) / 8) * c->nt_b;
  if (c->coop) {
    // the dual records of the large components (one wave per (row, direction)) as extra workgroups behind launch 3's tiles
    const int ngroups = (c->t.total_dirs + 7) / 8;
    const CoopDualArgs cd = {heads, eps, P + d.off_radii, duals, c->ldh, d.eps_dim, NH, dual_stride(bucket_of(c->dmax)), ngroups,
                             n_tile3};
    if (full)
      STEP_LAUNCH(k_dec1_fwd_duals<true>, dim3(n_tile3 + B * ngroups), dim3(512), 0, hd, P + d.off_w_logits,
                  P + d.off_b_logits, x, g, bce_part, logits, B, H, D, c->t, cd);
    else
      STEP_LAUNCH(k_dec1_fwd_duals<false>, dim3(n_tile3 + B * ngroups), dim3(512), 0, hd, P + d.off_w_logits,
                  P + d.off_b_logits, x, g, bce_part, logits, B, H, D, c->t, cd);
  } else if (full)
    STEP_LAUNCH(k_dec1_fwd<true>, dim3(n_tile3), dim3(512), 0, hd, P + d.off_w_logits,
                P + d.off_b_logits, x, g, bce_part, logits, B, H, D);
  else
    STEP_LAUNCH(k_dec1_fwd<false>, dim3(n_tile3), dim3(512), 0, hd, P + d.off_w_logits,
                P + d.off_b_logits, x, g, bce_part, logits, B, H, D);
  }
  {
    const int n_dhd = xcd_grid(c->nt_h, c->nt_b, kDhdGroup), n_db = (D + kColsPerBlock - 1) / kColsPerBlock;
    ki = 3;
    DualArgs da = {heads, eps, P + d.off_radii, duals, c->ldh, d.eps_dim, NH, 0};
    const bool duals_in_l4 = false;  // the fused forward's dual workgroups produce the records (job_duals stays available)
    if (fwd23 && duals_in_l4) da.n_dual = (B * c->t.total_dirs + 63) / 64;
    const FeedArgs fd = c->feed;  // one-shot: consumed by this step
    c->feed = FeedArgs{};
    const FragArgs fr = {hdF, dhdF, x, xF, (fr6 && !xf_in_l1) ? c->nt_d : 0, (dzp_blk || dzp_row) ? dzp : nullptr, P + d.off_w_d0, Z,
                         z, zF, c->ldz, (dzp_blk || dzp_row) ? (Z + 15) / 16 : 0, hdf_blk ? 1 : 0,
                         lite ? dzfix : nullptr, P + d.off_w_heads, whF, lite ? 4 : 0, NH, c->valid_rows};
    const int n_short = (da.n_dual + 1 + n_db + fd.n_wg + fr.n_xf + fr.n_zf + fr.n_snap + 7) & ~7;
#define DBX(AD, FU, DU, LI)                                                                                    \
  STEP_LAUNCH((k_dec1_bwd<AD, FU, DU, LI>), dim3(n_dhd + n_short), dim3(512), 0, c->t, g, hd, P + d.off_w_logits, \
              G + d.off_b_logits, dhd, bce_part, klw, bce, d.stats, beta, B, H, D, d.ncomp, n_dhd, n_db,          \
              at(d.off_b_logits), da, fd, fr)
#define DB(AD, FU, DU) DBX(AD, FU, DU, 0)
    if (lite) {
      if (fused) DBX(true, true, 0, 1); else DBX(false, true, 0, 1);
    } else if (dzp_blk || dzp_row) {
      if (fused) DBX(true, true, 0, 2); else DBX(false, true, 0, 2);
    } else
    if (fwd23 && duals_in_l4) {  // full && dmax bucket in {2, 4, 8}
      const int bk = bucket_of(c->dmax);
      if (fused) { if (bk == 2) DB(true, true, 2); else if (bk == 4) DB(true, true, 4); else DB(true, true, 8); }
      else { if (bk == 2) DB(false, true, 2); else if (bk == 4) DB(false, true, 4); else DB(false, true, 8); }
    } else if (fused) { if (full) DB(true, true, 0); else DB(true, false, 0); }
    else { if (full) DB(false, true, 0); else DB(false, false, 0); }
#undef DB
#undef DBX
  }
  {
    ki = 4;
    const int ntHg5 = (c->nt_h + kTileWaves5 - 1) / kTileWaves5;
    const int n_dwl = c->nt_d * ntHg5;
    const size_t lds = ((((size_t)H + 3) & ~(size_t)3) + 1024 + 8) * sizeof(float);  // dhd row | dz partials
#define LB(DM, FA, AD)                                                                                              \
  STEP_LAUNCH((k_latent_bwd<DM, FA, AD>), dim3(B + n_dwl), dim3(64 * kTileWaves5), lds, c->t, dhd, P + d.off_w_d0, \
                     c->ldh, h, P + d.off_w_heads, dheads, dh, drpart, g,                                           \
                     hd, G + d.off_w_logits, beta, B, H, D, NH, Z, B, at(d.off_w_logits), duals, dzp_row ? dzp : nullptr, dzp_row ? dheadsF : nullptr, dzp_row ? dhF : nullptr)
    if (lite) {
      // (no launch 5: k_bwd56, the TAIL part, does its work)
    } else if (uses_blk_bwd(c, aligned16(x))) {
      const int n_blk = c->nt_b * ((H + 63) / 64);
      const size_t lds_b = Z <= 16 ? (size_t)H * Z * sizeof(float) : 0;
      const int4* dirtab = reinterpret_cast<const int4*>(ws + c->o_dirtab);
      const StatsArgs sa = {hdf_blk ? bce_part : nullptr, klw, bce, d.stats, d.ncomp, c->valid_rows};  // (launch 4 skipped it: FragArgs::stats_later)
      const int n_tile_wg = hdf_blk ? (c->nt_d * c->nt_h + 7) / 8 : c->nt_d * ntHg5;
#define LBB(DM, AD, TT)                                                                                               \
  STEP_LAUNCH((k_latent_bwd_blk<DM, AD, TT>), dim3(n_blk + n_tile_wg + (hdf_blk ? 1 : 0)), dim3(hdf_blk ? 512 : 256), lds_b, c->t, dirtab, dhd, P + d.off_w_d0, \
              c->ldh, h, P + d.off_w_heads, dheads, dh, drpart, g, hd, G + d.off_w_logits, beta, B, H, D, NH, Z,       \
              n_blk, at(d.off_w_logits), duals, dzp_blk ? dzp : nullptr, dhF, hF, dzp_blk ? dheadsF : nullptr, hdf_blk ? hdF : nullptr, sa)
#define LBB2(DM, AD) do { if (Z <= 16) LBB(DM, AD, 1); else if (Z <= 48) LBB(DM, AD, 3); else LBB(DM, AD, 4); } while (0)
      const int bk = bucket_of(c->dmax);
      if (fused) { if (bk == 2) LBB2(2, true); else if (bk == 4) LBB2(4, true); else LBB2(8, true); }
      else { if (bk == 2) LBB2(2, false); else if (bk == 4) LBB2(4, false); else LBB2(8, false); }
#undef LBB2
#undef LBB
    } else if (fast_b) {
      if (fused) { DMAX_SWITCH(c->dmax, LB(DM, true, true)); } else { DMAX_SWITCH(c->dmax, LB(DM, true, false)); }
    } else {
      if (fused) { DMAX_SWITCH(c->dmax, LB(DM, false, true)); } else { DMAX_SWITCH(c->dmax, LB(DM, false, false)); }
    }
#undef LB
  }
  }
  {
    ki = 5;
    const int tw = kTileWaves;
    const int n_we0 = c->nt_h * ((c->nt_d + tw - 1) / tw), n_wh = ((NH + 15) / 16) * ((c->nt_h + tw - 1) / tw),
              n_wd0 = c->nt_h * (((Z + 15) / 16 + tw - 1) / tw);
    const int n_be0 = (H + kColsPerBlock - 1) / kColsPerBlock, n_bh = (NH + kColsPerBlock - 1) / kColsPerBlock,
              n_bd0 = n_be0;
    const int grid = n_we0 + n_wh + n_wd0 + n_be0 + n_bh + n_bd0 + 1;
    if (lite) {
      const int n_small = (c->nt_h + tw - 1) / tw;
      const int grid2 = 1 + n_small + n_we0;
      const L56Args la = {dzfix, recH, recR, g, hdF, dheads, drpart, c->ldh, beta, d.off_w_logits, c->valid_rows};
      const float* scal_p = reinterpret_cast<const float*>(d.step_count) + 2;
      const unsigned* mark_p = reinterpret_cast<const unsigned*>(dzfix + (size_t)B * 8);
#define B56(NVV, AD, MBT)                                                                                            \
  STEP_LAUNCH((k_bwd56<NVV, AD, MBT>), dim3(grid2), dim3(64 * kW56), 0, c->t, la, scal_p, mark_p, xF, hF, whF, dhdF, zF, G, P, B, H, D, \
              NH, Z, n_small, (c->nt_d % tw) != 0 ? 1 : 0, d.off_w_e0, d.off_b_e0, d.off_w_heads, d.off_b_heads,      \
              d.off_w_d0, d.off_b_d0, base, (double)d.curvature_lr, do_curv)
#define B56N(AD, MBT) do { if (c->rec_nv == 1) B56(1, AD, MBT); else if (c->rec_nv == 2) B56(2, AD, MBT); else B56(3, AD, MBT); } while (0)
      if (fused) { if (B <= 128) B56N(true, 8); else B56N(true, 16); }
      else { if (B <= 128) B56N(false, 8); else B56N(false, 16); }
#undef B56N
#undef B56
    } else if (dzp_blk || dzp_row) {
      // seven waves per workgroup: 49 x tiles of a dh column block = 7 x 7, and 1 + 11 + 18 + 175 workgroups for config [3] fit
      // one per CU (five-wave workgroups, 291 of them: 39.0 us per step against 37.8)
      constexpr int w3 = 7;
      const int n_wd0f = (c->nt_h * ((Z + 15) / 16) + w3 - 1) / w3, n_whf = (((NH + 15) / 16) * c->nt_h + w3 - 1) / w3;
      const int n_we0f = c->nt_h * ((c->nt_d + w3 - 1) / w3);
#define EB3(AD) EB3W(AD, w3)
#define EB3W(AD, WV)                                                                                                      \
  STEP_LAUNCH((k_enc_bwd3<AD, WV>), dim3(1 + n_wd0f + n_whf + n_we0f), dim3(64 * WV), 0, c->t, xF, hF, dhF, dheadsF, \
              dhdF, zF, drpart, G, P, B, H, D, NH, Z, n_wd0f, n_whf, d.off_w_e0, d.off_b_e0,           \
              d.off_w_heads, d.off_b_heads, d.off_w_d0, d.off_b_d0, base, (double)d.curvature_lr, do_curv)
      if (fused) EB3(true); else EB3(false);
#undef EB3
#undef EB3W
    } else
#define EB(AD, FU)                                                                                                   \
  STEP_LAUNCH((k_enc_bwd<AD, FU>), dim3(grid), dim3(64 * kTileWaves), 0, c->t, dh, x, dheads, c->ldh, h, dhd, z, c->ldz, \
                     drpart, G, P, B, H, D, NH, Z, n_we0, n_wh, n_wd0, n_be0, n_bh, n_bd0, d.off_w_e0, d.off_b_e0,   \
                     d.off_w_heads, d.off_b_heads, d.off_w_d0, d.off_b_d0, base, (double)d.curvature_lr, do_curv)
    {
    if (fused) { if (full) EB(true, true); else EB(true, false); }
    else { if (full) EB(false, true); else EB(false, false); }
    }
#undef EB
  }
#undef STEP_LAUNCH
  LAUNCH_CHECK("step launch");
  return 0;
}

extern "C" int mvae_set_next_batch_feed(mvae_ctx* c, const uint8_t* images, const int32_t* perm, int n_images, uint64_t seed,
                                        int batches_per_epoch, int train, float* x_next, float* eps_next) {
  if (!c) return fail(MVAE_E_BADARG, "null ctx%s", "");
  if (!images) {
    c->feed = FeedArgs{};
    return 0;
  }
  if (!x_next || !eps_next || n_images < 1 || batches_per_epoch < 1 || train < 0 || train > 2)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  const mvae_model_desc& d = c->d;
  // (with padding rows the batch of the data set is the VALID rows; rows past them are never written and stay what the caller
  // made them -- zeros)
  FeedArgs f = {images, perm, d.step_count, x_next, eps_next, (unsigned long long)seed, n_images, d.in_dim, c->valid_rows,
                d.eps_dim, batches_per_epoch, train, 0};
  // two items (of four values) per thread of a 512-thread workgroup; the workgroups join launch 4's short jobs
  static const int per_thread = [] { const char* e = getenv("MVAE_FEED_ITEMS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();
  const int items = (c->valid_rows * d.in_dim + 3) / 4 + (c->valid_rows * d.eps_dim + 3) / 4;
  f.n_wg = (items + 512 * per_thread - 1) / (512 * per_thread);
  if (f.n_wg > 128) f.n_wg = 128;
  c->feed = f;
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
  const int blocks = optim_blocks(n4);
  hipLaunchKernelGGL(k_optim<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c->t, d.params, d.grads, d.adam_m,
                     d.adam_v, n4, d.step_count, (double)d.lr, (double)d.curvature_lr, do_curvature_step ? 1 : 0, PeerSrc{});
  LAUNCH_CHECK("optimizer launch");
  return 0;
}

extern "C" int mvae_step_optimizer_peer(mvae_ctx* c, mvae_peer* peer, int do_curvature_step, void* stream) {
  if (!c || !peer) return fail(MVAE_E_BADARG, "null pointer%s", "");
  const mvae_model_desc& d = c->d;
  if (peer->n != d.n_params) return fail(MVAE_E_BADARG, "peer slots were sized for another parameter count%s", "");
  PeerSrc ps{};
  for (int r = 0; r < peer->world; ++r) {
    if (!peer->imported[r]) return fail(MVAE_E_BADARG, "peer %s%lld has not been imported", "", r);
    ps.slot[r] = peer->peer_slots[r];
  }
  ps.seq = peer->seq;
  ps.n = peer->n;
  ps.world = peer->world;
  ps.slice4 = peer->mode ? peer_slice4(peer) : 0;
  const int n4 = d.n_params / 4;
  if (peer->mode == 2) {  // sharded optimizer: Adam on the owned slice, then the all-gather of parameters
    const long long s4 = ps.slice4;
    int sb = (int)((s4 + 256 * kOptU - 1) / (256 * kOptU));
    sb = sb < 1 ? 1 : sb;
    hipLaunchKernelGGL(k_optim_shard, dim3(sb), dim3(256), 0, (hipStream_t)stream, c->t, d.params, d.grads, d.adam_m,
                       d.adam_v, n4, d.step_count, (double)d.lr, (double)d.curvature_lr, do_curvature_step ? 1 : 0, ps,
                       peer->rank, peer->slots);
    LAUNCH_CHECK("sharded peer optimizer launch");
    return peer_gather_params(peer, d.params, (hipStream_t)stream);
  }
  const int blocks = optim_blocks(n4);
  hipLaunchKernelGGL(k_optim<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c->t, d.params, d.grads, d.adam_m,
                     d.adam_v, n4, d.step_count, (double)d.lr, (double)d.curvature_lr, do_curvature_step ? 1 : 0, ps);
  LAUNCH_CHECK("peer optimizer launch");
  return 0;
}

// The sharded optimizer for exchanges that leave the SUMMED gradient of the rank's slice in `grads` itself (a reduce-scatter in
// place: mvae_flat_reduce_scatter, or a whole all-reduce): Adam on slice `rank` of `world` only, radii on rank 0; the caller
// all-gathers `params` afterwards.  The slices are the peer route's: s4 = max(16, ceil(n_params / 4 / world)) float4 each.
extern "C" int mvae_step_optimizer_slice(mvae_ctx* c, int rank, int world, int do_curvature_step, void* stream) {
  if (!c) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (world < 1 || rank < 0 || rank >= world) return fail(MVAE_E_BADARG, "rank must be in [0, world)%s", "");
  const mvae_model_desc& d = c->d;
  const int n4 = d.n_params / 4;
  PeerSrc ps{};
  ps.slot[0] = d.grads;
  ps.seq = nullptr;
  ps.n = d.n_params;
  ps.world = 1;  // one source: the gradient is already summed
  long long s4 = ((long long)n4 + world - 1) / world;
  ps.slice4 = s4 < 16 ? 16 : s4;
  int sb = (int)((ps.slice4 + 256 * kOptU - 1) / (256 * kOptU));
  sb = sb < 1 ? 1 : sb;
  hipLaunchKernelGGL(k_optim_shard, dim3(sb), dim3(256), 0, (hipStream_t)stream, c->t, d.params, d.grads, d.adam_m, d.adam_v,
                     n4, d.step_count, (double)d.lr, (double)d.curvature_lr, do_curvature_step ? 1 : 0, ps, rank,
                     (float*)nullptr);
  LAUNCH_CHECK("sliced optimizer launch");
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
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize"); break; }
    for (int k = 0; k < NK; ++k) {  // a slot the step did not use (launch 2 of the fused forward) stays 0
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) != hipSuccess) {
        ms = 0.f;
        (void)hipGetLastError();
      }
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
                                        int do_curvature_step, int advance_cursor, void* stream) {
  if (!params || !grads || !adam_m || !adam_v || !counters || n_params < kRadiiRegion || (n_params & 3) ||
      ncomp < 0 || ncomp > kMaxComp)
    return fail(MVAE_E_BADARG, "null pointer / bad size%s", "");
  CompTable t;
  memset(&t, 0, sizeof(t));
  t.n = ncomp;
  for (int i = 0; i < ncomp; ++i) t.trainable[i] = radius_trainable ? radius_trainable[i] : 0;
  const int n4 = (int)(n_params / 4);
  const int blocks = optim_blocks(n4);
  hipLaunchKernelGGL(k_optim<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, params, grads, adam_m, adam_v,
                     n4, counters, lr, curvature_lr, (do_curvature_step ? 1 : 0) | (advance_cursor ? 2 : 0), PeerSrc{});
  LAUNCH_CHECK("flat optimizer launch");
  return 0;
}
