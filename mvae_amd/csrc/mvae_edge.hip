// mvae_edge.hip -- the two 3-channel layers at the image boundary of the conv architecture, without a patch matrix.
//
// conv_vae.py:47,57  e0 = Conv2d(3, 64, 4, 2, 1) on the NCHW input x [B, 3, 32, 32]            (forward + weight gradient)
// conv_vae.py:54,74  d3 = ConvTranspose2d(64, 3, 4, 2, 1) to the NCHW logits [B, 3, 32, 32]      (backward-data + weight gradient)
// Both pairs are contractions against the [B * 256, 48] patch matrix of a 3 x 32 x 32 image (column c * 16 + ky * 4 + kx of
// pixel (b, oy, ox) = img[b, c, 2 oy - 1 + ky, 2 ox - 1 + kx], 0 outside).  Until round 4 that matrix was written by k_im2col
// (12.6 MB, ~14 us per image batch) and read back by a generic contraction each for the activation side (~20 us: 55-70 MB at
// 3 TB/s) and the weight gradient (~16 us); together 99 us of the 0.81 ms step for 1.6 GFLOP.  Here the patch entries are
// fetched straight from the image into the MFMA fragments:
//   * along a row of 16 output pixels, fragment lane (pixel ox = l & 15, k = l >> 4) of K step s = (c, ky) reads
//     img[b, c, 2 oy - 1 + ky, 2 ox - 1 + (l >> 4)]: ONE 128-byte image row per wave-level load, 12 loads per 16 pixels;
//   * v_mfma_f32_16x16x4_f32 in ascending k, one accumulator chain per output: the same sequence of matrix instructions as
//     the patch-matrix route through k_gemm_tiled (zero-padded K steps add +0), so k_edge3_nt returns the SAME BITS;
//   * the activation side (65536 x 64 f32 + optionally its bf16 planes, mvae_p3.hpp) is written once, 16 bytes per lane.
// An earlier direct attempt (round 2, VALU dot products from an LDS-staged window) lost to the patch matrix; this one is bound
// by the bytes it must write (k_edge3_nt) / read (k_edge3_tn).  Fixed geometry: 3 channels, 32 x 32 image, 64 features.
#include "mvae_common.hpp"
#include "mvae_p3.hpp"

constexpr int kEC = 3, kEH = 32, kEO = 16, kEF = 64, kEK = kEC * 16;  // channels, image extent, feature-map extent, features, patch

// y[(b, oy, ox), n] = mask(relu(bias[n] + sum_k patch(b, oy, ox; k) W[n, k])), + its planes.  One wave = G rows of 16 pixels.
// The CU's address path prices a request by the 128-byte lines it touches (~2.8 cycles each, mvae_p3.hip), so
//   * W (12 KB) is staged once per workgroup through LDS (coalesced 16-byte loads) instead of 48 strided loads per lane;
//   * the accumulators (lane = 4 features of one pixel: 64-byte pieces of 16 different rows per store) pass through a
//     per-wave LDS tile and leave as whole rows: a store covers 4 pixels x 256 bytes (f32) / 4 x 128 bytes (a plane), the
//     mask is read the same way -- 92-124 lines per 16 pixels instead of 268-332.
constexpr int kEWS = kEK + 1;   // LDS row stride of W (floats): 49 is odd, the 16 rows of a fragment fall into 16 banks
constexpr int kETS = kEF + 4;   // LDS row stride of a wave's result tile (floats)
template <int G, int NW>
__device__ __forceinline__ void edge3_nt_body(const int blk, const float* __restrict__ img, const float* __restrict__ W,
                                              const float* __restrict__ bias, const float* __restrict__ mask, const int relu,
                                              float* __restrict__ y, bf16r* __restrict__ yp, const long long ps, const int nrows,
                                              float* __restrict__ colpart = nullptr) {
  __shared__ float sW[kEF * kEWS];
  __shared__ __attribute__((aligned(16))) float sT[NW][16 * kETS];
  __shared__ __attribute__((aligned(16))) float sC[NW][kEF];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  for (int e = threadIdx.x; e < kEF * kEK / 4; e += 64 * NW) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(W + 4 * e);
    const int n = (4 * e) / kEK, k = (4 * e) % kEK;
#pragma unroll
    for (int r = 0; r < 4; ++r) sW[n * kEWS + k + r] = v[r];
  }
  const int row0 = (blk * NW + wave) * G;
  const int ix = 2 * l15 - 1 + l4;
  const bool okx = ix >= 0 && ix < kEH;
  float pf[G][12];
#pragma unroll
  for (int gI = 0; gI < G; ++gI) {
    const int row = row0 + gI, b = row >> 4, oy = row & 15;
    const bool live = row < nrows;
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int c = s >> 2, iy = 2 * oy - 1 + (s & 3);
      const bool ok = live && okx && iy >= 0 && iy < kEH;
      pf[gI][s] = ok ? img[(((size_t)b * kEC + c) * kEH + iy) * kEH + ix] : 0.f;
    }
  }
  // the ReLU mask of the wave's rows travels with the gather (asked for in the epilogue it is a memory round trip of its own
  // between the last MFMA and the first store)
  f32x4 mkv[G][4];
#pragma unroll
  for (int gI = 0; gI < G; ++gI)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + gI, px = 4 * i + l4;
      mkv[gI][i] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (mask && row < nrows) mkv[gI][i] = *reinterpret_cast<const f32x4*>(mask + ((size_t)row * 16 + px) * kEF + 4 * l15);
    }
  __syncthreads();
  if (row0 >= nrows && !colpart) return;
  // weight fragments for the whole life of the wave: operand "a" of tile t, step s = W[16 t + l15][4 s + l4]
  float wf[4][12];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = 0; s < 12; ++s) wf[t][s] = sW[(16 * t + l15) * kEWS + 4 * s + l4];
  // the row-wise side of the epilogue: lane = features 4 l15 .. + 3 of pixels l4, l4 + 4, l4 + 8, l4 + 12
  f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + 4 * l15);
  float* tile = sT[wave];
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};  // column sums of this wave's rows: features 4 l15 .. + 3 (pixels l4 + 4 i)
#pragma unroll
  for (int gI = 0; gI < G; ++gI) {
    const int row = row0 + gI;
    if (row >= nrows) break;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 12; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][s], pf[gI][s], acc[t], 0, 0, 0);
    // lane holds features 16 t + 4 l4 .. + 3 of pixel l15 -> the wave's LDS tile [pixel][feature]
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(tile + l15 * kETS + 16 * t + 4 * l4) = acc[t];
    // (one wave, LDS operations complete in order: no barrier)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = 4 * i + l4;
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + px * kETS + 4 * l15);
      const size_t o = ((size_t)row * 16 + px) * kEF + 4 * l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bv[r];
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (mask) {
        const f32x4 mk = mkv[gI][i];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (mk[r] > 0.f) ? v[r] : 0.f;
      }
      if (y) *reinterpret_cast<f32x4*>(y + o) = v;
      if (yp) store_planes4(yp, ps, o, v[0], v[1], v[2], v[3]);
      cs += v;
    }
  }
  // column sums of the workgroup's rows (the bias gradient of the layer whose backward-data this is): the four pixel groups of
  // a wave by two lane exchanges, the waves through LDS in wave order; the workgroups by the caller's (deferred) column sum
  if (colpart) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = cs[r] + __shfl_xor(cs[r], 16);
      cs[r] = t + __shfl_xor(t, 32);
    }
    if (l4 == 0) *reinterpret_cast<f32x4*>(&sC[wave][4 * l15]) = cs;
    __syncthreads();
    if (threadIdx.x < kEF) {
      float t = sC[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += sC[w][threadIdx.x];
      colpart[(size_t)blk * kEF + threadIdx.x] = t;
    }
  }
}

template <int G, int NW>
__global__ __launch_bounds__(64 * NW) void k_edge3_nt(const float* __restrict__ img, const float* __restrict__ W,
                                                  const float* __restrict__ bias, const float* __restrict__ mask, const int relu,
                                                  float* __restrict__ y, bf16r* __restrict__ yp, const long long ps,
                                                  const int nrows) {
  edge3_nt_body<G, NW>((int)blockIdx.x, img, W, bias, mask, relu, y, yp, ps, nrows);
}

// part[wg][n, k] = sum over the workgroup's pixels of act[p, n] patch(p; k): 8 waves x 32 pixels of one image per pass, the
// contraction index is the pixel.  Tile t of the "a" operand takes row i from feature 4 i + t, so ONE 16-byte load per lane
// (act[p][4 l15 .. + 3], p = p0 + 4 s + l4) feeds the four feature tiles of a step; tile u of "b" is image channel u, column
// j = (ky, kx).  The eight waves' sums are added through LDS in a fixed order, the workgroups' by the (deferrable) slice sum.
__device__ __forceinline__ void edge3_tn_body(const int blk, const float* __restrict__ act, const float* __restrict__ img,
                                              float* __restrict__ part, const int B, const int img_per_wg) {
  __shared__ __attribute__((aligned(16))) float red[4][kEF * kEK];  // 48 KB
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int ky = l15 >> 2, kx = l15 & 3;
  f32x4 acc[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 3; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ii = 0; ii < img_per_wg; ++ii) {
    const int b = blk * img_per_wg + ii;
    if (b >= B) break;
    f32x4 af[8];
    float bf[8][3];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int pix = wave * 32 + 4 * s + l4, oy = pix >> 4, ox = pix & 15;
      af[s] = *reinterpret_cast<const f32x4*>(act + ((size_t)b * 256 + pix) * kEF + 4 * l15);
      const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
      const bool ok = iy >= 0 && iy < kEH && ix >= 0 && ix < kEH;
#pragma unroll
      for (int u = 0; u < 3; ++u) bf[s][u] = ok ? img[(((size_t)b * kEC + u) * kEH + iy) * kEH + ix] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][t], bf[s][u], acc[t][u], 0, 0, 0);
  }
  // lane holds rows 4 l4 + r (features 4 (4 l4 + r) + t), column l15 (k = 16 u + l15) of tile (t, u).  Waves 4-7 hand their
  // sums to waves 0-3 (w += w + 4), then the four are added in wave order: a fixed order, whatever the timing.
  if (wave >= 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 4][(4 * (4 * l4 + r) + t) * kEK + 16 * u + l15] = acc[t][u][r];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* q = &red[wave][(4 * (4 * l4 + r) + t) * kEK + 16 * u + l15];
          *q = acc[t][u][r] + *q;
        }
  }
  __syncthreads();
  float* dst = part + (size_t)blk * (kEF * kEK);
  for (int e = threadIdx.x; e < kEF * kEK; e += 512) dst[e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

__global__ __launch_bounds__(512) void k_edge3_tn(const float* __restrict__ act, const float* __restrict__ img,
                                                  float* __restrict__ part, const int B, const int img_per_wg) {
  edge3_tn_body((int)blockIdx.x, act, img, part, B, img_per_wg);
}
// the same launch carrying the queued column sums of the backward pass (bias gradients) as extra workgroups
__global__ __launch_bounds__(512) void k_edge3_tn_cols(const float* __restrict__ act, const float* __restrict__ img,
                                                       float* __restrict__ part, const int B, const int img_per_wg,
                                                       const int n_tn, const ColJobs cols) {
  if ((int)blockIdx.x < n_tn) edge3_tn_body((int)blockIdx.x, act, img, part, B, img_per_wg);
  else colsum_batched_body(cols, (int)blockIdx.x - n_tn);
}
// both backward contractions of d3 in one launch (they read the same gradient image and not each other): workgroups
// 0 .. n_tn - 1 the weight gradient, the rest the backward-data
__global__ __launch_bounds__(512) void k_edge3_bwd(const float* __restrict__ act, const float* __restrict__ img,
                                                   float* __restrict__ part, const int B, const int img_per_wg, const int n_tn,
                                                   const float* __restrict__ W, const float* __restrict__ mask,
                                                   float* __restrict__ y, bf16r* __restrict__ yp, const long long ps,
                                                   const int nrows, float* __restrict__ colpart) {
  if ((int)blockIdx.x < n_tn) edge3_tn_body((int)blockIdx.x, act, img, part, B, img_per_wg);
  else edge3_nt_body<1, 8>((int)blockIdx.x - n_tn, img, W, nullptr, mask, 0, y, yp, ps, nrows, colpart);
}

void p3_sum_slices(const float* part, float* out, int64_t n, int slices, hipStream_t s);  // mvae_conv.hip (honours deferral)
bool p3_take_coljobs(ColJobs* out);                                                          // mvae_conv.hip

static bool edge_geometry(int C, int H, int Wd, int F) { return C == kEC && H == kEH && Wd == kEH && F == kEF; }

extern "C" int mvae_conv3_k4s2p1_nchw(const float* img, const float* W, const float* bias, const float* mask, int relu, float* y,
                                      uint16_t* y_planes, int64_t y_ps, int B, int C, int IH, int IW, int F, void* stream) {
  if (!img || !W || !y || B < 1) return fail(MVAE_E_BADARG, "null pointer / bad batch%s", "");
  if (!edge_geometry(C, IH, IW, F)) return fail(MVAE_E_UNSUPPORTED, "direct boundary convolution: 3 x 32 x 32 to 64 features%s", "");
  if (!aligned16(y) || (bias && !aligned16(bias)) || (mask && !aligned16(mask)) || (y_planes && (((uintptr_t)y_planes & 7) || (y_ps & 3))))
    return fail(MVAE_E_ALIGN, "direct boundary convolution: 16-byte aligned result / bias / mask, 8-byte aligned planes%s", "");
  const int nrows = B * kEO;
  // one row of 16 pixels per wave, 8 waves per workgroup (measured against 2 rows / 4, 8 or 16 waves: 13.5 / 14.7 us with
  // planes against 14.0-14.7 / 15.2-17.2 us; the W staging is shared by more waves, more waves hide the gather latency)
  constexpr int G = 1, NW = 8;
  hipLaunchKernelGGL((k_edge3_nt<G, NW>), dim3((unsigned)((nrows + NW * G - 1) / (NW * G))), dim3(64 * NW), 0,
                     (hipStream_t)stream, img, W, bias, mask, relu, y, y_planes, (long long)y_ps, nrows);
  LAUNCH_CHECK("direct boundary convolution launch");
  return 0;
}

static int edge_img_per_wg(int B) { return (B + 255) / 256; }
extern "C" int64_t mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats(int B, int C, int IH, int IW, int F) {
  if (B < 1 || !edge_geometry(C, IH, IW, F)) return 0;
  const int ipw = edge_img_per_wg(B);
  return (int64_t)((B + ipw - 1) / ipw) * kEF * kEK;
}
extern "C" int mvae_conv3_k4s2p1_nchw_wgrad(const float* act, const float* img, float* dW, int B, int C, int IH, int IW, int F,
                                            float* workspace, void* stream) {
  if (!act || !img || !dW || !workspace || B < 1) return fail(MVAE_E_BADARG, "null pointer / bad batch%s", "");
  if (!edge_geometry(C, IH, IW, F)) return fail(MVAE_E_UNSUPPORTED, "direct boundary weight gradient: 3 x 32 x 32, 64 features%s", "");
  if (!aligned16(act) || !aligned16(workspace) || !aligned16(dW))
    return fail(MVAE_E_ALIGN, "direct boundary weight gradient: 16-byte aligned operands%s", "");
  const int ipw = edge_img_per_wg(B), wgs = (B + ipw - 1) / ipw;
  // While sums are deferred, the column sums queued so far (the bias gradients of the backward pass: nothing this launch reads
  // or writes) ride along as extra workgroups instead of taking a launch of their own at the flush; their slice totals are
  // still added by the flush.
  ColJobs cols;
  static const bool no_ride = [] { const char* e = getenv("MVAE_COLSUM_RIDE"); return e && e[0] == '0'; }();
  if (!no_ride && p3_take_coljobs(&cols))
    hipLaunchKernelGGL(k_edge3_tn_cols, dim3((unsigned)(wgs + cols.blk0[cols.njobs])), dim3(512), 0, (hipStream_t)stream, act, img,
                       workspace, B, ipw, wgs, cols);
  else
    hipLaunchKernelGGL(k_edge3_tn, dim3((unsigned)wgs), dim3(512), 0, (hipStream_t)stream, act, img, workspace, B, ipw);
  p3_sum_slices(workspace, dW, (int64_t)kEF * kEK, wgs, (hipStream_t)stream);
  LAUNCH_CHECK("direct boundary weight gradient launch");
  return 0;
}

// mvae_conv3_k4s2p1_nchw_wgrad(act, img, dW) and mvae_conv3_k4s2p1_nchw(img, W, NULL, mask, 0, y, y_planes) in ONE launch: the
// backward pass of ConvTranspose2d(64, 3, 4, 2, 1) (conv_vae.py:54,74) -- img = the gradient of the logits, act = mask = the
// layer's input b2 (whose ReLU the backward-data passes through).
void p3_colsum_deferrable(const float* G, float* out, int64_t M, int N, float* ws, hipStream_t s);  // mvae_conv.hip

// floats of mvae_conv3_k4s2p1_nchw_backward's colsum_ws: the per-workgroup column sums + the column sum's own slice partials
extern "C" int64_t mvae_conv3_k4s2p1_nchw_backward_colsum_floats(int B) {
  if (B < 1) return 0;
  const int64_t wgs = ((int64_t)B * kEO + 7) / 8;
  return (wgs + (wgs + 511) / 512) * kEF;
}
extern "C" int mvae_conv3_k4s2p1_nchw_backward(const float* act, const float* img, const float* W, float* dW, float* y,
                                               uint16_t* y_planes, int64_t y_ps, float* colsum_out, float* colsum_ws, int B,
                                               int C, int IH, int IW, int F, float* workspace, void* stream) {
  if (!act || !img || !W || !dW || !workspace || B < 1) return fail(MVAE_E_BADARG, "null pointer / bad batch%s", "");
  if (!y && !(y_planes && colsum_out)) return fail(MVAE_E_BADARG, "y may only be NULL with planes + column sums%s", "");
  if ((colsum_out == nullptr) != (colsum_ws == nullptr)) return fail(MVAE_E_BADARG, "colsum_out and colsum_ws go together%s", "");
  if (!edge_geometry(C, IH, IW, F)) return fail(MVAE_E_UNSUPPORTED, "direct boundary backward: 3 x 32 x 32, 64 features%s", "");
  if (!aligned16(act) || !aligned16(workspace) || !aligned16(dW) || (y && !aligned16(y)) || (colsum_ws && !aligned16(colsum_ws)) ||
      (y_planes && (((uintptr_t)y_planes & 7) || (y_ps & 3))))
    return fail(MVAE_E_ALIGN, "direct boundary backward: 16-byte aligned operands, 8-byte aligned planes%s", "");
  const int ipw = edge_img_per_wg(B), wgs = (B + ipw - 1) / ipw, nrows = B * kEO, nt_wgs = (nrows + 7) / 8;
  hipLaunchKernelGGL(k_edge3_bwd, dim3((unsigned)(wgs + nt_wgs)), dim3(512), 0, (hipStream_t)stream, act, img, workspace, B, ipw,
                     wgs, W, act, y, y_planes, (long long)y_ps, nrows, colsum_ws);
  p3_sum_slices(workspace, dW, (int64_t)kEF * kEK, wgs, (hipStream_t)stream);
  if (colsum_out) p3_colsum_deferrable(colsum_ws, colsum_out, nt_wgs, kEF, colsum_ws + (size_t)nt_wgs * kEF, (hipStream_t)stream);
  LAUNCH_CHECK("direct boundary backward launch");
  return 0;
}
