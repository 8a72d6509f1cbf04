// mvae_conv.hip -- building blocks of the conv architecture (conv_vae.py:28-79), the LDS-tiled f32 MFMA contraction
// for large row counts, and the device-side input pipeline; the rest of the C ABI of include/mvae_hip.h.
#include "mvae_common.hpp"
#include "mvae_p3.hpp"

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
                                                   int relu, bf16r* dst_planes, int64_t dps) {
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
    if (dst_planes) store_planes4(dst_planes, dps, (size_t)o, acc.x, acc.y, acc.z, acc.w);
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
__device__ __forceinline__ void batch_stats_body(const float* bce, const float* kl, float* stats, float beta, int B,
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
  for (int r = tid; r < B && tid < 256; r += 256) {  // (the first four waves: callers may run with eight)
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
    for (int r = tid; r < B && tid < 256; r += 256) a += kl[(size_t)i * B + r];
    const float sres = block_sum(a);
    kt += sres;
    if (tid == 0) {
      kahan_add(stats, 4 + i, 2 * last, stats[4 + i], stats[2 * last + 4 + i], sres);
      stats[last + 4 + i] = sres;
    }
  }
  if (tid == 0) {
    kahan_add(stats, 0, 2 * last, stats[0], stats[2 * last], bs);
    kahan_add(stats, 1, 2 * last, stats[1], stats[2 * last + 1], kt);
    kahan_add(stats, 2, 2 * last, stats[2], stats[2 * last + 2], es);
    stats[3] += 1.f;
    stats[last] = bs; stats[last + 1] = kt; stats[last + 2] = es; stats[last + 3] = 1.f;
  }
}
// The tail of the loss end: d(bias of the last layer) = per-channel sums of the images' partial sums (rows in order: lane = row
// mod 64, then the DPP tree) and the batch statistics.  Run by the LAST workgroup of the loss-end launch (arrival counters), or
// -- while slice sums are deferred -- by one workgroup of the flush launch (loss_tail_deferred below).
__device__ __forceinline__ void loss_tail_body(const float* bce, const float* kl, float* stats, float beta, int B, int ncomp,
                                               const float* chan_part, float* dbias, int C) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    for (int c = 0; c < C; ++c) {
      float a = 0.f;
      for (int rr = tid; rr < B; rr += 64) a += chan_part[(size_t)rr * C + c];
      a = wave_sum(a);
      if (tid == 0) dbias[c] = a;
    }
  }
  batch_stats_body(bce, kl, stats, beta, B, ncomp);
}
__global__ __launch_bounds__(256) void k_batch_stats(const float* bce, const float* kl, float* stats, float beta, int B,
                                                     int ncomp) {
  batch_stats_body(bce, kl, stats, beta, B, ncomp);
}

// The loss end of the conv step in ONE launch (vae.py:125-147 + the bias gradient of the last ConvTranspose2d): workgroup r
// = k_bce_fwd_bwd for row r (NCHW logits: D = C x HW with HW a multiple of 1024, so every 1024-entry pass of the
// workgroup lies inside one channel) plus the row's per-channel sums of g; the LAST workgroup to finish (arrival counter)
// adds those sums over the rows in row order -> dbias[c] = sum_{b, y, x} g[b, c, y, x] -- and computes the batch statistics.
// Fixed orders everywhere: which workgroup arrives last does not change a bit.  (Separately: BCE, statistics, column sum
// of g, re-order, column sum = 5 launches.)
__global__ __launch_bounds__(256) void k_bce_stats(const float* logits, const float* x, float* bce, float* g,
                                                   const float* kl, float* stats, float beta, int B, int D, int HW,
                                                   int ncomp, float* chan_part, float* dbias, int* counter) {
  __shared__ float sm[4];
  __shared__ float chs[4][8];
  __shared__ int last_s;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int r = blockIdx.x, C = D / HW;
  const float* yl = logits + (size_t)r * D;
  const float* tl = x + (size_t)r * D;
  float* gl = g + (size_t)r * D;
  float s = 0.f;
  for (int c = 0; c < C; ++c) {
    float cs = 0.f;
    for (int j = c * HW + tid * 4; j < (c + 1) * HW; j += 1024) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(yl + j), t = *reinterpret_cast<const f32x4*>(tl + j);
      f32x4 gv;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float e = mvf::fexp(-fabsf(y[u]));
        gv[u] = ((y[u] >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e)) - t[u];
        s += (1.f - t[u]) * y[u] - (fminf(y[u], 0.f) - mvf::log1p_pos(e));
        cs += gv[u];
      }
      *reinterpret_cast<f32x4*>(gl + j) = gv;
    }
    cs = wave_sum(cs);
    if ((tid & 63) == 0) chs[wave][c] = cs;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) sm[wave] = s;
  __syncthreads();
  // The row's results leave as write-through stores and are waited for (vmcnt) before the arrival is counted; the last
  // workgroup acquires before reading everyone's.  (A device-scope release fence here -- __threadfence() -- writes back the
  // XCD's whole L2, dirty with 12 KB of g per row, once per workgroup: 28 us instead of 9.)
  if (tid == 0) store4_wt(bce, (size_t)r, (sm[0] + sm[1]) + (sm[2] + sm[3]));
  if (tid < C) store4_wt(chan_part, (size_t)r * C + tid, (chs[0][tid] + chs[1][tid]) + (chs[2][tid] + chs[3][tid]));
  if (!counter) return;  // the tail is queued with the deferred slice sums (loss_tail_deferred)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // two-level count (one hot word would serialise B device-scope atomics at ~12 ns each): 16 group words, then counter[16];
    // every word is re-armed (0) by the workgroup that completes it
    constexpr int NG = 16;
    const int grp = r % NG, gsize = (B - grp + NG - 1) / NG, ngroups = B < NG ? B : NG;
    int last = 0;
    if (atomicAdd(&counter[grp], 1) == gsize - 1) {
      counter[grp] = 0;
      if (atomicAdd(&counter[NG], 1) == ngroups - 1) {
        counter[NG] = 0;
        last = 1;
      }
    }
    last_s = last;
  }
  __syncthreads();
  if (!last_s) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  loss_tail_body(bce, kl, stats, beta, B, ncomp, chan_part, dbias, C);
}

static int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

static bool planes_out_ok(const uint16_t* p, int64_t ps) { return !p || (((uintptr_t)p & 7) == 0 && (ps & 3) == 0); }

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
                                  int taps_major, uint16_t* dst_planes, int64_t dst_ps, void* stream) {
  if (!col || !dst || B < 1 || C < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (taps_major) {
    if (!taps_major_ok(col, dst, mask, C, sc, sb, sy, sx) || (bias && !aligned16(bias)))
      return fail(MVAE_E_ALIGN, "taps-major col2im needs a channel-last destination with C %% 4 == 0%s", "");
    if (!planes_out_ok(dst_planes, dst_ps)) return fail(MVAE_E_ALIGN, "col2im planes must be 8-byte aligned%s", "");
    hipLaunchKernelGGL(k_col2im_tm, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       col, bias, mask, dst, B, C, H, W, sb, sy, sx, relu, dst_planes, dst_ps);
    LAUNCH_CHECK("col2im launch");
    return 0;
  }
  if (dst_planes) return fail(MVAE_E_UNSUPPORTED, "col2im writes planes for taps-major (channel-last) tensors only%s", "");
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
// GATHER (implicit contraction for the k4 s2 p1 convolutions on channel-last tensors; the patch matrix is never written):
//   1: the A operand is im2col(src) -- row m = (b, oy, ox), column k = (ky, kx, c) taps-major -- fetched straight from
//      src[b, 2oy-1+ky, 2ox-1+kx, c] (zero outside the image).  Conv2d forward / ConvTranspose2d backward-data (A_KC, B_KC).
//   2: the B operand is im2col(src) with the roles k = row m, j = (ky, kx, c): the weight gradient dy^T im2col(src).
//   3: the TRANSPOSED convolution (ConvTranspose2d forward, Conv2d backward-data) without its [M, 16 C] product: the
//      output pixels of one parity class (oy % 2, ox % 2) = blockIdx.z receive exactly 4 taps each, ky = 1 - py + 2 ty,
//      iy = oy/2 + py - ty (ty = 0, 1; the same in x), so per class it is a contraction over K = (4 taps, C_in) whose A
//      operand is gathered from src (zero outside the image), whose B operand is the tap's column block of the weight
//      [C_in, (ky, kx, oc)], and whose rows are scattered to the class's pixels by the epilogue.
// A K step of 32 lies inside one tap (C % 32 == 0), so the tap of a step is uniform and a lane moves 16 contiguous bytes.
// (struct ConvGeom: mvae_p3.hpp)
bool f32pp_try(int form, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, bf16r* Cp,
               long long psc, const float* bias, const float* mask, int relu, int M, int N, int K, ConvGeom cg,
               hipStream_t s);  // mvae_f32pp.hip
template <int BM, int BN, int BK, int NW, bool A_KC, bool B_KC, int GATHER = 0>
__global__ __launch_bounds__(64 * NW) void k_gemm_tiled(const float* __restrict__ A, int64_t sai, int64_t sak,
                                                    const float* __restrict__ Bm, int64_t sbk, int64_t sbj,
                                                    float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                                    const float* __restrict__ mask, int relu, int M, int N, int K,
                                                    int k_per_slice, int64_t slice_stride, ConvGeom cg,
                                                    bf16r* __restrict__ Cp, int64_t psc) {
  // Cp != NULL: the result's bf16 PLANES (mvae_p3.hpp) are written next to it, plane stride psc, same ldc -- for the plane
  // contractions of the backward pass (mvae_p3.hip), so that nobody has to split the tensor again.
  constexpr int kGT_BK = BK, kGT_LD = BK + 8, KQ = BK / 4;
  __shared__ __attribute__((aligned(16))) float As[BM * kGT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * kGT_LD];
  constexpr int NT = 64 * NW, WCOLS = NW / 2;  // waves as 2 x WCOLS
  constexpr int WM = BM / 2, WN = BN / WCOLS, TM = WM / 16, TN = WN / 16;
  constexpr int LA = BM * KQ / NT, LB = BN * KQ / NT;  // 16-byte vectors per thread per K step
  static_assert(LA >= 1 && LB >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kb = GATHER == 3 ? 0 : blockIdx.z * k_per_slice;
  const int ke = GATHER == 3 ? K : ((kb + k_per_slice < K) ? kb + k_per_slice : K);
  if (GATHER != 3) C += (size_t)blockIdx.z * slice_stride;
  const int par_y = GATHER == 3 ? (int)(blockIdx.z >> 1) : 0, par_x = GATHER == 3 ? (int)(blockIdx.z & 1) : 0;

  float4 ra[LA], rb[LB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < LA; ++r) {
      const int f = tid + NT * r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (A_KC && GATHER == 1) {  // row = output pixel, k = (tap, c): the tap of this K step is uniform
        const int i = f / KQ, k = k0 + ((f % KQ) << 2);
        const int m = m0 + i;
        const int tap = k0 / cg.Cc, c = k - tap * cg.Cc;
        const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = m >> cg.lOHW;
        const int iy = 2 * oy - 1 + (tap >> 2), ix = 2 * ox - 1 + (tap & 3);
        if (m < M && k < ke && iy >= 0 && iy < cg.IH && ix >= 0 && ix < cg.IW)
          v = *reinterpret_cast<const float4*>(A + ((size_t)(b * cg.IH + iy) * cg.IW + ix) * cg.Cc + c);
      } else if (A_KC && GATHER == 3) {  // row = output pixel of this parity class, k = (tap, c)
        const int i = f / KQ, k = k0 + ((f % KQ) << 2);
        const int m = m0 + i;
        const int tap = k0 / cg.Cc, c = k - tap * cg.Cc;  // tap = 2 ty + tx, uniform for the K step
        const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = m >> cg.lOHW;
        const int iy = oy + par_y - (tap >> 1), ix = ox + par_x - (tap & 1);
        if (m < M && k < ke && iy >= 0 && iy < cg.IH && ix >= 0 && ix < cg.IW)
          v = *reinterpret_cast<const float4*>(A + ((size_t)(b * cg.IH + iy) * cg.IW + ix) * cg.Cc + c);
      } else if (A_KC) {  // 4 consecutive k of row i
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
      } else if (GATHER == 2) {  // k = output pixel (the contraction index), j = (tap, c)
        const int k = k0 + (f % BK), j = n0 + ((f / BK) << 2);
        const int tap = j / cg.Cc, c = j - tap * cg.Cc;
        const int ox = k & ((1 << cg.lOW) - 1), oy = (k & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = k >> cg.lOHW;
        const int iy = 2 * oy - 1 + (tap >> 2), ix = 2 * ox - 1 + (tap & 3);
        if (j < N && k < ke && iy >= 0 && iy < cg.IH && ix >= 0 && ix < cg.IW)
          v = *reinterpret_cast<const float4*>(Bm + ((size_t)(b * cg.IH + iy) * cg.IW + ix) * cg.Cc + c);
      } else if (GATHER == 3) {  // row k = (tap, c) of the weight [C_in][(ky, kx, oc)]: the tap picks the column block
        const int k = k0 + (f % BK), j = (f / BK) << 2;
        const int tap = k0 / cg.Cc, c = k - tap * cg.Cc;
        const int ky = 1 - par_y + 2 * (tap >> 1), kx = 1 - par_x + 2 * (tap & 1);
        if (n0 + j < N && k < ke)
          v = *reinterpret_cast<const float4*>(Bm + (size_t)c * sbk + (size_t)(ky * 4 + kx) * N + (n0 + j));
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
    int m = m0 + wm + a * 16 + li;
    if (m >= M) continue;
    if (GATHER == 3) {  // row of the parity class -> its pixel of the (2 IH) x (2 IW) output
      const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, bb = m >> cg.lOHW;
      m = (bb * 2 * cg.IH + 2 * oy + par_y) * 2 * cg.IW + 2 * ox + par_x;
    }
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
        if (Cp) store_planes4(Cp, psc, (size_t)m * ldc + n, v[0], v[1], v[2], v[3]);
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

// ---- f32 contraction through SPLIT bf16 products.  On gfx950 the f32-input MFMA runs at the f32 vector rate, 1/16 of the
// bf16 MFMA (MI355X_MICROARCH.md: 157 vs 2500 TFLOP/s).  A float splits EXACTLY into three bf16 pieces by truncation,
// a = a_h + a_m + a_l (8 + 8 + 8 = the 24 significant bits), so a * b = the nine piece products; the six largest are
// kept -- lh, hl, mm, mh, hm, hh, each EXACT in the f32 accumulator's input (8 x 8-bit factors) and added in f32 from
// the smallest up -- and ml + lm + ll <= 2^-23 |a b| is dropped: about one more f32 rounding per product, against 8/3 of
// the MFMA rate (6 instructions of 16 cycles per 16 x 16 x 32 sub-product instead of 8 of 32).  Accumulation, bias,
// activation and storage stay f32.  tests/test_conv_gpu.py::test_split_product_contractions_vs_float64 holds this
// path to the same float64 bar as the f32-MFMA path.
// Tile BM x BN, K step 32 = one MFMA; both operands K-contiguous (NT: Linear / Conv2d forward, the gathered implicit
// forms of those).  Staging splits the 16-byte global vectors and stores each piece of an operand tile as [row][32 k]
// bf16 with a 96-byte row stride (conflict-free for the ds_read_b128 lane groups: a lane's fragment is the 8 consecutive
// k of its row, lane = (row, k octet)).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kB3LD = 24;  // dwords per LDS row: 16 of data + 8 of padding
__device__ __forceinline__ void split3(const float4 v, u32x2* hi, u32x2* mi, u32x2* lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned int h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned int xb = __float_as_uint(x[e]);
    h[e] = xb;
    const float r1 = x[e] - __uint_as_float(xb & 0xffff0000u);
    const unsigned int r1b = __float_as_uint(r1);
    m[e] = r1b;
    l[e] = __float_as_uint(r1 - __uint_as_float(r1b & 0xffff0000u));
  }
  // two bf16 per dword: the upper halves of elements (0, 1) and (2, 3)
  *hi = u32x2{__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
  *mi = u32x2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
  *lo = u32x2{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
}
// 4 x 4 transpose across the four lanes of a quad (DPP quad_perm, no LDS): lane q of the quad holds row q (4 entries) before,
// column q after.  The TN forms load 4 consecutive i (or j) of one contraction row k per lane, lanes of a quad = 4 consecutive
// k: transposed, a lane holds 4 consecutive k of ONE output row -- what the [row][k] staging of k_gemm_b3 stores.
__device__ __forceinline__ float dpp_quad(float x, int ctrl_xor1) {
  const int v = __float_as_int(x);
  return __int_as_float(ctrl_xor1 ? __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false)    // quad_perm [1,0,3,2]
                                  : __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float4 quad_transpose(float4 v, int lane) {
  const bool hi2 = lane & 2, hi1 = lane & 1;
  // distance 2: exchange the off-diagonal 2 x 2 blocks
  const float a = dpp_quad(hi2 ? v.x : v.z, 0), b = dpp_quad(hi2 ? v.y : v.w, 0);
  if (hi2) { v.x = a; v.y = b; } else { v.z = a; v.w = b; }
  // distance 1: transpose inside the 2 x 2 blocks
  const float c = dpp_quad(hi1 ? v.x : v.y, 1), d = dpp_quad(hi1 ? v.z : v.w, 1);
  if (hi1) { v.x = c; v.z = d; } else { v.y = c; v.w = d; }
  return v;
}
template <int BM, int BN, int NW, int GATHER, bool TA = false, bool TB = false, bool PIPE = false>
__global__ __launch_bounds__(64 * NW) void k_gemm_b3(const float* __restrict__ A, int64_t sai,
                                                 const float* __restrict__ Bm, int64_t sbj, float* __restrict__ C,
                                                 int64_t ldc, const float* __restrict__ bias,
                                                 const float* __restrict__ mask, int relu, int M, int N, int K,
                                                 int k_per_slice, int64_t slice_stride, ConvGeom cg) {
  // TA / TB = the operand is stored with its OUTPUT index contiguous (A(i, k) = A[k * sai + i], B(k, j) = Bm[k * sbj + j]:
  // sai / sbj are then the strides of the contraction index) and goes through the transposing stage; otherwise the
  // contraction index is contiguous (A[i * sai + k], Bm[j * sbj + k]).  Forms: NT (Linear / Conv2d forward, GATHER 0 | 1),
  // NN (TB: ConvTranspose2d forward / backward-data products; GATHER 3: the transposed convolution per parity class,
  // blockIdx.z), TN (TA + TB: weight gradients; GATHER 2: against the implicit patch matrix).
  constexpr int BK = 32, KQ = BK / 4;
  constexpr int NBUF = PIPE ? 2 : 1;
  __shared__ __attribute__((aligned(16))) unsigned int As_[NBUF][3][BM * kB3LD];
  __shared__ __attribute__((aligned(16))) unsigned int Bs_[NBUF][3][BN * kB3LD];
  unsigned int (*As)[BM * kB3LD] = As_[0];
  unsigned int (*Bs)[BN * kB3LD] = Bs_[0];
  constexpr int NT = 64 * NW, WCOLS = NW / 2;
  constexpr int WM = BM / 2, WN = BN / WCOLS, TM = WM / 16, TN = WN / 16;
  constexpr int LA = BM * KQ / NT, LB = BN * KQ / NT;
  static_assert(LA >= 1 && LB >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kb = GATHER == 3 ? 0 : blockIdx.z * k_per_slice;
  const int ke = GATHER == 3 ? K : ((kb + k_per_slice < K) ? kb + k_per_slice : K);
  if (GATHER != 3) C += (size_t)blockIdx.z * slice_stride;
  const int par_y = GATHER == 3 ? (int)(blockIdx.z >> 1) : 0, par_x = GATHER == 3 ? (int)(blockIdx.z & 1) : 0;
  // Requests run TWO K steps ahead of their stage, in two register sets.  They are branch-free -- a lane outside the
  // tile / image / K range reads the operand's base address and is zeroed at the stage: with exec-masked requests inside
  // branches hipcc's wait-count model loses the number in flight at the join and waits for ALL of them (vmcnt(0)), which
  // silently turns two steps ahead into one.
  struct RegSet {
    float4 ra[LA], rb[LB];
    bool oka[LA], okb[LB];
  };
  RegSet rs0, rs1;
  auto fetch = [&](int k0, RegSet& rs) {
#pragma unroll
    for (int r = 0; r < LA; ++r) {
      const int f = tid + NT * r;
      bool ok;
      const float* p;
      if constexpr (TA) {  // lane = one contraction row k (32 consecutive lanes = the K step), 4 consecutive i
        const int k = k0 + (f % BK), i = (f / BK) << 2;
        ok = m0 + i < M && k < ke;
        p = A + (size_t)k * sai + (m0 + i);
      } else {
        const int i = f / KQ, k = k0 + ((f % KQ) << 2);
        const int m = m0 + i;
        if (GATHER == 1 || GATHER == 3) {  // row = output pixel, k = (tap, c): the tap of a K step is uniform
          const int tap = k0 / cg.Cc, c = k - tap * cg.Cc;
          const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = m >> cg.lOHW;
          const int iy = GATHER == 1 ? 2 * oy - 1 + (tap >> 2) : oy + par_y - (tap >> 1);
          const int ix = GATHER == 1 ? 2 * ox - 1 + (tap & 3) : ox + par_x - (tap & 1);
          ok = m < M && k < ke && iy >= 0 && iy < cg.IH && ix >= 0 && ix < cg.IW;
          p = A + ((size_t)(b * cg.IH + iy) * cg.IW + ix) * cg.Cc + c;
        } else {
          ok = m < M && k < ke;
          p = A + (size_t)m * sai + k;
        }
      }
      rs.ra[r] = *reinterpret_cast<const float4*>(ok ? p : A);
      rs.oka[r] = ok;
    }
#pragma unroll
    for (int r = 0; r < LB; ++r) {
      const int f = tid + NT * r;
      bool ok;
      const float* p;
      if constexpr (TB) {
        const int k = k0 + (f % BK), j = n0 + ((f / BK) << 2);
        if (GATHER == 2) {  // B = the patch matrix of a channel-last image: k = output pixel, j = (tap, c)
          const int tap = j / cg.Cc, c = j - tap * cg.Cc;
          const int ox = k & ((1 << cg.lOW) - 1), oy = (k & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = k >> cg.lOHW;
          const int iy = 2 * oy - 1 + (tap >> 2), ix = 2 * ox - 1 + (tap & 3);
          ok = j < N && k < ke && iy >= 0 && iy < cg.IH && ix >= 0 && ix < cg.IW;
          p = Bm + ((size_t)(b * cg.IH + iy) * cg.IW + ix) * cg.Cc + c;
        } else if (GATHER == 3) {  // row k = (tap, c) of the weight [C_in][(ky, kx, oc)]: the tap picks the column block
          const int tap = k0 / cg.Cc, c = k - tap * cg.Cc;
          const int ky = 1 - par_y + 2 * (tap >> 1), kx = 1 - par_x + 2 * (tap & 1);
          ok = j < N && k < ke;
          p = Bm + (size_t)c * sbj + (size_t)(ky * 4 + kx) * N + j;
        } else {
          ok = j < N && k < ke;
          p = Bm + (size_t)k * sbj + j;
        }
      } else {
        const int j = f / KQ, k = k0 + ((f % KQ) << 2);
        ok = n0 + j < N && k < ke;
        p = Bm + (size_t)(n0 + j) * sbj + k;
      }
      rs.rb[r] = *reinterpret_cast<const float4*>(ok ? p : Bm);
      rs.okb[r] = ok;
    }
  };
  // stage: split every float into its three bf16 pieces and store 4 consecutive k of one tile row (8 bytes per piece).  A
  // transposed operand first turns its quad's 4 x 4 block (4 rows k x 4 entries i) in registers, after which this lane holds
  // k = 4 (kl / 4) .. + 3 of tile row i + (lane & 3).
  auto stage = [&](const RegSet& rs) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < LA; ++r) {
      const int f = tid + NT * r;
      float4 v = rs.oka[r] ? rs.ra[r] : zero4;
      int o;
      if constexpr (TA) {
        v = quad_transpose(v, lane);
        o = (((f / BK) << 2) + (lane & 3)) * kB3LD + (((f % BK) >> 2) << 1);
      } else {
        o = (f / KQ) * kB3LD + ((f % KQ) << 1);
      }
      u32x2 h, m, l;
      split3(v, &h, &m, &l);
      *reinterpret_cast<u32x2*>(&As[0][o]) = h;
      *reinterpret_cast<u32x2*>(&As[1][o]) = m;
      *reinterpret_cast<u32x2*>(&As[2][o]) = l;
    }
#pragma unroll
    for (int r = 0; r < LB; ++r) {
      const int f = tid + NT * r;
      float4 v = rs.okb[r] ? rs.rb[r] : zero4;
      int o;
      if constexpr (TB) {
        v = quad_transpose(v, lane);
        o = (((f / BK) << 2) + (lane & 3)) * kB3LD + (((f % BK) >> 2) << 1);
      } else {
        o = (f / KQ) * kB3LD + ((f % KQ) << 1);
      }
      u32x2 h, m, l;
      split3(v, &h, &m, &l);
      *reinterpret_cast<u32x2*>(&Bs[0][o]) = h;
      *reinterpret_cast<u32x2*>(&Bs[1][o]) = m;
      *reinterpret_cast<u32x2*>(&Bs[2][o]) = l;
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = (wave / WCOLS) * WM, wn = (wave % WCOLS) * WN;
  const int li = lane & 15, lk = (lane >> 4) << 2;  // row of the fragment, dword offset of its k octet

  auto compute = [&]() {
    bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        af[a][q] = *reinterpret_cast<const bf16x8*>(&As[q][(wm + a * 16 + li) * kB3LD + lk]);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        bf[b][q] = *reinterpret_cast<const bf16x8*>(&Bs[q][(wn + b * 16 + li) * kB3LD + lk]);
    // piece pairs from the smallest products up; operands swapped (B fragment first) so that a lane's four accumulator
    // values are four consecutive columns of one output row, as in k_gemm_tiled
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[b][PB[t]], af[a][PA[t]], acc[a][b], 0, 0, 0);
  };
  fetch(kb, rs0);
  fetch(kb + BK, rs1);
  if constexpr (PIPE) {
    // Two LDS buffers, ONE barrier per K step: the stage of step k + 1 (split + stores) sits in the same block as the MFMAs of
    // step k, so hipcc interleaves its VALU work with them instead of serialising stage -> barrier -> reads -> MFMAs -> barrier.
    As = As_[0]; Bs = Bs_[0];
    stage(rs0);
    __syncthreads();
    fetch(kb + 2 * BK, rs0);
    __builtin_amdgcn_sched_barrier(0);
    for (int k0 = kb; k0 < ke; k0 += 2 * BK) {
      As = As_[0]; Bs = Bs_[0];
      compute();
      As = As_[1]; Bs = Bs_[1];
      stage(rs1);
      fetch(k0 + 3 * BK, rs1);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      compute();
      As = As_[0]; Bs = Bs_[0];
      stage(rs0);
      fetch(k0 + 4 * BK, rs0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int k0 = kb; k0 < ke; k0 += 2 * BK) {
      // Both halves are unconditional (an odd number of K steps runs one all-zero step): the number of requests in flight is
      // then the same on every path into the loop head and the waits stay counted.  sched_barrier: hipcc otherwise hoists the
      // OTHER set's zeroing selects above this half's requests and MFMAs, and with them the wait for that set.
      stage(rs0);
      __syncthreads();
      fetch(k0 + 2 * BK, rs0);
      compute();
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      stage(rs1);
      __syncthreads();
      fetch(k0 + 3 * BK, rs1);
      compute();
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const bool vec = (((uintptr_t)C | (uintptr_t)bias | (uintptr_t)mask) & 15) == 0 && (ldc & 3) == 0;
  const int lc = (lane >> 4) << 2;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    int m = m0 + wm + a * 16 + li;
    if (m >= M) continue;
    if (GATHER == 3) {  // row of the parity class -> its pixel of the (2 IH) x (2 IW) output
      const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, bb = m >> cg.lOHW;
      m = (bb * 2 * cg.IH + 2 * oy + par_y) * 2 * cg.IW + 2 * ox + par_x;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + wn + b * 16 + lc;
      if (n >= N) continue;
      f32x4 v = acc[a][b];
      if (vec && n + 3 < N) {
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (relu)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
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

#ifndef MV_WGRAD_TARGET_SPLIT
#define MV_WGRAD_TARGET_SPLIT 256
#endif
#ifndef MV_B3_MIN128G
#define MV_B3_MIN128G 256  // gathered forms: 128 x 128 tiles from this many workgroups (one per CU) on
#endif
#ifndef MV_B3_TN128
#define MV_B3_TN128 256
#endif
#ifndef MV_B3_MIN12864
#define MV_B3_MIN12864 256  // plain NT shapes with < 512 128 x 128 tiles take 128 x 64 tiles when that still gives one per CU
#endif
// Which multiply the LDS-tiled contractions take (mvae_set_contraction_mode):
//   0  the f32-input MFMA everywhere (v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered f32 accumulation);
//   1  split bf16 products everywhere a kernel exists (k_gemm_b3);
//   2  (DEFAULT) split products in the BACKWARD pass only: every forward contraction -- whose output decides a ReLU mask
//      (conv_vae.py:57-79) or is the logits -- stays on the exact f32 MFMA, so the masks and every forward value are
//      bit-identical to mode 0; backward-data and weight-gradient contractions (2/3 of the step's flops) take k_gemm_b3,
//      whose error against float64 is no larger than the f32 MFMA's.
// Every entry point states the pass it belongs to: the inherently backward ones (weight gradients, mvae_gemm_tn,
// mvae_linear_forward_masked) implicitly, the shared ones through their `pass` argument (MVAE_PASS_FORWARD / _BACKWARD).
// The mode is read ONCE per call (relaxed atomic); workspace sizes do not depend on it.
#include <atomic>
static std::atomic<int> g_contraction_mode{2};
extern "C" int mvae_set_contraction_mode(int mode) {
  const int old = g_contraction_mode.load(std::memory_order_relaxed);
  if (mode >= 0) g_contraction_mode.store(mode > 2 ? 2 : mode, std::memory_order_relaxed);  // (< 0: query only)
  return old;
}
static inline bool split_for(int pass) {
  const int m = g_contraction_mode.load(std::memory_order_relaxed);
  return m == 1 || (m == 2 && pass == MVAE_PASS_BACKWARD);
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
template <bool A_KC, bool B_KC, int GATHER = 0>
static void launch_gemm_tiled(const float* A, int64_t sai, int64_t sak, const float* Bm, int64_t sbk, int64_t sbj,
                              float* C, int64_t ldc, const float* bias, const float* mask, int relu, int M, int N,
                              int K, int slices, int k_per_slice, int64_t slice_stride, hipStream_t s, bool split,
                              ConvGeom cg = ConvGeom{0, 0, 0, 0, 0}, bf16r* Cp = nullptr, int64_t psc = 0) {
  if (Cp) split = false;  // planes come out of the f32-MFMA kernels' epilogue only (forward results, the small layers)
  // exact f32 products, whole tiles, no K slices: the ping-pong LDS-DMA kernel (mvae_f32pp.hip; bit-identical to k_gemm_tiled)
  if constexpr (A_KC && (GATHER == 0 || GATHER == 1 || GATHER == 3) && !(B_KC && GATHER == 3)) {
    if (!split && (slices == 1 || GATHER == 3) && (GATHER != 0 || sak == 1) && (B_KC ? sbk == 1 : sbj == 1)) {
      const int form = GATHER == 1 ? 1 : (GATHER == 3 ? 3 : (B_KC ? 0 : 2));
      if (f32pp_try(form, A, sai, Bm, B_KC ? sbj : sbk, C, ldc, Cp, psc, bias, mask, relu, M, N, K, cg, s)) return;
    }
  }
  // 128 x 128 tiles need >= ~2 workgroups per CU to hide their own latencies; below that 64 x 64 tiles (4x the
  // workgroups, half the LDS reuse) win on every conv layer shape of the reference
  const int64_t wg128 = (int64_t)((N + 127) / 128) * ((M + 127) / 128) * slices;
  if constexpr (!A_KC && !B_KC && (GATHER == 0 || GATHER == 2)) {
    // TN (weight gradients): the same kernel with a transposing stage; 64 x 64 tiles, slices as chosen by the caller
    if (split && N > 64 && M > 64 && sai == 1 && (sak & 3) == 0 && (GATHER == 2 || (sbj == 1 && (sbk & 3) == 0))) {
#if MV_B3_TN128
      if (wg128 >= MV_B3_TN128) {
        dim3 grid((N + 127) / 128, (M + 127) / 128, slices);
        if (wg128 < 512)
          hipLaunchKernelGGL((k_gemm_b3<128, 128, 8, GATHER, true, true, true>), grid, dim3(512), 0, s, A, sak, Bm, sbk, C, ldc,
                             bias, mask, relu, M, N, K, k_per_slice, slice_stride, cg);
        else
          hipLaunchKernelGGL((k_gemm_b3<128, 128, 8, GATHER, true, true>), grid, dim3(512), 0, s, A, sak, Bm, sbk, C, ldc, bias,
                             mask, relu, M, N, K, k_per_slice, slice_stride, cg);
        return;
      }
#endif
      dim3 grid((N + 63) / 64, (M + 63) / 64, slices);
      hipLaunchKernelGGL((k_gemm_b3<64, 64, 8, GATHER, true, true, true>), grid, dim3(512), 0, s, A, sak, Bm, sbk, C, ldc, bias, mask,
                         relu, M, N, K, k_per_slice, slice_stride, cg);
      return;
    }
  }
  if constexpr (A_KC && !B_KC && (GATHER == 0 || GATHER == 3)) {
    // NN (B = a weight stored [K][N]) and the transposed convolution per parity class: transposing stage for B only
    if (split && N > 64 && (GATHER == 3 || sak == 1) && sbj == 1 && (sbk & 3) == 0) {
      const int zdim = GATHER == 3 ? 4 : slices;
      if (wg128 >= (GATHER == 0 ? 512 : MV_B3_MIN128G)) {
        dim3 grid((N + 127) / 128, (M + 127) / 128, zdim);
        // (two LDS buffers = 144 KB: only where there is one workgroup per CU anyway)
        if (wg128 < 512)
          hipLaunchKernelGGL((k_gemm_b3<128, 128, 8, GATHER, false, true, true>), grid, dim3(512), 0, s, A, sai, Bm, sbk, C, ldc,
                             bias, mask, relu, M, N, K, k_per_slice, slice_stride, cg);
        else
          hipLaunchKernelGGL((k_gemm_b3<128, 128, 8, GATHER, false, true>), grid, dim3(512), 0, s, A, sai, Bm, sbk, C, ldc, bias,
                             mask, relu, M, N, K, k_per_slice, slice_stride, cg);
      } else {
        dim3 grid((N + 63) / 64, (M + 63) / 64, zdim);
        hipLaunchKernelGGL((k_gemm_b3<64, 64, 8, GATHER, false, true, true>), grid, dim3(512), 0, s, A, sai, Bm, sbk, C, ldc, bias,
                           mask, relu, M, N, K, k_per_slice, slice_stride, cg);
      }
      return;
    }
  }
  if constexpr (A_KC && B_KC && (GATHER == 0 || GATHER == 1)) {
    if (split && N > 64 && (GATHER == 1 || sak == 1) && sbk == 1) {
      const int64_t wg12864 = (int64_t)((N + 63) / 64) * ((M + 127) / 128) * slices;
      const int min128 = GATHER == 0 ? 512 : MV_B3_MIN128G;
      if (GATHER == 0 && wg128 < 512 && wg12864 >= MV_B3_MIN12864) {
        dim3 grid((N + 63) / 64, (M + 127) / 128, slices);
        hipLaunchKernelGGL((k_gemm_b3<128, 64, 8, GATHER, false, false, true>), grid, dim3(512), 0, s, A, sai, Bm, sbj, C, ldc, bias, mask, relu, M, N,
                           K, k_per_slice, slice_stride, cg);
      } else if (wg128 < min128) {
        dim3 grid((N + 63) / 64, (M + 63) / 64, slices);
        hipLaunchKernelGGL((k_gemm_b3<64, 64, 8, GATHER, false, false, true>), grid, dim3(512), 0, s, A, sai, Bm, sbj, C, ldc, bias, mask, relu, M, N,
                           K, k_per_slice, slice_stride, cg);
      } else {
        dim3 grid((N + 127) / 128, (M + 127) / 128, slices);
        if (wg128 < 512)
          hipLaunchKernelGGL((k_gemm_b3<128, 128, 8, GATHER, false, false, true>), grid, dim3(512), 0, s, A, sai, Bm, sbj, C, ldc,
                             bias, mask, relu, M, N, K, k_per_slice, slice_stride, cg);
        else
          hipLaunchKernelGGL((k_gemm_b3<128, 128, 8, GATHER>), grid, dim3(512), 0, s, A, sai, Bm, sbj, C, ldc, bias, mask, relu,
                             M, N, K, k_per_slice, slice_stride, cg);
      }
      return;
    }
  }
  if (N > 64 && wg128 < 512) {
    dim3 grid((N + 63) / 64, (M + 63) / 64, slices);
    hipLaunchKernelGGL((k_gemm_tiled<64, 64, kBK64, kNW64, A_KC, B_KC, GATHER>), grid, dim3(64 * kNW64), 0, s, A, sai, sak, Bm, sbk, sbj, C, ldc,
                       bias, mask, relu, M, N, K, k_per_slice, slice_stride, cg, Cp, psc);
  } else if (N > 64) {
    dim3 grid((N + 127) / 128, (M + 127) / 128, slices);
    hipLaunchKernelGGL((k_gemm_tiled<128, 128, kBK128, kNW128, A_KC, B_KC, GATHER>), grid, dim3(64 * kNW128), 0, s, A, sai, sak, Bm, sbk, sbj, C, ldc,
                       bias, mask, relu, M, N, K, k_per_slice, slice_stride, cg, Cp, psc);
  } else {
    dim3 grid((N + 63) / 64, (M + 127) / 128, slices);
    hipLaunchKernelGGL((k_gemm_tiled<128, 64, kBK128, 4, A_KC, B_KC, GATHER>), grid, dim3(256), 0, s, A, sai, sak, Bm, sbk, sbj, C, ldc,
                       bias, mask, relu, M, N, K, k_per_slice, slice_stride, cg, Cp, psc);
  }
}

extern "C" int mvae_linear_forward_masked(const float* x, const float* W, const float* mask, float* y, int64_t M, int N,
                                          int K, uint16_t* y_planes, int64_t y_ps, void* stream) {
  if (!x || !W || !mask || !y || M < 1 || N < 1 || K < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (!tiled_ok(x, K) || !tiled_ok(W, K) || !tiled_ok(y, N) || !tiled_ok(mask, N) || M > 0x7fffffff || !planes_out_ok(y_planes, y_ps))
    return fail(MVAE_E_ALIGN, "masked linear needs 16-byte aligned operands with K, N multiples of 4%s", "");
  launch_gemm_tiled<true, true>(x, K, 1, W, 1, K, y, N, nullptr, mask, 0, (int)M, N, K, 1, (K + 15) & ~15, 0,
                                (hipStream_t)stream, split_for(MVAE_PASS_BACKWARD), ConvGeom{0, 0, 0, 0, 0}, y_planes,
                                y_ps);  // a Linear backward-data
  LAUNCH_CHECK("masked linear launch");
  return 0;
}

bool linear_forward_tiled(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K,
                                 int relu, hipStream_t s) {
  if (!tiled_ok(x, K) || !tiled_ok(W, K) || M > 0x7fffffff) return false;
  launch_gemm_tiled<true, true>(x, K, 1, W, 1, K, y, N, b, nullptr, relu, (int)M, N, K, 1, (K + 15) & ~15, 0, s,
                                split_for(MVAE_PASS_FORWARD));
  return true;
}

// mvae_linear_forward on the LDS-tiled kernel with the result's planes written by the epilogue (the first conv layer of
// conv_vae.py:47 as patch matrix x weight: its output is the gathered operand of the next layer's weight gradient)
extern "C" int mvae_linear_forward_planes(const float* x, const float* W, const float* b, float* y, uint16_t* y_planes,
                                          int64_t y_ps, int64_t M, int N, int K, int relu, void* stream) {
  if (!x || !W || !y || !y_planes || M < 1 || N < 1 || K < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (!tiled_ok(x, K) || !tiled_ok(W, K) || !tiled_ok(y, N) || (b && !aligned16(b)) || M > 0x7fffffff ||
      !planes_out_ok(y_planes, y_ps))
    return fail(MVAE_E_ALIGN, "mvae_linear_forward_planes needs 16-byte aligned operands with K, N multiples of 4%s", "");
  launch_gemm_tiled<true, true>(x, K, 1, W, 1, K, y, N, b, nullptr, relu, (int)M, N, K, 1, (K + 15) & ~15, 0,
                                (hipStream_t)stream, false, ConvGeom{0, 0, 0, 0, 0}, y_planes, y_ps);
  LAUNCH_CHECK("linear forward (planes) launch");
  return 0;
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
// Deferred slice sums: between mvae_slice_sums_defer(1) and mvae_slice_sums_flush() the producers below (weight
// gradients, tall column sums) only write their slices and QUEUE the final "add the slices in index order"; the flush
// performs every queued sum in ONE launch.  In the conv step these sums are ~13 launches of ~5 us each whose outputs
// nobody reads before the optimizer.  Host-side state, per calling thread; nothing here synchronises.
constexpr int kMaxSumJobs = 24;
struct SumJobs {
  const float* part[kMaxSumJobs];
  float* out[kMaxSumJobs];
  long long n[kMaxSumJobs];
  int slices[kMaxSumJobs];
  int blk0[kMaxSumJobs + 1];
  int njobs;
  // the queued tail of the loss end (loss_tail_body), run by ONE extra workgroup of the flush launch; lt_stats == NULL: none
  const float* lt_bce;
  const float* lt_kl;
  float* lt_stats;
  const float* lt_chan;
  float* lt_dbias;
  float lt_beta;
  int lt_B, lt_ncomp, lt_C;
};
static thread_local SumJobs g_sums;
static thread_local bool g_defer = false;
static thread_local bool g_suspended = false;  // deferral paused (mvae_slice_sums_defer(2)): immediate sums, queue kept

// One workgroup owns 64 lanes x (4 | 1) consecutive outputs of one job; its four waves take the slices k = w, w + 4, ...
// (8 requests in flight per lane: 16-byte ones when the job's n is a multiple of 4 and its pointers are 16-byte aligned),
// the four partial sums meet in LDS and are added in wave order -- the same additions in the same order as k_sum_slices.
__global__ __launch_bounds__(256) void k_sum_slices_batched(SumJobs jobs) {
  __shared__ f32x4 sm[4][64];
  // the extra workgroup of a queued loss tail is dispatched FIRST: a chain of ~6 us of dependent round trips that must not start
  // when the last slice-sum workgroup does
  const int bx = (int)blockIdx.x - (jobs.lt_stats ? 1 : 0);
  if (bx < 0) {
    loss_tail_body(jobs.lt_bce, jobs.lt_kl, jobs.lt_stats, jobs.lt_beta, jobs.lt_B, jobs.lt_ncomp, jobs.lt_chan, jobs.lt_dbias,
                   jobs.lt_C);
    return;
  }
  int j = 0;
  while (j + 1 < jobs.njobs && bx >= jobs.blk0[j + 1]) ++j;  // uniform
  const float* part = jobs.part[j];
  float* out = jobs.out[j];
  const long long n = jobs.n[j];
  const int slices = jobs.slices[j];
  const int nblk = jobs.blk0[j + 1] - jobs.blk0[j];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool vec = (n & 3) == 0 && ((((uintptr_t)part) | ((uintptr_t)out)) & 15) == 0;  // uniform
  if (vec) {
    const long long n4 = n >> 2;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
    for (long long base = (long long)(bx - jobs.blk0[j]) * 64; base < n4; base += (long long)nblk * 64) {
      const long long i = base + lane;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      if (i < n4) {
        int k = w;
        for (; k + 28 < slices; k += 32) {
          f32x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = p4[(size_t)(k + 4 * u) * n4 + i];
#pragma unroll
          for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < slices; k += 4) s += p4[(size_t)k * n4 + i];
      }
      sm[w][lane] = s;
      __syncthreads();
      if (w == 0 && i < n4)
        reinterpret_cast<f32x4*>(out)[i] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
      __syncthreads();
    }
    return;
  }
  float* sms = reinterpret_cast<float*>(&sm[0][0]);  // [4][64] floats
  for (long long base = (long long)(bx - jobs.blk0[j]) * 64; base < n; base += (long long)nblk * 64) {
    const long long i = base + lane;
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
    sms[w * 64 + lane] = s;
    __syncthreads();
    if (w == 0 && i < n) out[i] = (sms[lane] + sms[64 + lane]) + (sms[128 + lane] + sms[192 + lane]);  // same order as k_sum_slices
    __syncthreads();
  }
}

// Deferred column sums of tall matrices (bias gradients of the conv layers: 5 per backward pass, each its own ~5 us launch
// of 256-512 workgroups before): queued like the slice sums and performed by ONE launch at the flush, ahead of the batched
// slice sums that add their slice totals.  Thread (row group g = tid >> 4, column quad c = tid & 15) adds rows g, g + 16,
// ... of its 512-row slice, 8 requests of 16 bytes in flight; the 16 row groups meet in LDS and are added in group order:
// per column the same additions in the same order as k_colsum_sliced.
static thread_local ColJobs g_cols;  // (ColJobs, kColSlice and colsum_batched_body: mvae_p3.hpp)

__global__ __launch_bounds__(256) void k_colsum_batched(ColJobs jobs) { colsum_batched_body(jobs, (int)blockIdx.x); }

// The queued column sums handed to a launch that can carry them as extra workgroups (the boundary layer's weight gradient,
// mvae_edge.hip: the last launch of the backward pass before the flush, independent of them) instead of a launch of their own.
bool p3_take_splitjobs(SplitJobs4* out);  // mvae_p3.hip
bool p3_take_coljobs(ColJobs* out) {
  if (!g_defer || g_cols.njobs == 0) return false;
  *out = g_cols;
  g_cols.njobs = 0;
  return true;
}

static void flush_sums(hipStream_t s) {
  if (g_cols.njobs > 0) {  // the slice totals the queued sums below add up
    hipLaunchKernelGGL(k_colsum_batched, dim3((unsigned)g_cols.blk0[g_cols.njobs]), dim3(256), 0, s, g_cols);
    g_cols.njobs = 0;
  }
  if (g_sums.njobs == 0 && !g_sums.lt_stats) return;
  if (g_sums.njobs == 0) g_sums.blk0[0] = 0;
  hipLaunchKernelGGL(k_sum_slices_batched, dim3((unsigned)(g_sums.blk0[g_sums.njobs] + (g_sums.lt_stats ? 1 : 0))), dim3(256), 0, s,
                     g_sums);
  g_sums.njobs = 0;
  g_sums.lt_stats = nullptr;
}

// The loss end's tail while deferral is on: nobody reads the statistics or d(bias) before the optimizer, and inside the
// loss-end launch the tail is five dependent memory round trips of ONE workgroup after all the others have finished (stores
// drained, two arrival atomics, acquire, the partial sums, the statistics: 8.7 of that launch's 19 us at B = 256).  Queued, it
// runs beside the slice sums.  Returns false when deferral is off (the caller keeps the arrival-counted tail).
static bool loss_tail_deferred(const float* bce, const float* kl, float* stats, float beta, int B, int ncomp,
                               const float* chan_part, float* dbias, int C, hipStream_t s) {
  static const bool off = [] { const char* e = getenv("MVAE_LOSS_TAIL_DEFER"); return e && e[0] == '0'; }();
  if (!g_defer || off) return false;
  if (g_sums.lt_stats) flush_sums(s);  // (a second loss end before the flush: the first one's tail goes now)
  g_sums.lt_bce = bce; g_sums.lt_kl = kl; g_sums.lt_stats = stats; g_sums.lt_chan = chan_part; g_sums.lt_dbias = dbias;
  g_sums.lt_beta = beta; g_sums.lt_B = B; g_sums.lt_ncomp = ncomp; g_sums.lt_C = C;
  return true;
}

// the final sum of `slices` partial results: now, or queued while deferral is on
static void sum_slices(const float* part, float* out, int64_t n, int slices, hipStream_t s) {
  if (!g_defer) {
    hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * n)), dim3(256), 0, s, part, out, n, slices);
    return;
  }
  if (g_sums.njobs == kMaxSumJobs) flush_sums(s);
  const int j = g_sums.njobs++;
  if (j == 0) g_sums.blk0[0] = 0;
  g_sums.part[j] = part;
  g_sums.out[j] = out;
  g_sums.n[j] = n;
  g_sums.slices[j] = slices;
  const bool vec = (n & 3) == 0 && ((((uintptr_t)part) | ((uintptr_t)out)) & 15) == 0;
  const long long want = vec ? (n / 4 + 63) / 64 : (n + 63) / 64;
  g_sums.blk0[j + 1] = g_sums.blk0[j] + (int)(want < 1024 ? want : 1024);
}

// the same for the plane contractions of mvae_p3.hip: deferrable (weight gradients: nobody reads them before the flush) and
// immediate (the split-K slices of a backward-data result, which the next launch reads)
void p3_sum_slices(const float* part, float* out, int64_t n, int slices, hipStream_t s) { sum_slices(part, out, n, slices, s); }
void p3_sum_slices_now(const float* part, float* out, int64_t n, int slices, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * n)), dim3(256), 0, s, part, out, n, slices, nullptr, 1, 0);
}

// on = 1: queue the final sums (a queue left behind by an aborted pass is dropped when deferral is switched on);
// on = 2: SUSPEND -- sums requested now are performed immediately, the queue is kept (for an intermediate result that is
//         read before the flush); on = 0: off, and anything still queued is DROPPED (the normal path has flushed; after an
//         exception the queued outputs / workspaces may be gone, so the stale jobs must not run with the next pass).
extern "C" int mvae_slice_sums_defer(int on) {
  if (on == 1) {
    if (!g_defer && !g_suspended) {
      g_sums.njobs = g_cols.njobs = 0;
      g_sums.lt_stats = nullptr;
    }
    g_defer = true;
    g_suspended = false;
  } else if (on == 2) {
    g_suspended = g_defer || g_suspended;
    g_defer = false;
  } else {
    g_defer = false;
    g_suspended = false;
    g_sums.njobs = g_cols.njobs = 0;
    g_sums.lt_stats = nullptr;
  }
  return 0;
}

extern "C" int mvae_slice_sums_flush(void* stream) {
  flush_sums((hipStream_t)stream);
  LAUNCH_CHECK("batched slice sums");
  return 0;
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
    int slices = (MV_WGRAD_TARGET_SPLIT + wg - 1) / wg;  // (the same in every contraction mode)
    const int max_slices = (int)((M + kTnSlice - 1) / kTnSlice);  // what mvae_gemm_tn_workspace_floats provides
    if (slices > max_slices) slices = max_slices;
    if (slices > 1 && !workspace) return fail(MVAE_E_BADARG, "mvae_gemm_tn needs a workspace for M > 256%s", "");
    const int kps = (int)((((M + slices - 1) / slices) + 15) & ~(int64_t)15);
    slices = (int)((M + kps - 1) / kps);
    const int64_t n = (int64_t)NP * NQ;
    launch_gemm_tiled<false, false>(P, 1, NP, Q, NQ, 1, slices > 1 ? workspace : out, NQ, nullptr, nullptr, 0, NP, NQ,
                                    (int)M, slices, kps, n, (hipStream_t)stream, split_for(MVAE_PASS_BACKWARD));
    if (slices > 1) sum_slices(workspace, out, n, slices, (hipStream_t)stream);
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
    sum_slices(workspace, out, n, slices, (hipStream_t)stream);
  }
  LAUNCH_CHECK("gemm_tn launch");
  return 0;
}

// ---- implicit contractions of the channel-last k4 s2 p1 convolutions (no patch matrix in memory)
static int conv_geom(ConvGeom* g, int B, int Cc, int IH, int IW) {
  constexpr int kStep = kBK64 > kBK128 ? kBK64 : kBK128;  // a K step must lie inside one tap
  if (B < 1 || Cc < kStep || (Cc % kStep) || IH < 2 || IW < 2 || (IH & (IH - 1)) || (IW & (IW - 1)))
    return fail(MVAE_E_UNSUPPORTED, "implicit conv needs C %% 32 == 0 and power-of-two extents%s (%lld)", "", Cc);
  int lOW = 0, lOH = 0;
  while ((1 << lOW) < IW / 2) ++lOW;
  while ((1 << lOH) < IH / 2) ++lOH;
  *g = ConvGeom{Cc, IH, IW, lOW, lOW + lOH};
  return 0;
}

// Split-K for the implicit forward contraction: a layer with few output tiles and a long patch axis (ConvTranspose d1
// backward-data at B = 256: 4096 x 128 outputs = 128 tiles of 64 x 64, K = 4096) leaves half of the chip idle; its K range
// is cut into up to 4 slices whose partial products are added in index order (+ bias, ReLU) by k_sum_slices.
static int conv_fwd_slices(int64_t M, int OC, int K, bool has_mask, int* kps) {
  *kps = K;
  if (has_mask || OC <= 64) return 1;
  const int64_t tiles = ((OC + 63) / 64) * ((M + 63) / 64);
  if (tiles >= 256 || K < 2048) return 1;
  int slices = (int)((256 + tiles - 1) / tiles);
  if (slices > 4) slices = 4;
  *kps = ((K + slices - 1) / slices + 31) & ~31;
  return (K + *kps - 1) / *kps;
}

extern "C" int64_t mvae_conv_k4s2p1_nhwc_workspace_floats(int B, int Cc, int IH, int IW, int OC, int has_mask) {
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  int kps;
  const int slices = conv_fwd_slices(M, OC, 16 * Cc, has_mask != 0, &kps);
  return slices > 1 ? (int64_t)slices * M * OC : 0;
}

extern "C" int mvae_conv_k4s2p1_nhwc(const float* src, const float* Wt, const float* bias, const float* mask, float* y,
                                     int B, int Cc, int IH, int IW, int OC, int relu, float* workspace, int pass,
                                     uint16_t* y_planes, int64_t y_ps, void* stream) {
  if (!src || !Wt || !y || OC < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  ConvGeom g;
  int rc = conv_geom(&g, B, Cc, IH, IW);
  if (rc) return rc;
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  const int K = 16 * Cc;
  if (!tiled_ok(src, Cc) || !tiled_ok(Wt, K) || !tiled_ok(y, OC) || (mask && !tiled_ok(mask, OC)) ||
      (bias && ((uintptr_t)bias & 15)) || M > 0x7fffffff || !planes_out_ok(y_planes, y_ps))
    return fail(MVAE_E_ALIGN, "implicit conv needs 16-byte aligned operands%s", "");
  int kps;
  const int slices = (workspace && !y_planes) ? conv_fwd_slices(M, OC, K, mask != nullptr, &kps) : 1;
  if (slices > 1) {
    if (!tiled_ok(workspace, OC)) return fail(MVAE_E_ALIGN, "implicit conv workspace must be 16-byte aligned%s", "");
    const int64_t n = M * OC;
    launch_gemm_tiled<true, true, 1>(src, 0, 0, Wt, 1, K, workspace, OC, nullptr, nullptr, 0, (int)M, OC, K, slices, kps,
                                     n, (hipStream_t)stream, split_for(pass), g);
    hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * n)), dim3(256), 0, (hipStream_t)stream, workspace, y, n, slices,
                       bias, OC, relu);
  } else {
    launch_gemm_tiled<true, true, 1>(src, 0, 0, Wt, 1, K, y, OC, bias, mask, relu, (int)M, OC, K, 1, K, 0,
                                     (hipStream_t)stream, split_for(pass), g, y_planes, y_ps);
  }
  LAUNCH_CHECK("implicit conv launch");
  return 0;
}

// Transposed convolution (k4 s2 p1, channel-last) as four implicit contractions, one per output parity class (GATHER 3):
// src[B, IH, IW, C] -> y[B, 2 IH, 2 IW, OC]; Wt[C, (ky, kx, oc)] = the ConvTranspose2d weight [C, OC, 4, 4] taps-major
// (or, for the backward-data of a Conv2d with weight [C, OC', 4, 4] stored [C][(ky, kx, oc')], that same matrix).
extern "C" int mvae_conv_transpose_k4s2p1_nhwc(const float* src, const float* Wt, const float* bias, const float* mask, float* y,
                                      int B, int Cc, int IH, int IW, int OC, int relu, int pass, uint16_t* y_planes,
                                      int64_t y_ps, void* stream) {
  if (!src || !Wt || !y || OC < 4 || (OC & 3)) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (B < 1 || Cc < 32 || (Cc & 31) || IH < 1 || IW < 1 || (IH & (IH - 1)) || (IW & (IW - 1)))
    return fail(MVAE_E_UNSUPPORTED, "implicit transposed conv needs C %% 32 == 0 and power-of-two extents%s (%lld)", "",
                Cc);
  int lOW = 0, lOH = 0;
  while ((1 << lOW) < IW) ++lOW;
  while ((1 << lOH) < IH) ++lOH;
  const ConvGeom g{Cc, IH, IW, lOW, lOW + lOH};  // rows of a parity class = (b, oy / 2, ox / 2): the INPUT extent
  const int64_t M = (int64_t)B * IH * IW;
  const int K = 4 * Cc;
  if (!tiled_ok(src, Cc) || !tiled_ok(Wt, 16 * OC) || !tiled_ok(y, OC) || (mask && !tiled_ok(mask, OC)) ||
      (bias && ((uintptr_t)bias & 15)) || 4 * M > 0x7fffffff || !planes_out_ok(y_planes, y_ps))
    return fail(MVAE_E_ALIGN, "implicit transposed conv needs 16-byte aligned operands%s", "");
  launch_gemm_tiled<true, false, 3>(src, 0, 0, Wt, (int64_t)16 * OC, 1, y, OC, bias, mask, relu, (int)M, OC, K, 4, K, 0,
                                    (hipStream_t)stream, split_for(pass), g, y_planes, y_ps);
  LAUNCH_CHECK("implicit transposed conv launch");
  return 0;
}

static int wgrad_slices(int64_t M, int NP, int NQ, int* kps) {
  const int wg = ((NP + 127) / 128) * ((NQ + (NQ > 64 ? 127 : 63)) / (NQ > 64 ? 128 : 64));
  int slices = (MV_WGRAD_TARGET_SPLIT + wg - 1) / wg;  // (the same in every contraction mode)
  const int max_slices = (int)((M + kTnSlice - 1) / kTnSlice);
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  *kps = (int)((((M + slices - 1) / slices) + 31) & ~(int64_t)31);
  return (int)((M + *kps - 1) / *kps);
}

extern "C" int64_t mvae_conv_k4s2p1_nhwc_wgrad_workspace_floats(int B, int Cc, int IH, int IW, int OC) {
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  int kps;
  const int slices = wgrad_slices(M, OC, 16 * Cc, &kps);
  return slices > 1 ? (int64_t)slices * OC * 16 * Cc : 0;
}

extern "C" int mvae_conv_k4s2p1_nhwc_wgrad(const float* dy, const float* src, float* dWt, int B, int Cc, int IH, int IW,
                                           int OC, float* workspace, void* stream) {
  if (!dy || !src || !dWt || OC < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  ConvGeom g;
  int rc = conv_geom(&g, B, Cc, IH, IW);
  if (rc) return rc;
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  const int NQ = 16 * Cc;
  if (!tiled_ok(dy, OC) || !tiled_ok(src, Cc) || !tiled_ok(dWt, NQ) || M > 0x7fffffff)
    return fail(MVAE_E_ALIGN, "implicit conv needs 16-byte aligned operands%s", "");
  int kps;
  const int slices = wgrad_slices(M, OC, NQ, &kps);
  if (slices > 1 && !workspace) return fail(MVAE_E_BADARG, "the weight gradient needs its workspace%s", "");
  const int64_t n = (int64_t)OC * NQ;
  launch_gemm_tiled<false, false, 2>(dy, 1, OC, src, 0, 0, slices > 1 ? workspace : dWt, NQ, nullptr, nullptr, 0, OC, NQ,
                                     (int)M, slices, kps, n, (hipStream_t)stream, split_for(MVAE_PASS_BACKWARD), g);
  if (slices > 1) sum_slices(workspace, dWt, n, slices, (hipStream_t)stream);
  LAUNCH_CHECK("implicit conv weight gradient launch");
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
                            int pass, void* stream) {
  if (!G || !W || !out || M < 1 || K < 1 || N < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (M >= kTiledMinRows && tiled_ok(G, K) && tiled_ok(W, N) && M <= 0x7fffffff) {
    launch_gemm_tiled<true, false>(G, K, 1, W, N, 1, out, N, nullptr, mask, 0, (int)M, N, K, 1, (K + 15) & ~15, 0,
                                   (hipStream_t)stream, split_for(pass));
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
                                (hipStream_t)stream, split_for(MVAE_PASS_FORWARD));
  hipLaunchKernelGGL(k_sum_slices, dim3(grid_for(4 * n)), dim3(256), 0, (hipStream_t)stream, workspace, y, n, slices, b,
                     N, relu);
  LAUNCH_CHECK("split-K linear forward launch");
  return 0;
}

// tall matrices (conv activations: up to 65536 rows): row slices of kColSlice are summed by separate workgroups, the
// slice totals are then added in index order
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
    if (g_defer && (N & 3) == 0 && ((((uintptr_t)G) | ((uintptr_t)workspace)) & 15) == 0 && M <= 0x7fffffff) {
      // queued: the caller keeps G (and the workspace) alive until the flush
      if (g_cols.njobs == kMaxColJobs) flush_sums((hipStream_t)stream);
      const int j = g_cols.njobs++;
      if (j == 0) g_cols.blk0[0] = 0;
      g_cols.G[j] = G;
      g_cols.part[j] = workspace;
      g_cols.M[j] = (int)M;
      g_cols.N[j] = N;
      g_cols.ncb[j] = (N + 63) / 64;
      g_cols.blk0[j + 1] = g_cols.blk0[j] + g_cols.ncb[j] * slices;
    } else {
      hipLaunchKernelGGL(k_colsum_sliced, dim3((unsigned)(ncb * slices)), dim3(256), 0, (hipStream_t)stream, G,
                         workspace, (int)M, N, ncb);
    }
    sum_slices(workspace, out, (int64_t)N, slices, (hipStream_t)stream);
  }
  LAUNCH_CHECK("colsum launch");
  return 0;
}

// mvae_colsum for a caller inside the library (the per-workgroup partial column sums of mvae_edge.hip): queued like the tall
// ones whenever deferral is on, whatever M is; ws: ceil(M / 512) * N floats.
void p3_colsum_deferrable(const float* G, float* out, int64_t M, int N, float* ws, hipStream_t s) {
  const int ncb = (N + kColsPerBlock - 1) / kColsPerBlock;
  const int slices = (int)((M + kColSlice - 1) / kColSlice);
  if (g_defer && (N & 3) == 0 && ((((uintptr_t)G) | ((uintptr_t)ws)) & 15) == 0) {
    if (g_cols.njobs == kMaxColJobs) flush_sums(s);
    const int j = g_cols.njobs++;
    if (j == 0) g_cols.blk0[0] = 0;
    g_cols.G[j] = G;
    g_cols.part[j] = ws;
    g_cols.M[j] = (int)M;
    g_cols.N[j] = N;
    g_cols.ncb[j] = (N + 63) / 64;
    g_cols.blk0[j + 1] = g_cols.blk0[j] + g_cols.ncb[j] * slices;
  } else {
    hipLaunchKernelGGL(k_colsum_sliced, dim3((unsigned)(ncb * slices)), dim3(256), 0, s, G, ws, (int)M, N, ncb);
  }
  sum_slices(ws, out, (int64_t)N, slices, s);
}

extern "C" int mvae_bce_forward_backward(const float* logits, const float* x, float* bce, float* g, int64_t rows,
                                         int D, void* stream) {
  if (!logits || !x || !bce || !g || rows < 1 || D < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  hipLaunchKernelGGL(k_bce_fwd_bwd, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, x,
                     bce, g, rows, D);
  LAUNCH_CHECK("bce launch");
  return 0;
}

extern "C" int mvae_conv_bce_stats(const float* logits, const float* x, float* bce, float* g, const float* kl,
                                   float* stats, float beta, int64_t B, int D, int HW, int ncomp, float* chan_part,
                                   float* dbias, int32_t* counter, void* stream) {
  if (!logits || !x || !bce || !g || !kl || !stats || !chan_part || !dbias || !counter || B < 1 || B > 0x7fffffff ||
      D < 1 || HW < 1 || ncomp < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (D % HW != 0 || D / HW > 8 || (HW & 1023) != 0)
    return fail(MVAE_E_UNSUPPORTED, "mvae_conv_bce_stats: D = C x HW with C <= 8 and HW a multiple of 1024%s", "");
  if (((((uintptr_t)logits) | ((uintptr_t)x) | ((uintptr_t)g)) & 15) != 0)
    return fail(MVAE_E_ALIGN, "mvae_conv_bce_stats needs 16-byte aligned logits / x / g%s", "");
  const bool queued = loss_tail_deferred(bce, kl, stats, beta, (int)B, ncomp, chan_part, dbias, D / HW, (hipStream_t)stream);
  hipLaunchKernelGGL(k_bce_stats, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, x, bce, g, kl, stats, beta,
                     (int)B, D, HW, ncomp, chan_part, dbias, queued ? nullptr : counter);
  LAUNCH_CHECK("bce + statistics launch");
  return 0;
}

extern "C" int mvae_batch_stats(const float* bce, const float* kl, float* stats, float beta, int B, int ncomp,
                                void* stream) {
  if (!bce || !kl || !stats || B < 1 || ncomp < 1) return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  hipLaunchKernelGGL(k_batch_stats, dim3(1), dim3(256), 0, (hipStream_t)stream, bce, kl, stats, beta, B, ncomp);
  LAUNCH_CHECK("batch stats launch");
  return 0;
}


// ------------------------------------------------------------------------------------------------ conv architecture: the last transposed convolution, direct
// conv_vae.py:54, ConvTranspose2d(64, 3, 4, 2, 1) to the NCHW logits: 0.4 GFLOP; as product + col2im it costs 26 us, all of it
// moving the 12.6 MB [B * 256, 48] product; the direct kernel reads the activation once (19.6 us).  Fixed geometry: 64 features
// to 3 x 32 x 32.  (The same was built for the Conv2d(3 -> 64) side -- forward, ConvT backward-data, both weight gradients -- and
// measured SLOWER than patch matrix + contraction (24 / 18 us against 27 / 12 with the patch matrix shared by two uses): removed.)
constexpr int kBC = 3, kBF = 64, kBH = 32, kBO = 16, kBK = kBC * 16;  // channels, features, image / feature-map extent, patch

// y[b, c, Y, X] = bias[c] + sum over the 4 (y, ky) x (x, kx) pairs with Y = 2 y - 1 + ky, X = 2 x - 1 + kx and the 64 input features
// of src[(b, y, x)][ic] W[ic][c * 16 + ky * 4 + kx]: ConvTranspose2d(64 -> 3) to NCHW.  Workgroup = (image, 8 output rows); wave =
// one parity class (Y % 2, X % 2), so that the taps -- the weight rows -- are uniform per wave (broadcast reads).
__global__ __launch_bounds__(256) void k_convT_to3_fwd(const float* __restrict__ src, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ y) {
  constexpr int LDP = kBF + 4;
  __shared__ __attribute__((aligned(16))) float ss[6 * kBO * LDP];   // source rows Y0/2 - 1 .. Y0/2 + 4
  __shared__ __attribute__((aligned(16))) float Wp[16][kBC][kBF];   // [ky * 4 + kx][c][ic]
  const int tid = threadIdx.x, b = blockIdx.x >> 2, Y0 = (blockIdx.x & 3) << 3, ybase = (Y0 >> 1) - 1;
  for (int e = tid; e < 6 * kBO * (kBF / 4); e += 256) {
    const int q = e & 15, pixl = e >> 4, r = pixl >> 4, xx = pixl & 15, yy = ybase + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (yy >= 0 && yy < kBO) v = *reinterpret_cast<const f32x4*>(src + (((size_t)b * kBO + yy) * kBO + xx) * kBF + 4 * q);
    *reinterpret_cast<f32x4*>(&ss[(r * kBO + xx) * LDP + 4 * q]) = v;
  }
  for (int e = tid; e < kBF * kBK; e += 256) {
    const int ic = e / kBK, k = e - ic * kBK;
    Wp[k & 15][k >> 4][ic] = W[e];
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, pyp = wave >> 1, pxp = wave & 1;
  const int Y = Y0 + 2 * (lane >> 4) + pyp, X = 2 * (lane & 15) + pxp;
  float acc[kBC];
#pragma unroll
  for (int c = 0; c < kBC; ++c) acc[c] = 0.f;
#pragma unroll 1
  for (int ty = 0; ty < 2; ++ty)
#pragma unroll 1
    for (int tx = 0; tx < 2; ++tx) {
      const int ky = ((pyp + 1) & 1) + 2 * ty, kx = ((pxp + 1) & 1) + 2 * tx;  // uniform per wave
      const int yy = (Y + 1 - ky) >> 1, xx = (X + 1 - kx) >> 1;
      const bool ok = yy >= 0 && yy < kBO && xx >= 0 && xx < kBO;
      const float* sp = &ss[((ok ? yy - ybase : 0) * kBO + (ok ? xx : 0)) * LDP];
#pragma unroll 4
      for (int q = 0; q < kBF / 4; ++q) {
        f32x4 sv = *reinterpret_cast<const f32x4*>(sp + 4 * q);
        if (!ok) sv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < kBC; ++c) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wp[ky * 4 + kx][c][4 * q]);
          acc[c] = fmaf(sv[0], wv[0], fmaf(sv[1], wv[1], fmaf(sv[2], wv[2], fmaf(sv[3], wv[3], acc[c]))));
        }
      }
    }
#pragma unroll
  for (int c = 0; c < kBC; ++c)
    y[(((size_t)b * kBC + c) * kBH + Y) * kBH + X] = acc[c] + (bias ? bias[c] : 0.f);
}

static bool boundary_geometry(int C, int H, int Wd, int F) { return C == kBC && H == kBH && Wd == kBH && F == kBF; }

extern "C" int mvae_convt_to3_k4s2p1_forward(const float* src, const float* W, const float* bias, float* y, int B, int F,
                                             int IH, int IW, int C, void* stream) {
  if (!src || !W || !y || B < 1) return fail(MVAE_E_BADARG, "null pointer / bad batch%s", "");
  if (!boundary_geometry(C, 2 * IH, 2 * IW, F)) return fail(MVAE_E_UNSUPPORTED, "direct boundary convolution: 64 features to 3 x 32 x 32%s", "");
  if ((((uintptr_t)src) & 15) != 0) return fail(MVAE_E_ALIGN, "direct boundary transposed convolution needs a 16-byte aligned src%s", "");
  hipLaunchKernelGGL(k_convT_to3_fwd, dim3((unsigned)B * 4), dim3(256), 0, (hipStream_t)stream, src, W, bias, y);
  LAUNCH_CHECK("direct boundary transposed convolution launch");
  return 0;
}

// The last decoder layer AND the loss end in one launch (conv_vae.py:54,74 + vae.py:125-147): workgroup b computes image b's
// logits on the matrix cores -- P[256 pixels][48 = (c, ky, kx)] = b2[b] [256 x 64] W [64 x 48], 768 v_mfma_f32_16x16x4_f32, kept
// in LDS --, folds the four taps of every output pixel (the col2im of the transposed convolution: no [B * 256, 48] product in
// memory), adds the bias, writes the logits and continues exactly as k_bce_stats (BCE, its gradient, the per-image sums, the
// arrival-counted batch statistics and d3.bias).  Replaces k_convT_to3_fwd (VALU dot products, 20 us) + k_bce_stats (14 us)
// in the training step; the contraction index runs in the order (16 j + 4 (lane >> 4) + e), j, e = 0..3, so that ONE 16-byte
// load per lane and j feeds four MFMA steps.  Fixed geometry as k_convT_to3_fwd.
constexpr int kPS = kBK + 1;  // LDS row stride of P (floats)
__global__ __launch_bounds__(512) void k_d3_bce_stats(const float* __restrict__ src, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ x,
                                                      float* __restrict__ logits, float* bce, float* g, const float* kl,
                                                      float* stats, float beta, int B, int ncomp, float* chan_part,
                                                      float* dbias, int* counter) {
  __shared__ float Ps[kBO * kBO * kPS];  // 50 KB
  __shared__ float sm[8];
  __shared__ float chs[8][8];
  __shared__ int last_s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int r = blockIdx.x;
  // W (12 KB) once per workgroup through LDS (96 lines) instead of 48 four-row gathers per lane in each of the eight waves
  // (~2300 line accesses); row stride 52 floats: the four k rows of a fragment fall into four different bank quarters
  constexpr int kWS = 52;
  __shared__ float sWd[kBF * kWS];
  for (int e = tid; e < kBF * kBK / 4; e += 512) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(W + 4 * e);
    const int k = (4 * e) / kBK, n = (4 * e) % kBK;
#pragma unroll
    for (int i = 0; i < 4; ++i) sWd[k * kWS + n + i] = v[i];
  }
  constexpr int GW = 2;  // 16-pixel groups per wave (8 waves: two per SIMD, one multiplies while the other loads / folds)
  f32x4 av[GW][4];
#pragma unroll
  for (int gI = 0; gI < GW; ++gI) {
    const int px = (wave * GW + gI) * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      av[gI][j] = *reinterpret_cast<const f32x4*>(src + ((size_t)r * (kBO * kBO) + px) * kBF + 16 * j + 4 * l4);
  }
  __syncthreads();
  float wf[kBC][16];
#pragma unroll
  for (int u = 0; u < kBC; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) wf[u][4 * j + e] = sWd[(16 * j + 4 * l4 + e) * kWS + 16 * u + l15];
#pragma unroll
  for (int gI = 0; gI < GW; ++gI) {
    f32x4 acc[kBC];
#pragma unroll
    for (int u = 0; u < kBC; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < kBC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gI][j][e], wf[u][4 * j + e], acc[u], 0, 0, 0);
    // lane holds pixels 4 l4 + rr of the group, column 16 u + l15
#pragma unroll
    for (int u = 0; u < kBC; ++u)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Ps[((wave * GW + gI) * 16 + 4 * l4 + rr) * kPS + 16 * u + l15] = acc[u][rr];
  }
  __syncthreads();
  constexpr int HW = kBH * kBH, D = kBC * HW;
  const float* tl = x + (size_t)r * D;
  float* gl = g + (size_t)r * D;
  float* ll = logits + (size_t)r * D;
  const int idx = tid * 2, Y = idx >> 5, X0 = idx & 31, ky0 = (Y + 1) & 1;  // two consecutive outputs per thread and channel
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kBC; ++c) {
    const float bc = bias ? bias[c] : 0.f;
    float y[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int X = X0 + u, kx0 = (X + 1) & 1;
      float a = 0.f;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        const int ky = ky0 + 2 * ty, yy = (Y + 1 - ky) >> 1;
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
          const int kx = kx0 + 2 * tx, xx = (X + 1 - kx) >> 1;
          if (yy >= 0 && yy < kBO && xx >= 0 && xx < kBO) a += Ps[(yy * kBO + xx) * kPS + c * 16 + ky * 4 + kx];
        }
      }
      y[u] = a + bc;
    }
    const float2 t2 = *reinterpret_cast<const float2*>(tl + c * HW + idx);
    const float t[2] = {t2.x, t2.y};
    float gv[2];
    float cs = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float e = mvf::fexp(-fabsf(y[u]));
      gv[u] = ((y[u] >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e)) - t[u];
      s += (1.f - t[u]) * y[u] - (fminf(y[u], 0.f) - mvf::log1p_pos(e));
      cs += gv[u];
    }
    *reinterpret_cast<float2*>(ll + c * HW + idx) = float2{y[0], y[1]};
    *reinterpret_cast<float2*>(gl + c * HW + idx) = float2{gv[0], gv[1]};
    cs = wave_sum(cs);
    if (lane == 0) chs[wave][c] = cs;
  }
  s = wave_sum(s);
  if (lane == 0) sm[wave] = s;
  __syncthreads();
  // (from here on: k_bce_stats, see there)
  if (tid == 0) store4_wt(bce, (size_t)r, ((sm[0] + sm[1]) + (sm[2] + sm[3])) + ((sm[4] + sm[5]) + (sm[6] + sm[7])));
  if (tid < kBC)
    store4_wt(chan_part, (size_t)r * kBC + tid, ((chs[0][tid] + chs[1][tid]) + (chs[2][tid] + chs[3][tid])) +
                                                    ((chs[4][tid] + chs[5][tid]) + (chs[6][tid] + chs[7][tid])));
  if (!counter) return;  // the tail is queued with the deferred slice sums (loss_tail_deferred)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    constexpr int NG = 16;
    const int grp = r % NG, gsize = (B - grp + NG - 1) / NG, ngroups = B < NG ? B : NG;
    int last = 0;
    if (atomicAdd(&counter[grp], 1) == gsize - 1) {
      counter[grp] = 0;
      if (atomicAdd(&counter[NG], 1) == ngroups - 1) {
        counter[NG] = 0;
        last = 1;
      }
    }
    last_s = last;
  }
  __syncthreads();
  if (!last_s) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  loss_tail_body(bce, kl, stats, beta, B, ncomp, chan_part, dbias, kBC);
}

extern "C" int mvae_convt_to3_bce_stats(const float* src, const float* W, const float* bias, const float* x, float* logits,
                                        float* bce, float* g, const float* kl, float* stats, float beta, int64_t B, int F,
                                        int IH, int IW, int C, int ncomp, float* chan_part, float* dbias, int32_t* counter,
                                        void* stream) {
  if (!src || !W || !x || !logits || !bce || !g || !kl || !stats || !chan_part || !dbias || !counter || B < 1 ||
      B > 0x7fffff || ncomp < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  if (!boundary_geometry(C, 2 * IH, 2 * IW, F))
    return fail(MVAE_E_UNSUPPORTED, "fused last layer + loss end: 64 features to 3 x 32 x 32%s", "");
  if (((((uintptr_t)src) | ((uintptr_t)x) | ((uintptr_t)logits) | ((uintptr_t)g)) & 15) != 0)
    return fail(MVAE_E_ALIGN, "mvae_convt_to3_bce_stats needs 16-byte aligned src / x / logits / g%s", "");
  const bool queued = loss_tail_deferred(bce, kl, stats, beta, (int)B, ncomp, chan_part, dbias, kBC, (hipStream_t)stream);
  hipLaunchKernelGGL(k_d3_bce_stats, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, src, W, bias, x, logits, bce, g, kl,
                     stats, beta, (int)B, ncomp, chan_part, dbias, queued ? nullptr : counter);
  LAUNCH_CHECK("fused last layer + loss end launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ conv architecture: the latent section in four launches
// conv_vae.py:65-71 between the last encoder convolution and the first decoder convolution:
//   h = a2.view(bs, -1)  (NCHW flatten: column c * 16 + p)  ->  fc_mean / fc_logvar of every component (component.py:52-57)
//   ->  rsample + KL (component.py:59-78)  ->  fc(z) + ReLU  ->  .view(-1, 128, 4, 4)
// and its backward.  Built from the generic operators this is 6 + 10 launches of ~5 us (re-order W_heads, split-K heads,
// slice sum, components, fc, re-order | re-order, ReLU mask, fc backward, memset, components, row sum, heads backward,
// re-order dW); here it is 2 + 2.  The activations on both sides are CHANNEL-LAST (a2: [B, 16, 512], t0: [B, 16, 128]) as
// the convolutions want them; the weights keep the reference's orders (W_heads columns c * 16 + p, W_d0 rows c * 16 + p)
// and the kernels index them accordingly.  Shapes: heads_dim <= 16, z_dim <= 16, true dimensions <= 8
// (mvae_conv_latent_supported); other models take the generic operators.
constexpr int kEncC = 512, kPix = 16, kDecC = 128;       // encoder output 512 x 4 x 4, decoder input 128 x 4 x 4
constexpr int kFlat = kEncC * kPix, kD0 = kDecC * kPix;  // 8192, 2048
constexpr int kClSlice = 128, kClSlices = kFlat / kClSlice;  // split-K slices of the heads: 128 channel-last columns each
constexpr int kClNN = 16;                                // row stride of the slice partials (heads_dim <= 16)

// sum over each 32-lane half of the wave: lanes 16..31 end up with the total of lanes 0..31, lanes 48..63 with that of
// lanes 32..63 (DPP row operations + one row broadcast, fixed order)
__device__ __forceinline__ float half_wave_sum(float v) {
  int x = __float_as_int(v);
#define MV_DPP_ADD(CTRL, ROWMASK)                                                                       \
  x = __float_as_int(__int_as_float(x) +                                                                \
                     __int_as_float(__builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xF, true)));
  MV_DPP_ADD(0xB1, 0xF)   // quad_perm [1,0,3,2]
  MV_DPP_ADD(0x4E, 0xF)   // quad_perm [2,3,0,1]
  MV_DPP_ADD(0x141, 0xF)  // row_half_mirror
  MV_DPP_ADD(0x140, 0xF)  // row_mirror: every lane of a 16-lane row holds the row sum
  MV_DPP_ADD(0x142, 0xA)  // row_bcast15 into rows 1 and 3
#undef MV_DPP_ADD
  return __int_as_float(x);
}

// launch 1 of the forward: part[s][r][n] = sum_{k in slice s} a2[r][k] W_heads[n][ref(k)].  Workgroup = (slice s = (pixel
// p, 128 channels), 64 rows); thread (row group g = tid >> 5, channel quad l = tid & 31) keeps its 4 x NN weights in
// registers and walks 8 rows, one 16-byte load each (all requested first); the 32 quads of a row meet by DPP.
template <int NN>
__global__ __launch_bounds__(256) void k_cl_heads_part(const float* __restrict__ a2, const float* __restrict__ W,
                                                       float* __restrict__ part, int B, int N) {
  // The workgroup's NN x 128 weights sit 64 bytes apart in W (the reference's column order c * 16 + p).  Fetched per thread
  // (4 x NN scalar loads, 32 different 128-byte lines per wave-level load) they cost 6144 line accesses per workgroup -- 7 us of
  // the CU's address path for 6 KB; staged once through LDS, 64 consecutive channels per wave-level load, 768.
  __shared__ __attribute__((aligned(16))) float sW[NN][128];
  const int tid = threadIdx.x, l = tid & 31, g = tid >> 5;
  const int s = blockIdx.x, p = s >> 2, c0 = (s & 3) << 7, c = c0 + (l << 2);
  // the activation rows are requested BEFORE the weights are staged: behind the staging barrier they were a second memory round
  // trip of a launch that is two round trips long
  const int r0 = blockIdx.y * 64;
  f32x4 xv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int r = r0 + g + 8 * u;
    xv[u] = *reinterpret_cast<const f32x4*>(a2 + (size_t)(r < B ? r : B - 1) * kFlat + p * kEncC + c);
  }
  constexpr int kWPer = NN * 128 / 256;
  float wst[kWPer];
#pragma unroll
  for (int q = 0; q < kWPer; ++q) {
    const int e = tid + 256 * q, n = e >> 7, cc = e & 127;
    wst[q] = W[(size_t)(n < N ? n : 0) * kFlat + (c0 + cc) * kPix + p];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < kWPer; ++q) {
    const int e = tid + 256 * q;
    sW[e >> 7][e & 127] = wst[q];
  }
  __syncthreads();
  float w[NN][4];
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(&sW[n][l << 2]);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[n][i] = wv[i];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int r = r0 + g + 8 * u;
    f32x4 o[NN / 4];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a = xv[u][0] * w[n][0];
      a = fmaf(xv[u][1], w[n][1], a);
      a = fmaf(xv[u][2], w[n][2], a);
      a = fmaf(xv[u][3], w[n][3], a);
      o[n >> 2][n & 3] = half_wave_sum(a);
    }
    if (l == 31 && r < B) {
      float* dst = part + ((size_t)s * B + r) * kClNN;
#pragma unroll
      for (int q = 0; q < NN / 4; ++q) *reinterpret_cast<f32x4*>(dst + 4 * q) = o[q];
    }
  }
}

// launch 2 of the forward, one workgroup per batch row: heads = bias + the slices in index order (16 groups of 4, then the
// groups in order); the components, each on the wave fill_table gave it (kinds do not share a wave); t0 = relu(z W_d0^T
// + b) written channel-last through LDS.
template <int DMAX>
__global__ __launch_bounds__(256) void k_cl_latent_fwd(CompTable t, const float* __restrict__ part,
                                                       const float* __restrict__ b_heads, int NH,
                                                       const float* __restrict__ eps, int eps_ld,
                                                       const float* __restrict__ radii, const float* __restrict__ W_d0,
                                                       const float* __restrict__ b_d0, int Z, float* __restrict__ heads,
                                                       float* __restrict__ z, float* __restrict__ kl,
                                                       float* __restrict__ t0, bf16r* __restrict__ t0p, long long t0ps,
                                                       int B, const SplitJobs4 ride) {
  // workgroups past the batch rows carry queued plane splits (the weight planes of the backward pass: independent of this launch,
  // which is one dependent chain per row and leaves the memory system idle)
  if ((int)blockIdx.x >= B) {
    split3_body(ride, (int)blockIdx.x - B, 256);
    return;
  }
  __shared__ float hp[16][17];
  __shared__ float heads_s[16], z_s[16];
  __shared__ float eps_s[16], rad_s[16];
  __shared__ float t0_s[kPix * (kDecC + 1)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = blockIdx.x;
  // the row's eps and the radii travel with the first requests and wait in LDS: read from global memory inside the component
  // chain they put a memory round trip (~2 us) on its critical path
  if (tid >= 64 && tid < 80) eps_s[tid - 64] = (tid - 64) < eps_ld ? eps[(size_t)r * eps_ld + (tid - 64)] : 0.f;
  if (tid >= 128 && tid < 144) rad_s[tid - 128] = (tid - 128) < t.n ? radii[tid - 128] : 0.f;
  // this thread's 8 rows of W_d0 are requested now and used after the component chain (Z = 8: the BASELINE model)
  f32x4 wq[8][2];
  float bq[8];
  if (Z == 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = tid + 256 * i;
      wq[i][0] = *reinterpret_cast<const f32x4*>(W_d0 + (size_t)n * 8);
      wq[i][1] = *reinterpret_cast<const f32x4*>(W_d0 + (size_t)n * 8 + 4);
      bq[i] = b_d0[n];
    }
  }
  {
    const int j = tid & 15, q = tid >> 4;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = part[((size_t)(q * 4 + i) * B + r) * kClNN + j];
    hp[q][j] = ((v[0] + v[1]) + v[2]) + v[3];
    __syncthreads();
    if (tid < 16) {
      float h = hp[0][tid];
#pragma unroll
      for (int qq = 1; qq < 16; ++qq) h += hp[qq][tid];
      if (tid < NH) {
        h += b_heads[tid];
        heads[(size_t)r * NH + tid] = h;
      }
      heads_s[tid] = tid < NH ? h : 0.f;
    }
    __syncthreads();
  }
#ifdef MV_CL_DBG
  if (!(MV_CL_DBG & 32))
#endif
  for (int ci = 0; ci < t.n; ++ci)
    if (t.wave_of[ci] == wave && t.lane_of[ci] == lane)
      comp_fwd_row<DMAX>(t.c[ci], heads_s, eps_s, rad_s, z_s, z + (size_t)r * Z, kl + (size_t)ci * B + r, nullptr, nullptr,
                         nullptr, nullptr);
  __syncthreads();
#ifdef MV_CL_DBG
  if (MV_CL_DBG & 64) return;
#endif
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = tid + 256 * i;
    float a = 0.f;
    if (Z == 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) a = fmaf(z_s[k], wq[i][0][k], a);
#pragma unroll
      for (int k = 0; k < 4; ++k) a = fmaf(z_s[4 + k], wq[i][1][k], a);
      a += bq[i];
    } else {
      for (int k = 0; k < Z; ++k) a = fmaf(z_s[k], W_d0[(size_t)n * Z + k], a);
      a += b_d0[n];
    }
    t0_s[(n & 15) * (kDecC + 1) + (n >> 4)] = a < 0.f ? 0.f : a;  // torch.relu: NaN propagates
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid * 4 + 1024 * i, pp = idx >> 7, cc = idx & 127;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t0_s[pp * (kDecC + 1) + cc + e];
    *reinterpret_cast<f32x4*>(t0 + (size_t)r * kD0 + idx) = v;
    if (t0p) store_planes4(t0p, t0ps, (size_t)r * kD0 + idx, v[0], v[1], v[2], v[3]);
  }
}

// launch 1 of the backward, one workgroup per batch row: dd0 = dt0 [t0 > 0] (kept channel-last for launch 2); dz = dd0
// W_d0 (thread = row n of W_d0, fixed-order workgroup sum); the components' derivative directions (component ci on its
// wave, lane = direction: dual numbers with the reference's derivative rules) -> dheads, per-row radius terms.
template <int DMAX>
__global__ __launch_bounds__(256) void k_cl_latent_bwd_rows(CompTable t, const float* __restrict__ heads, int NH,
                                                            const float* __restrict__ eps, int eps_ld,
                                                            const float* __restrict__ radii,
                                                            const float* __restrict__ W_d0, int Z,
                                                            const float* __restrict__ t0, const float* __restrict__ dt0,
                                                            const int dt0_slices, const long long dt0_stride,
                                                            float beta, float* __restrict__ dd0,
                                                            float* __restrict__ dheads, float* __restrict__ drad_rows,
                                                            int B) {
  __shared__ float dd_s[kPix * (kDecC + 1)];
  __shared__ float red_s[4][16];
  __shared__ float dz_s[16], heads_s[16], eps_s[16], rad_s[16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = blockIdx.x;
  // ---- every request of the row first: the chain's small operands (into LDS), W_d0, the dt0 slices and the mask
  float hv = 0.f, ev = 0.f, rv = 0.f;
  if (tid < 16) hv = tid < NH ? heads[(size_t)r * NH + tid] : 0.f;
  if (tid >= 64 && tid < 80) ev = (tid - 64) < eps_ld ? eps[(size_t)r * eps_ld + (tid - 64)] : 0.f;
  if (tid >= 128 && tid < 144) rv = (tid - 128) < t.n ? radii[tid - 128] : 0.f;
  f32x4 wq[8][2];  // this thread's 8 rows of W_d0 (Z = 8)
  if (Z == 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = tid + 256 * i;
      wq[i][0] = *reinterpret_cast<const f32x4*>(W_d0 + (size_t)n * 8);
      wq[i][1] = *reinterpret_cast<const f32x4*>(W_d0 + (size_t)n * 8 + 4);
    }
  }
  f32x4 dsl[2][4], m4v[2];  // up to four K slices of dt0 per 16-byte piece, held until after the dual chain
  const bool few = dt0_slices <= 4;  // uniform
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid * 4 + 1024 * i;
    m4v[i] = *reinterpret_cast<const f32x4*>(t0 + (size_t)r * kD0 + idx);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      dsl[i][w] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (few && w < (dt0_slices < 1 ? 1 : dt0_slices))
        dsl[i][w] = *reinterpret_cast<const f32x4*>(dt0 + (size_t)w * dt0_stride + (size_t)r * kD0 + idx);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (tid < 16) heads_s[tid] = hv;
  if (tid >= 64 && tid < 80) eps_s[tid - 64] = ev;
  if (tid >= 128 && tid < 144) rad_s[tid - 128] = rv;
  __syncthreads();
  // ---- the forward-mode dual chain of this lane's (component, input direction): needs no upstream gradient, so it runs
  // while the large requests above travel (after the dz reduction it was 6 of the kernel's 13.5 us, back to back with them)
  // (the (component, direction) pairs of a wave's components lie side by side on its lanes: heads_dim <= 16 keeps them under 64)
  float zd[DMAX + 2], kld = 0.f;
  int my_ci = -1, my_dir = 0, my_A = 0, my_zcol = 0, my_out = 0;  // my_out: column of dheads, or -1: the radius direction
#pragma unroll
  for (int i = 0; i < DMAX + 2; ++i) zd[i] = 0.f;
#ifdef MV_CL_DBG
  if (!(MV_CL_DBG & 8))
#endif
  {
    int off = 0;
    for (int ci = 0; ci < t.n; ++ci) {
      if (t.wave_of[ci] != wave) continue;  // uniform per wave
      const int ndir = t.dir_off[ci + 1] - t.dir_off[ci];
      if (lane >= off && lane < off + ndir) {
        const mvae_component_desc& c = t.c[ci];
        my_ci = ci;
        my_dir = lane - off;
        my_A = ambient_dim(c.kind, c.true_dim);
        my_zcol = c.z_col;
        my_out = my_dir < c.true_dim ? c.mean_col + my_dir
                                     : (my_dir < c.true_dim + c.logvar_dim ? c.logvar_col + (my_dir - c.true_dim) : -1);
        kld = comp_dual_dir<DMAX>(c, heads_s, eps_s, rad_s, my_dir, zd);
      }
      off += ndir;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- dd0 = dt0 * [t0 > 0] (dt0 = the sum of its K slices, in k_sum_slices' order: four partial sums over the slices
  // k = w, w + 4, ..., then (p0 + p1) + (p2 + p3)), kept in LDS for the dz contraction
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid * 4 + 1024 * i, pp = idx >> 7, cc = idx & 127;
    f32x4 d4;
    if (few) {
      d4 = dt0_slices <= 1 ? dsl[i][0] : (dsl[i][0] + dsl[i][1]) + (dsl[i][2] + dsl[i][3]);
    } else {
      f32x4 ps[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        ps[w] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = w; k < dt0_slices; k += 4) ps[w] += *reinterpret_cast<const f32x4*>(dt0 + (size_t)k * dt0_stride + (size_t)r * kD0 + idx);
      }
      d4 = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    }
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = (m4v[i][e] > 0.f) ? d4[e] : 0.f;
      dd_s[pp * (kDecC + 1) + cc + e] = v[e];
    }
    *reinterpret_cast<f32x4*>(dd0 + (size_t)r * kD0 + idx) = v;
  }
  __syncthreads();
  float dzp[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dzp[k] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = tid + 256 * i;
    const float dd = dd_s[(n & 15) * (kDecC + 1) + (n >> 4)];
    if (Z == 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dzp[k] = fmaf(dd, wq[i][0][k], dzp[k]);
        dzp[4 + k] = fmaf(dd, wq[i][1][k], dzp[4 + k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < Z) dzp[k] = fmaf(dd, W_d0[(size_t)n * Z + k], dzp[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k < Z) {  // uniform
      const float sres = wave_sum(dzp[k]);
      if (lane == 0) red_s[wave][k] = sres;
    }
  }
  __syncthreads();
  if (tid < Z) dz_s[tid] = (red_s[0][tid] + red_s[1][tid]) + (red_s[2][tid] + red_s[3][tid]);
  __syncthreads();
  // ---- d(loss) / d(direction) = beta * d kl + <dz, d z>   (comp_bwd_dir's sum)
  if (my_ci >= 0) {
    float gval = beta * kld;
#pragma unroll
    for (int i = 0; i < DMAX + 1; ++i)
      if (i < my_A) gval += dz_s[my_zcol + i] * zd[i];
    if (my_out >= 0) dheads[(size_t)r * NH + my_out] = gval;
    else drad_rows[(size_t)my_ci * B + r] = gval;
  }
}

// launch 2 of the backward: everything that sums over the batch.  The first 64 workgroups: dW_d0 / db_d0 = dd0^T [z | 1] for
// 32 channel-last entries each (thread = (row group, entry quad), rows g, g + 32, ... requested first, the 32 groups added in
// order through LDS, stored at the reference's row c * 16 + p); the next: the radius gradients, each a fixed-order sum over
// the rows; the last 257: the heads -- job_linear_bwd_skn against the channel-last flatten (dW_heads, da2 = (dheads W_heads)
// [a2 > 0], db_heads).
template <int NN>
__global__ __launch_bounds__(256) void k_cl_latent_bwd_cols(CompTable t, const float* __restrict__ a2,
                                                            const float* __restrict__ W_heads,
                                                            const float* __restrict__ dheads, int NH,
                                                            float* __restrict__ dW_heads, float* __restrict__ db_heads,
                                                            float* __restrict__ da2, bf16r* __restrict__ da2p,
                                                            long long da2ps, float* __restrict__ da2cs,
                                                            const float* __restrict__ dd0,
                                                            const float* __restrict__ z, int Z,
                                                            float* __restrict__ dW_d0, float* __restrict__ db_d0,
                                                            const float* __restrict__ drad_rows,
                                                            float* __restrict__ dradii, int B) {
  // order: the 64 + 1 short jobs first, then the 257 workgroups of the heads -- the grid has 66 more workgroups than the chip
  // has CUs, and a short job sharing a CU with a heads workgroup costs less at the front than as the kernel's tail
  // one LDS block, overlaid by the job classes (two workgroups share a CU)
  struct Dd0Lds {
    f32x4 sm[9][32][9];
    __attribute__((aligned(16))) float z_sh[256][16];
  };
  constexpr size_t kLds = sizeof(SknLds<NN>) > sizeof(Dd0Lds) ? sizeof(SknLds<NN>) : sizeof(Dd0Lds);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kLds];
  const int nshort = kD0 / 32 + 1;
  int blk = blockIdx.x;
#ifdef MV_CL_DBG  // A/B builds (tools/build_variant.py): which job class bounds the launch
  if ((MV_CL_DBG & 1) && blk < kD0 / 32) return;
  if ((MV_CL_DBG & 2) && blk >= nshort) return;
  if ((MV_CL_DBG & 4) && blk == kD0 / 32) return;
#endif
  if (blk >= nshort) {
    job_linear_bwd_skn<NN, true>(*reinterpret_cast<SknLds<NN>*>(lds_raw), blk - nshort, a2, W_heads, dheads, dW_heads, db_heads, da2, B, NH, kFlat, 1, da2p, da2ps, da2cs);
    return;
  }
  const int tid = threadIdx.x;
  if (blk < kD0 / 32) {
    // dW_d0[n][k] = sum_m dd0[m][n] z[m][k], db_d0[n] = sum_m dd0[m][n] for 32 columns n.  Latency-bound: every request of a
    // 256-row chunk (dd0 and the chunk's z rows, through LDS: one coalesced request per thread instead of Z scalar loads per
    // row inside the loop) is in flight before the first use, and the 32 row groups meet in LDS with ONE barrier per 9 outputs.
    Dd0Lds& D = *reinterpret_cast<Dd0Lds*>(lds_raw);
    auto& sm = D.sm;
    auto& z_sh = D.z_sh;
    const int c8 = tid & 7, g = tid >> 3;
    const int idx = blk * 32 + c8 * 4, pp = idx >> 7, cc = idx & 127;
    f32x4 acc[17];  // [k < Z]: dW_d0 column k; [16]: the bias gradient
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int m0 = 0; m0 < B; m0 += 256) {
      f32x4 dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = m0 + g + 32 * u;
        dv[u] = *reinterpret_cast<const f32x4*>(dd0 + (size_t)(m < B ? m : B - 1) * kD0 + idx);
      }
      float zv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // 256 rows x Z values, consecutive threads consecutive addresses
        const int e = tid + 256 * q;
        zv[q] = 0.f;
        if (q < Z) {  // uniform
          const int m = m0 + e / Z;
          zv[q] = z[(size_t)(m < B ? m : B - 1) * Z + (e - (e / Z) * Z)];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // the previous chunk's readers of z_sh are done
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (q < Z) {
          const int e = tid + 256 * q, r = e / Z;
          z_sh[r][e - r * Z] = zv[q];
        }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = g + 32 * u;
        if (m0 + r >= B) continue;
        acc[16] += dv[u];
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < Z) acc[k] += z_sh[r][k] * dv[u];
      }
    }
    const int nout = Z + 1;  // outputs 0 .. Z - 1: columns of dW_d0; output Z: the bias gradient
    for (int k0 = 0; k0 < nout; k0 += 9) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 17; ++k) {
        const int o = (k == 16 ? Z : k) - k0;
        if ((k < Z || k == 16) && o >= 0 && o < 9) sm[o][g][c8] = acc[k];  // uniform conditions
      }
      __syncthreads();
      // two threads per (output, column quad): each adds 16 of the 32 row groups in group order, the halves meet by a lane swap
      const int o = tid >> 4, c2 = (tid >> 1) & 7, hsel = tid & 1;
      if (o < 9 && k0 + o < nout) {
        f32x4 tt = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < 16; ++q) tt += sm[o][hsel * 16 + q][c2];
#pragma unroll
        for (int e = 0; e < 4; ++e) tt[e] += lane_swap1(tt[e]);
        if (hsel == 0) {
          const int k = k0 + o;
          const int idx2 = blk * 32 + c2 * 4, pp2 = idx2 >> 7, cc2 = idx2 & 127;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = (cc2 + e) * kPix + pp2;
            if (k < Z) dW_d0[(size_t)n * Z + k] = tt[e];
            else db_d0[n] = tt[e];
          }
        }
      }
    }
    (void)pp; (void)cc;
    return;
  }
  __shared__ float sw[4];
  for (int ci = 0; ci < t.n; ++ci) {
    if (tid == 0) dradii[ci] = 0.f;
  }
  __syncthreads();
  for (int ci = 0; ci < t.n; ++ci) {
    if (!(t.trainable[ci] & 1)) continue;  // uniform: no radius direction (Euclidean)
    const float* prow = drad_rows + (size_t)ci * B;
    float sacc = 0.f;
    for (int rr = tid; rr < B; rr += 256) sacc += prow[rr];
    sacc = wave_sum(sacc);
    __syncthreads();
    if ((tid & 63) == 0) sw[tid >> 6] = sacc;
    __syncthreads();
    if (tid == 0) dradii[t.c[ci].radius_idx] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
  }
}

static int conv_latent_dims(const mvae_component_desc* comps, int ncomp, int* NH, int* Z, int* eps_dim, int* dmax) {
  if (!comps || ncomp < 1 || ncomp > kMaxComp) return 0;
  int nh = 0, zz = 0, ed = 0, dm = 0;
  for (int i = 0; i < ncomp; ++i) {
    const mvae_component_desc& c = comps[i];
    if (c.kind < 0 || c.kind >= kNumKinds || c.true_dim < 1) return 0;
    const int a = c.mean_col + c.true_dim, b = c.logvar_col + c.logvar_dim, zc = c.z_col + ambient_dim(c.kind, c.true_dim);
    nh = a > nh ? a : nh;
    nh = b > nh ? b : nh;
    zz = zc > zz ? zc : zz;
    ed = (c.eps_col + c.true_dim) > ed ? (c.eps_col + c.true_dim) : ed;
    dm = c.true_dim > dm ? c.true_dim : dm;
  }
  *NH = nh; *Z = zz; *eps_dim = ed; *dmax = dm;
  return 1;
}

extern "C" int mvae_conv_latent_supported(const mvae_component_desc* comps, int ncomp) {
  int NH, Z, ed, dm;
  if (!conv_latent_dims(comps, ncomp, &NH, &Z, &ed, &dm)) return 0;
  return (NH <= 16 && Z <= 16 && dm <= 8) ? 1 : 0;
}

extern "C" int64_t mvae_conv_latent_workspace_floats(int64_t B, int ncomp) {
  if (B < 1 || ncomp < 1) return -1;
  const int64_t fwd = (int64_t)kClSlices * B * kClNN, bwd = B * kD0 + (int64_t)ncomp * B;
  return fwd > bwd ? fwd : bwd;
}

#define MV_CL_DMAX_SWITCH(dmax, ...)                              \
  if (dmax <= 2) { constexpr int DM = 2; __VA_ARGS__; }           \
  else if (dmax <= 4) { constexpr int DM = 4; __VA_ARGS__; }      \
  else { constexpr int DM = 8; __VA_ARGS__; }
#define MV_CL_NN_SWITCH(nh, ...)                                  \
  if (nh <= 4) { constexpr int NN = 4; __VA_ARGS__; }             \
  else if (nh <= 8) { constexpr int NN = 8; __VA_ARGS__; }        \
  else if (nh <= 12) { constexpr int NN = 12; __VA_ARGS__; }      \
  else { constexpr int NN = 16; __VA_ARGS__; }

extern "C" int mvae_conv_latent_forward(const mvae_component_desc* comps, int ncomp, const float* a2,
                                        const float* W_heads, const float* b_heads, const float* eps, int eps_ld,
                                        const float* radii, const float* W_d0, const float* b_d0, float* heads, float* z,
                                        float* kl, float* t0, uint16_t* t0_planes, int64_t t0_ps, float* workspace,
                                        int64_t B, void* stream) {
  if (!a2 || !W_heads || !b_heads || !eps || !W_d0 || !b_d0 || !heads || !z || !kl || !t0 || !workspace || B < 1 ||
      B > 0x3fffff)
    return fail(MVAE_E_BADARG, "null pointer / bad batch%s", "");
  int NH, Z, ed, dmax;
  if (!mvae_conv_latent_supported(comps, ncomp) || !conv_latent_dims(comps, ncomp, &NH, &Z, &ed, &dmax))
    return fail(MVAE_E_UNSUPPORTED, "fused conv latent section: heads_dim, z_dim <= 16 and true dimensions <= 8%s", "");
  if (eps_ld < ed) return fail(MVAE_E_BADARG, "eps_ld smaller than the components' eps columns%s", "");
  for (int i = 0; i < ncomp; ++i)
    if (comps[i].radius_idx < 0 || comps[i].radius_idx >= ncomp)
      return fail(MVAE_E_BADARG, "radius_idx out of range%s (%lld)", "", comps[i].radius_idx);
  if (((((uintptr_t)a2) | ((uintptr_t)t0) | ((uintptr_t)workspace) | ((uintptr_t)W_d0)) & 15) != 0)
    return fail(MVAE_E_ALIGN, "fused conv latent section needs 16-byte aligned a2 / t0 / W_d0 / workspace%s", "");
  if (t0_planes && ((((uintptr_t)t0_planes) & 7) || (t0_ps & 3))) return fail(MVAE_E_ALIGN, "t0 planes: 8-byte aligned%s", "");
  CompTable t;
  unsigned char all[kMaxComp];
  memset(all, 1, sizeof(all));
  int dm2;
  int rc = fill_table(&t, comps, ncomp, all, &dm2);
  if (rc) return rc;
  for (int i = 0; i < ncomp; ++i) {
    if (comps[i].kind != MVAE_EUCLIDEAN && !radii) return fail(MVAE_E_BADARG, "radii is NULL%s", "");
    if (t.lane_of[i] >= 64) return fail(MVAE_E_UNSUPPORTED, "too many components of one kind%s", "");
  }
  hipStream_t s = (hipStream_t)stream;
  const dim3 gridA(kClSlices, (unsigned)((B + 63) / 64));
  MV_CL_NN_SWITCH(NH, hipLaunchKernelGGL((k_cl_heads_part<NN>), gridA, dim3(256), 0, s, a2, W_heads, workspace, (int)B, NH));
  SplitJobs4 ride;
  ride.njobs = 0;
  ride.blk0[0] = 0;
  static const bool no_ride = [] { const char* e = getenv("MVAE_SPLIT_RIDE"); return e && e[0] == '0'; }();
  const int n_ride = (!no_ride && p3_take_splitjobs(&ride)) ? ride.blk0[ride.njobs] : 0;
  MV_CL_DMAX_SWITCH(dmax, hipLaunchKernelGGL((k_cl_latent_fwd<DM>), dim3((unsigned)(B + n_ride)), dim3(256), 0, s, t, workspace,
                                             b_heads, NH, eps, eps_ld, radii, W_d0, b_d0, Z, heads, z, kl, t0,
                                             reinterpret_cast<bf16r*>(t0_planes), (long long)t0_ps, (int)B, ride));
  LAUNCH_CHECK("fused conv latent forward launch");
  return 0;
}

extern "C" int mvae_conv_latent_backward(const mvae_component_desc* comps, int ncomp, const float* a2,
                                         const float* W_heads, const float* heads, const float* eps, int eps_ld,
                                         const float* radii, const float* z, const float* W_d0, const float* t0,
                                         const float* dt0, int dt0_slices, int64_t dt0_slice_stride, float beta,
                                         float* dW_heads, float* db_heads, float* da2,
                                         uint16_t* da2_planes, int64_t da2_ps, float* da2_chansum, float* da2_chansum_ws,
                                         float* dW_d0, float* db_d0, float* dradii, float* dheads, float* workspace, int64_t B,
                                         void* stream) {
  if ((da2_chansum == nullptr) != (da2_chansum_ws == nullptr)) return fail(MVAE_E_BADARG, "da2_chansum and its workspace go together%s", "");
  if (!da2 && !(da2_planes && da2_chansum)) return fail(MVAE_E_BADARG, "da2 may only be NULL with planes + channel sums%s", "");
  if (!a2 || !W_heads || !heads || !eps || !z || !W_d0 || !t0 || !dt0 || !dW_heads || !db_heads || !dW_d0 ||
      !db_d0 || !dradii || !dheads || !workspace || B < 1 || B > 0x3fffff)
    return fail(MVAE_E_BADARG, "null pointer / bad batch%s", "");
  int NH, Z, ed, dmax;
  if (!mvae_conv_latent_supported(comps, ncomp) || !conv_latent_dims(comps, ncomp, &NH, &Z, &ed, &dmax))
    return fail(MVAE_E_UNSUPPORTED, "fused conv latent section: heads_dim, z_dim <= 16 and true dimensions <= 8%s", "");
  if (eps_ld < ed) return fail(MVAE_E_BADARG, "eps_ld smaller than the components' eps columns%s", "");
  if (((((uintptr_t)a2) | ((uintptr_t)t0) | ((uintptr_t)dt0) | ((uintptr_t)da2) | ((uintptr_t)workspace) |
        ((uintptr_t)W_d0) | ((uintptr_t)da2_chansum_ws)) & 15) != 0)
    return fail(MVAE_E_ALIGN, "fused conv latent section needs 16-byte aligned activations / W_d0 / workspace%s", "");
  if (da2_planes && ((((uintptr_t)da2_planes) & 7) || (da2_ps & 3))) return fail(MVAE_E_ALIGN, "da2 planes: 8-byte aligned%s", "");
  CompTable t;
  unsigned char tr[kMaxComp];
  memset(tr, 1, sizeof(tr));
  int dm2;
  int rc = fill_table(&t, comps, ncomp, tr, &dm2);
  if (rc) return rc;
  for (int i = 0; i < ncomp; ++i) {
    if (comps[i].kind != MVAE_EUCLIDEAN && !radii) return fail(MVAE_E_BADARG, "radii is NULL%s", "");
    if (comps[i].radius_idx < 0 || comps[i].radius_idx >= ncomp)
      return fail(MVAE_E_BADARG, "radius_idx out of range%s (%lld)", "", comps[i].radius_idx);
  }
  hipStream_t s = (hipStream_t)stream;
  float* dd0 = workspace;
  float* drad_rows = workspace + (size_t)B * kD0;
  MV_CL_DMAX_SWITCH(dmax, hipLaunchKernelGGL((k_cl_latent_bwd_rows<DM>), dim3((unsigned)B), dim3(256), 0, s, t, heads, NH,
                                             eps, eps_ld, radii, W_d0, Z, t0, dt0, dt0_slices,
                                             (long long)dt0_slice_stride, beta, dd0, dheads, drad_rows, (int)B));
  const unsigned grid = kFlat / 32 + 1 + kD0 / 32 + 1;
  MV_CL_NN_SWITCH(NH, hipLaunchKernelGGL((k_cl_latent_bwd_cols<NN>), dim3(grid), dim3(256), 0, s, t, a2, W_heads, dheads,
                                         NH, dW_heads, db_heads, da2, da2_planes, (long long)da2_ps, da2_chansum_ws, dd0, z, Z,
                                         dW_d0, db_d0, drad_rows, dradii, (int)B));
  // the 16 pixels of a channel: added in pixel order by the (deferrable) slice sum
  if (da2_chansum) sum_slices(da2_chansum_ws, da2_chansum, kEncC, kPix, s);
  LAUNCH_CHECK("fused conv latent backward launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ device-side input pipeline
// (items and their arithmetic: mvae_common.hpp, "device-side input pipeline")
__global__ __launch_bounds__(256) void k_prepare_batch(FeedArgs f) {
  const unsigned cursor = (unsigned)f.counters[8];
  const int n = feed_items(f);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) feed_item(f, cursor, i);
}

// out[0 .. n) ~ N(0, 1), never exactly 0 (box_muller4): the eps draw of an eager step / of log_likelihood's n x B samples in ONE
// launch.  Counter-based: item i (four values) = Philox4x32-10 at counter (i lo, i hi, offset lo, offset hi), key = seed --
// the same (seed, offset) gives the same bits on every launch geometry.
__global__ __launch_bounds__(256) void k_randn(float* __restrict__ out, long long n, unsigned long long seed,
                                               unsigned long long offset) {
  const long long items = (n + 3) >> 2;
  const bool vec = (((uintptr_t)out) & 15) == 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)((unsigned long long)i >> 32), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    float v[4];
    box_muller4(r, v);
    if (vec && i * 4 + 3 < n) {
      *reinterpret_cast<f32x4*>(out + i * 4) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (i * 4 + t < n) out[i * 4 + t] = v[t];
    }
  }
}

extern "C" int mvae_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  if (!out || n < 0) return fail(MVAE_E_BADARG, "null pointer / negative count%s", "");
  if (n == 0) return 0;
  const long long items = (n + 3) / 4;
  const long long blocks = (items + 255) / 256;
  hipLaunchKernelGGL(k_randn, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, out,
                     (long long)n, (unsigned long long)seed, (unsigned long long)offset);
  LAUNCH_CHECK("randn launch");
  return 0;
}

extern "C" int mvae_prepare_batch(const uint8_t* images, const int32_t* perm, int n_images, int D, int B, int E,
                                  uint64_t seed, const int32_t* counters, int batches_per_epoch, int train, float* x,
                                  float* eps, void* stream) {
  if (!images || !counters || !x || !eps || n_images < 1 || D < 1 || B < 1 || E < 1 || batches_per_epoch < 1)
    return fail(MVAE_E_BADARG, "null pointer / bad shape%s", "");
  const int work = (B * D + 3) / 4 + (B * E + 3) / 4;
  FeedArgs f = {images, perm, counters, x, eps, (unsigned long long)seed, n_images, D, B, E, batches_per_epoch, train, 0};
  hipLaunchKernelGGL(k_prepare_batch, dim3((work + 255) / 256), dim3(256), 0, (hipStream_t)stream, f);
  LAUNCH_CHECK("prepare batch launch");
  return 0;
}
