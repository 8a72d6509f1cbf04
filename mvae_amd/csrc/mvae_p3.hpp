// mvae_p3.hpp -- shared by mvae_conv.hip and mvae_p3.hip: the geometry record of the k4 s2 p1 convolutions and the exact
// three-way bf16 split of a float ("planes").
//
// A float has 24 significant bits; truncating to bf16 keeps 8.  x = h + m + l EXACTLY with h = bf16_trunc(x),
// m = bf16_trunc(x - h), l = bf16_trunc(x - h - m) (each subtraction is exact: the operands share the leading bits).
// Stored as three bf16 PLANES of the tensor's shape, a contraction multiplies through the six largest piece products on the
// bf16 MFMA (mvae_p3.hip); a producing epilogue writes the planes next to its f32 result, so no consumer splits again.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ConvGeom {
  int Cc, IH, IW;    // channels and extent of the SOURCE image
  int lOW, lOHW;     // log2(OW), log2(OH * OW), OH = IH / 2, OW = IW / 2 (powers of two)
};

typedef unsigned short bf16r;  // storage type of a plane entry (raw bf16 bits)
typedef unsigned int p3_u32x2 __attribute__((ext_vector_type(2)));

// 4 consecutive floats -> their (hi, mid, lo) bf16 pieces, two per dword (element 0 in the low half)
__device__ __forceinline__ void split3_planes(const float x0, const float x1, const float x2, const float x3, p3_u32x2* hi,
                                              p3_u32x2* mi, p3_u32x2* lo) {
  const float x[4] = {x0, x1, x2, x3};
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
  *hi = p3_u32x2{__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
  *mi = p3_u32x2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
  *lo = p3_u32x2{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
}

// store the planes of 4 consecutive entries starting at element index `idx` (a multiple of 4): three 8-byte stores
__device__ __forceinline__ void store_planes4(bf16r* planes, long long plane_stride, size_t idx, float x0, float x1, float x2,
                                              float x3) {
  p3_u32x2 h, m, l;
  split3_planes(x0, x1, x2, x3, &h, &m, &l);
  *reinterpret_cast<p3_u32x2*>(planes + idx) = h;
  *reinterpret_cast<p3_u32x2*>(planes + plane_stride + idx) = m;
  *reinterpret_cast<p3_u32x2*>(planes + 2 * plane_stride + idx) = l;
}

// ---- deferred column sums of tall matrices (bias gradients): job table and workgroup body, shared by the flush launch
// (mvae_conv.hip) and the launches that carry the jobs as extra workgroups (mvae_edge.hip).  Thread (row group g = tid >> 4,
// column quad c = tid & 15) of the first 256 threads adds rows g, g + 16, ... of its 512-row slice, 8 requests of 16 bytes in
// flight; the 16 row groups meet in LDS and are added in group order: per column the same additions in the same order as
// k_colsum_sliced.
constexpr int kColSlice = 512;  // rows per slice of the tall column sums (mvae_colsum)
// ---- planes of existing f32 tensors (mvae_split3_planes): the job list and the workgroup body, shared by the stand-alone launch
// (k_split3, mvae_p3.hip) and by a launch that carries the jobs as extra workgroups (the conv latent forward, mvae_conv.hip)
constexpr int kMaxSplitJobs = 12;
struct SplitJobs {
  const float* src[kMaxSplitJobs];
  bf16r* dst[kMaxSplitJobs];
  long long n[kMaxSplitJobs];  // floats, a multiple of 4; planes at dst, dst + n, dst + 2 n
  int blk0[kMaxSplitJobs + 1];
  int njobs;
};
constexpr int kRideSplitJobs = 4;  // what a carrying launch takes (the four channel-last weight matrices of the conv step)
struct SplitJobs4 {
  const float* src[kRideSplitJobs];
  bf16r* dst[kRideSplitJobs];
  long long n[kRideSplitJobs];
  int blk0[kRideSplitJobs + 1];
  int njobs;  // 0: nothing rides
};
template <class Jobs>
__device__ __forceinline__ void split3_body(const Jobs& jobs, const int blk, const int nthreads) {
  int j = 0;
  while (j + 1 < jobs.njobs && blk >= jobs.blk0[j + 1]) ++j;  // uniform
  const float* src = jobs.src[j];
  bf16r* dst = jobs.dst[j];
  const long long n = jobs.n[j], n4 = n >> 2;
  const int nblk = jobs.blk0[j + 1] - jobs.blk0[j];
  for (long long i = (long long)(blk - jobs.blk0[j]) * nthreads + threadIdx.x; i < n4; i += (long long)nblk * nthreads) {
    typedef float split_f4 __attribute__((ext_vector_type(4)));  // (f32x4 of mvae_common.hpp is declared after this header)
    const split_f4 v = reinterpret_cast<const split_f4*>(src)[i];
    store_planes4(dst, n, (size_t)i * 4, v[0], v[1], v[2], v[3]);
  }
}

constexpr int kMaxColJobs = 12;
struct ColJobs {
  const float* G[kMaxColJobs];
  float* part[kMaxColJobs];
  int M[kMaxColJobs], N[kMaxColJobs], ncb[kMaxColJobs];
  int blk0[kMaxColJobs + 1];
  int njobs;
};
typedef float p3_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void colsum_batched_body(const ColJobs& jobs, const int blk) {
  __shared__ p3_f32x4 sm[16][17];
  int j = 0;
  while (j + 1 < jobs.njobs && blk >= jobs.blk0[j + 1]) ++j;  // uniform
  const int b = blk - jobs.blk0[j];
  const int ncb = jobs.ncb[j], N = jobs.N[j], M = jobs.M[j];
  const int slice = b / ncb, cb = b - slice * ncb;
  const int m0 = slice * kColSlice;
  const int rows = (M - m0) < kColSlice ? (M - m0) : kColSlice;
  const int c = threadIdx.x & 15, g = (threadIdx.x >> 4) & 15;
  const bool worker = threadIdx.x < 256;  // (callers may run wider workgroups)
  const int col = cb * 64 + c * 4;
  const bool act = col < N;
  const float* base = jobs.G[j] + (size_t)m0 * N + (act ? col : 0);
  p3_f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (worker) {
    for (int m = g; m < rows; m += 16 * 8) {
      p3_f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int mm = m + 16 * u;
        v[u] = *reinterpret_cast<const p3_f32x4*>(base + (size_t)(mm < rows ? mm : 0) * N);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (m + 16 * u < rows) s += v[u];
    }
    sm[g][c] = s;
  }
  __syncthreads();
  if (worker && g == 0 && act) {
    p3_f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 16; ++q) t += sm[q][c];
    *reinterpret_cast<p3_f32x4*>(jobs.part[j] + (size_t)slice * N + col) = t;
  }
}
