// mvae_gemm.hpp -- fp32 MFMA building blocks for the dense layers of the step (gfx950 / CDNA4).
//
// All three contractions of a Linear layer are expressed on ONE wave-level primitive: a 16x16 output tile
// accumulated with v_mfma_f32_16x16x4_f32 (exact f32, == an fmaf chain; 32-cycle issue), operands loaded straight
// from global/L2 into VGPRs (no LDS staging: at batch 128 every operand tile is used by exactly one wave, so an LDS
// round trip would be pure latency).  A workgroup is 4 waves that split the contraction dimension; the four partial
// tiles meet in LDS (4 KiB) and the 256 threads apply the epilogue, one output element each, with coalesced stores.
//
// Lane mapping of v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md section 3): lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15]; it receives D[row=(l>>4)*4+r][col=l&15], r=0..3.
// The k index inside an MFMA step is a free relabelling as long as A and B agree, which the NT form uses to load
// 16 bytes per lane: lane (i, q) loads k = k0+4q..k0+4q+3 of its row and feeds the four components to four MFMAs.
#pragma once
#include <hip/hip_runtime.h>

namespace mv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float4 load4_guarded(const float* __restrict__ row, int k, int K, bool row_ok, bool vec_ok) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok) return r;
  if (vec_ok && k + 3 < K) return *reinterpret_cast<const float4*>(row + k);
  if (k < K) r.x = row[k];
  if (k + 1 < K) r.y = row[k + 1];
  if (k + 2 < K) r.z = row[k + 2];
  if (k + 3 < K) r.w = row[k + 3];
  return r;
}

// NT:  acc[i][j] += sum_{k in chunks} A[m0+i][k] * W[n0+j][k]      (y = x W^T: both operands K-contiguous)
// The wave processes the 16-wide k-chunks c = chunk0, chunk0+stride, ... < nchunks.
__device__ __forceinline__ f32x4 tile_nt(const float* __restrict__ A, int lda, int M, int m0,
                                         const float* __restrict__ W, int ldw, int N, int n0, int K, int chunk0,
                                         int stride, bool vecA, bool vecW, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const bool a_ok = (m0 + i) < M, w_ok = (n0 + i) < N;
  const float* arow = A + (size_t)(a_ok ? m0 + i : 0) * lda;
  const float* wrow = W + (size_t)(w_ok ? n0 + i : 0) * ldw;
  const int nchunks = (K + 15) >> 4;
#pragma unroll 4
  for (int c = chunk0; c < nchunks; c += stride) {
    const int k = (c << 4) + (q << 2);
    float4 a = load4_guarded(arow, k, K, a_ok, vecA);
    float4 b = load4_guarded(wrow, k, K, w_ok, vecW);
    acc = mfma16(a.x, b.x, acc);
    acc = mfma16(a.y, b.y, acc);
    acc = mfma16(a.z, b.z, acc);
    acc = mfma16(a.w, b.w, acc);
  }
  return acc;
}

// TN:  acc[i][j] += sum_{m} P[m][p0+i] * Q[m][q0+j]     (dW = dy^T x: contraction over the batch rows)
// The wave processes rows m = 4*(s) + q for steps s = step0, step0+stride, ... ; 4 rows per MFMA.
__device__ __forceinline__ f32x4 tile_tn(const float* __restrict__ P, int ldp, int NP, int p0,
                                         const float* __restrict__ Q, int ldq, int NQ, int q0, int Mrows, int step0,
                                         int stride, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const bool p_ok = (p0 + i) < NP, q_ok = (q0 + i) < NQ;
  const int nsteps = (Mrows + 3) >> 2;
#pragma unroll 8
  for (int s = step0; s < nsteps; s += stride) {
    const int m = (s << 2) + q;
    const bool m_ok = m < Mrows;
    float a = (p_ok && m_ok) ? P[(size_t)m * ldp + p0 + i] : 0.f;
    float b = (q_ok && m_ok) ? Q[(size_t)m * ldq + q0 + i] : 0.f;
    acc = mfma16(a, b, acc);
  }
  return acc;
}

// NN:  acc[i][j] += sum_{k} G[m0+i][k] * W[k][n0+j]      (dx = dy W: G is K-contiguous, W is N-contiguous)
__device__ __forceinline__ f32x4 tile_nn(const float* __restrict__ G, int ldg, int M, int m0,
                                         const float* __restrict__ W, int ldw, int N, int n0, int K, int chunk0,
                                         int stride, bool vecG, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const bool g_ok = (m0 + i) < M, w_ok = (n0 + i) < N;
  const float* grow = G + (size_t)(g_ok ? m0 + i : 0) * ldg;
  const int nchunks = (K + 15) >> 4;
#pragma unroll 2
  for (int c = chunk0; c < nchunks; c += stride) {
    const int k = (c << 4) + (q << 2);
    float4 a = load4_guarded(grow, k, K, g_ok, vecG);
    float b0 = (w_ok && k < K) ? W[(size_t)k * ldw + n0 + i] : 0.f;
    float b1 = (w_ok && k + 1 < K) ? W[(size_t)(k + 1) * ldw + n0 + i] : 0.f;
    float b2 = (w_ok && k + 2 < K) ? W[(size_t)(k + 2) * ldw + n0 + i] : 0.f;
    float b3 = (w_ok && k + 3 < K) ? W[(size_t)(k + 3) * ldw + n0 + i] : 0.f;
    acc = mfma16(a.x, b0, acc);
    acc = mfma16(a.y, b1, acc);
    acc = mfma16(a.z, b2, acc);
    acc = mfma16(a.w, b3, acc);
  }
  return acc;
}

// Combine the four waves' partial tiles.  red is float[4][16][17] in LDS (row padded: the epilogue reads a column of
// the wave dimension).  Returns the full sum for element (row = tid>>4, col = tid&15) of the 16x16 tile.
__device__ __forceinline__ float reduce_tiles(float (*red)[16][17], f32x4 acc) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, rbase = (lane >> 4) << 2;
  red[wave][rbase + 0][col] = acc[0];
  red[wave][rbase + 1][col] = acc[1];
  red[wave][rbase + 2][col] = acc[2];
  red[wave][rbase + 3][col] = acc[3];
  __syncthreads();
  const int r = tid >> 4, c = tid & 15;
  float s = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
  __syncthreads();
  return s;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mv
