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

// The K loops below issue the operand loads of G consecutive k-chunks before any MFMA consumes them (the loads are
// independent of the accumulator, so G L2/HBM round trips overlap), and alternate between two accumulators so that
// consecutive MFMAs do not wait on the 40-cycle dependent-accumulator latency.

// NT:  acc[i][j] += sum_{k in chunks} A[m0+i][k] * W[n0+j][k]      (y = x W^T: both operands K-contiguous)
// The wave processes the 16-wide k-chunks c = chunk0, chunk0+stride, ... < nchunks.
// FULL: every row of the tile exists, K is a multiple of 16 and both operands are 16-byte aligned -- the loads are
// then plain dwordx4 without per-load branches (a chunk index past the end is clamped and its values zeroed).
template <int G = 4, bool FULL = false>
__device__ __forceinline__ f32x4 tile_nt(const float* __restrict__ A, int lda, int M, int m0,
                                         const float* __restrict__ W, int ldw, int N, int n0, int K, int chunk0,
                                         int stride, bool vecA, bool vecW, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const bool a_ok = (m0 + i) < M, w_ok = (n0 + i) < N;
  const float* arow = A + (size_t)(a_ok ? m0 + i : 0) * lda;
  const float* wrow = W + (size_t)(w_ok ? n0 + i : 0) * ldw;
  const int nchunks = (K + 15) >> 4;
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int c = chunk0; c < nchunks; c += stride * G) {
    float4 a[G], b[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int cc = c + g * stride;
      const int k = (cc << 4) + (q << 2);
      const bool ok = cc < nchunks;
      if (FULL) {
        // pure requests: NO select on a loaded value in this block (it needs the value, so hipcc puts an s_waitcnt
        // between the requests and the batch becomes two serialized memory round trips); a chunk past the end reads a
        // clamped address and is zeroed after the barrier
        const int kk = ok ? k : 0;
        a[g] = *reinterpret_cast<const float4*>(arow + kk);
        b[g] = *reinterpret_cast<const float4*>(wrow + kk);
      } else {
        a[g] = load4_guarded(arow, k, K, a_ok && ok, vecA);
        b[g] = load4_guarded(wrow, k, K, w_ok && ok, vecW);
      }
    }
    // keep every load of the batch above the first MFMA: without this the scheduler trades registers for a
    // load -> wait -> MFMA interleaving, i.e. G serialized memory round trips
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (FULL && c + g * stride >= nchunks) a[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = mfma16(a[g].x, b[g].x, acc);
      acc2 = mfma16(a[g].y, b[g].y, acc2);
      acc = mfma16(a[g].z, b[g].z, acc);
      acc2 = mfma16(a[g].w, b[g].w, acc2);
    }
  }
  return acc + acc2;
}

// TN:  acc[i][j] += sum_{m} P[m][p0+i] * Q[m][q0+j]     (dW = dy^T x: contraction over the batch rows)
// The wave processes rows m = 4*s + q for steps s = step0, step0+stride, ... ; 4 rows per MFMA.
template <int G = 8, bool FULL = false>
__device__ __forceinline__ f32x4 tile_tn(const float* __restrict__ P, int ldp, int NP, int p0,
                                         const float* __restrict__ Q, int ldq, int NQ, int q0, int Mrows, int step0,
                                         int stride, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const bool p_ok = (p0 + i) < NP, q_ok = (q0 + i) < NQ;
  const int nsteps = (Mrows + 3) >> 2;
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int s = step0; s < nsteps; s += stride * G) {
    float a[G], b[G];
    // requests: branch-free, indices past the end clamped to valid addresses.  NO select on a loaded value in this
    // block -- it would need the value and so serialise the requests; the masks are applied after the barrier.
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int m = ((s + g * stride) << 2) + q;
      const int mm = (m < Mrows) ? m : 0;
      if (FULL) {  // all columns exist
        a[g] = P[(size_t)mm * ldp + p0 + i];
        b[g] = Q[(size_t)mm * ldq + q0 + i];
      } else {
        a[g] = P[(size_t)mm * ldp + (p_ok ? p0 + i : 0)];
        b[g] = Q[(size_t)mm * ldq + (q_ok ? q0 + i : 0)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {  // mask, then multiply, in arrival order
      const bool m_ok = (((s + g * stride) << 2) + q) < Mrows;
      a[g] = (m_ok && (FULL || p_ok)) ? a[g] : 0.f;  // this lane's P column; a zero row factor covers m
      if (!FULL) b[g] = q_ok ? b[g] : 0.f;           // this lane's Q column
      if (g & 1) acc2 = mfma16(a[g], b[g], acc2);
      else acc = mfma16(a[g], b[g], acc);
    }
  }
  return acc + acc2;
}

// NN:  acc[i][j] += sum_{k} G[m0+i][k] * W[k][n0+j]      (dx = dy W: G is K-contiguous, W is N-contiguous)
template <int G = 4, bool FULL = false>
__device__ __forceinline__ f32x4 tile_nn(const float* __restrict__ Gm, int ldg, int M, int m0,
                                         const float* __restrict__ W, int ldw, int N, int n0, int K, int chunk0,
                                         int stride, bool vecG, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const bool g_ok = (m0 + i) < M, w_ok = (n0 + i) < N;
  const float* grow = Gm + (size_t)(g_ok ? m0 + i : 0) * ldg;
  const float* wcol = W + (w_ok ? n0 + i : 0);
  const int nchunks = (K + 15) >> 4;
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int c = chunk0; c < nchunks; c += stride * G) {
    float4 a[G];
    float b[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int cc = c + g * stride;
      const int k = (cc << 4) + (q << 2);
      const bool ok = cc < nchunks;
      if (FULL) {  // pure requests (see tile_nt); a chunk past the end is zeroed after the barrier
        const int kk = ok ? k : 0;
        a[g] = *reinterpret_cast<const float4*>(grow + kk);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[g][t] = wcol[(size_t)(kk + t) * ldw];
        continue;
      }
      a[g] = load4_guarded(grow, k, K, g_ok && ok, vecG);
#pragma unroll
      for (int t = 0; t < 4; ++t) b[g][t] = (w_ok && ok && k + t < K) ? wcol[(size_t)(k + t) * ldw] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (FULL && c + g * stride >= nchunks) a[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = mfma16(a[g].x, b[g][0], acc);
      acc2 = mfma16(a[g].y, b[g][1], acc2);
      acc = mfma16(a[g].z, b[g][2], acc);
      acc2 = mfma16(a[g].w, b[g][3], acc2);
    }
  }
  return acc + acc2;
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic (lgkmcnt) and NOT for its
// outstanding global loads/stores.  __syncthreads() also drains vmcnt, i.e. it stalls every wave until prefetches issued
// for later phases have landed and earlier global stores have been acknowledged -- a full memory round trip per barrier
// in kernels whose phases are ~1 us long.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
  lds_barrier();
  const int r = tid >> 4, c = tid & 15;
  float s = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
  lds_barrier();
  return s;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mv
