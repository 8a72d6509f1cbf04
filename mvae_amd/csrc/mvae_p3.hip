// mvae_p3.hip -- f32 contractions on PRE-SPLIT operands ("planes", mvae_p3.hpp) for the backward pass of the conv
// architecture (autograd of conv_vae.py:57-79: backward-data and weight gradients of the k4 s2 p1 convolutions).
//
// Why: on gfx950 the f32-input MFMA runs at 1/16 of the bf16 MFMA.  k_gemm_b3 (mvae_conv.hip) already multiplies through the
// exact three-way bf16 split, but splits WHILE STAGING -- 5.5 VALU instructions and three LDS stores per float per K step --
// and its MFMA pipe is 22-34 % busy (profiles/r03b_conv_split_pmc_mfma.json).  Here the operands arrive as three bf16 planes
// written once by whoever produced the tensor (a contraction's epilogue, mvae_split3_planes for the weights), so a K step is
//   LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS, no VGPRs, no VALU, no ds_write)  ->  ds_read_b128 / ds_read_b64_tr_b16
//   ->  6 x v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 sub-product (lh, hl, mm, mh, hm, hh; f32 accumulation).
// Tile BM x BN (128 x 128 or 128 x 64), K step 32, 8 waves, two LDS buffers (one workgroup per CU), ONE barrier per K
// step: [wait for this wave's pieces of step t | barrier | fragment reads of step t | LDS-DMA of step t + 1 into the other
// buffer | MFMAs of step t].  The other buffer was last read in step t - 1, which every wave has left when it arrives at the
// barrier of step t.
//
// Operand forms.  "KC": the contraction index is contiguous in memory, X(i, k) = X[i * ld + k] -- tile image in LDS
// [rows][32 k] (64-byte rows), fragments by ds_read_b128; 16-byte chunk c of row r sits at slot c ^ h((r >> 2) & 3),
// h = {0, 2, 3, 1}: conflict-free under the 16-lane service groups of ds_read_b128.  "KM": the contraction index is the ROW
// index in memory, X(i, k) = X[k * ld + i] (the weight of an NN product, both operands of a weight gradient) -- image
// [32 k][rows], fragments by ds_read_b64_tr_b16 (two per fragment: a 16-lane block reads a [4 k][16 rows] block and
// receives it transposed -- probed in tools/probe_tr_dma.hip: lane j, element e <- (k row e, column j)); 32-byte chunk c of
// k row kr sits at chunk c ^ x(kr), x = the row's index among the rows sharing a 256-byte bank row, so the 8 k rows a
// 32-lane service group touches fall into 8 different bank octets.  LDS-DMA writes lane-linear (base + lane * 16), so both
// swizzles are applied to the per-lane SOURCE address and again at the read (cdna_hip_programming.md, rule 21).
// Gathered forms (implicit convolution, channel-last images; the patch matrix is never written): A_G1 / A_G3 / B_G2 / B_G3W
// as GATHER 1 / 3 / 2 / 3 of k_gemm_tiled; a pixel outside the image reads 16 zero bytes from g_p3_zero.
#include <type_traits>

#include "mvae_common.hpp"
#include "mvae_p3.hpp"

typedef __bf16 p3_bf16x8 __attribute__((ext_vector_type(8)));
typedef short p3_s16x4 __attribute__((ext_vector_type(4)));
typedef short p3_s16x8 __attribute__((ext_vector_type(8)));

#ifndef MV_P3_PRIO
#define MV_P3_PRIO 1  // s_setprio(1) around the MFMA phase (A/B: tools/build_variant.py noprio -DMV_P3_PRIO=0)
#endif
enum { A_KC = 0, A_G1 = 1, A_G3 = 2, A_KM = 3 };
enum { B_KC = 0, B_KM = 1, B_G2 = 2, B_G3W = 3 };

struct P3Args {
  const bf16r* A; long long lda, psa;  // plane q of A at A + q * psa
  const bf16r* B; long long ldb, psb;
  float* C; long long ldc;
  bf16r* Cp; long long psc;            // planes of the result (NULL: none), same ldc
  const float* mask;                   // result zeroed where mask <= 0 (same layout as C), or NULL
  const float* bias; int relu;         // forward epilogue: + bias[n] (or NULL), then max(., 0) -- before the mask
  float* colpart;                      // (two-phase form, no parity) per-row-tile column sums of the result: [M / BM][N], or NULL
  int M, N, K, k_per_slice;
  long long slice_stride;              // floats between the partial results of consecutive K slices (blockIdx.z)
  ConvGeom cg;
  int lCc;                             // log2(cg.Cc)
  unsigned long long* dbgbuf;          // (-DMV_P3_DBG builds only) phase time stamps of workgroup 0
  int xcd;                             // XCD-aware tile order on (the host checked the divisibility it needs)
  int dbg;                             // (-DMV_P3_DBG builds only) bit 0: no DMA waits, 1: no A DMA, 2: no B DMA, 3: no MFMAs, 4: every step loads tile 0, 5: no barriers, 6: no fragment reads
};

__device__ __attribute__((aligned(16))) unsigned int g_p3_zero[4];

// sum over the 16 lanes of a DPP row, every lane of the row receives it (fixed order)
__device__ __forceinline__ float p3_row16_sum(float v) {
  int x = __float_as_int(v);
#define MV_P3_DPP_ADD(CTRL) x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true)));
  MV_P3_DPP_ADD(0xB1)   // quad_perm [1,0,3,2]
  MV_P3_DPP_ADD(0x4E)   // quad_perm [2,3,0,1]
  MV_P3_DPP_ADD(0x141)  // row_half_mirror
  MV_P3_DPP_ADD(0x140)  // row_mirror
#undef MV_P3_DPP_ADD
  return __int_as_float(x);
}
__device__ __forceinline__ int p3_h(int x) { return (0x78 >> (2 * x)) & 3; }  // {0, 2, 3, 1}
// Chunk swizzle of a contraction-major image ([32 k][rows] bf16, CPR = rows / 16 chunks of 32 bytes per k row): a 32-lane
// service group of ds_read_b64_tr_b16 touches the k rows {4 t + i} and {8 + 4 t + i}, i = 0..3, of one 16-column block; with
// x = (k & 7) the two halves met in the same banks (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE on the weight
// gradients, gpurun_out/prof_p3_dWe2); this x sends the eight rows to eight different (bank-row half, chunk) places.
__device__ __forceinline__ int p3_km_swz(int kr, int CPR) {
  const int RPB = 8 / CPR;  // k rows per 256-byte bank row
  return ((kr / RPB) & (CPR / 2 - 1)) + (CPR / 2) * ((kr >> 3) & 1);
}

// KALT: the two wave groups take ALTERNATE K steps of the whole tile (wave tiles twice as large: half the fragment reads per
// MFMA, one barrier interval per K step instead of two) and add their partial sums through LDS before the epilogue; WR then
// counts the rows of the 4-wave grid of one group.
// The body is a device function so that TWO contractions can share one launch (k_gemm_p3_pair below): L = the workgroup's number
// within its contraction's (nx, ny, nz) grid, x fastest; lds = the workgroup's whole LDS allocation.
template <int BM, int BN>
constexpr int p3_lds_bytes(int nst = 3) { return nst * 3 * (BM * 64 + BN * 64); }
template <int BM, int BN, int WR, int AF, int BF, int NST = 3, bool KALT = false>
__device__ __forceinline__ void p3_body(const P3Args& g, unsigned char* __restrict__ lds, const int L0, const int nx, const int ny,
                                        const int nz) {
  constexpr int NW = 8, NWT = KALT ? 4 : 8, WC = NWT / WR, WM = BM / WR, WN = BN / WC, TM = WM / 16, TN = WN / 16;
  constexpr int PLA = BM * 64, PLB = BN * 64;  // bytes per plane tile (32 k x 2 bytes per row)
  constexpr int SB = 3 * (PLA + PLB);          // bytes per LDS buffer
  constexpr int NPA = 3 * BM / 16, NPB = 3 * BN / 16, NP = NPA + NPB;  // 1-KiB DMA pieces per K step
  constexpr int U = (NP + NW - 1) / NW;        // pieces per wave
  constexpr bool A_IS_KM = AF == A_KM, B_IS_KM = BF != B_KC;
  constexpr bool PARITY = AF == A_G3;
  static_assert(TM >= 1 && TN >= 1 && WM % 16 == 0 && WN % 16 == 0, "wave tile");
  static_assert(NST == 2 || NST == 3, "LDS stages");
  static_assert(NST * SB == p3_lds_bytes<BM, BN>(NST), "LDS size");
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;  // LDS byte address of the image
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order.  Workgroup number L of a launch (x fastest, then y, then z) runs on XCD L % 8 (observed; only speed
  // depends on it) and each XCD has its own L2: in dispatch order the tiles sharing an operand panel -- the column tiles of a
  // row block, the tiles of one K slice -- sit on 8 different XCDs and every L2 fetches every panel.  Here an XCD takes tiles
  // that share: whole K slices when there are >= 8 of them (weight gradients), 8 / nz contiguous tile ranges of a slice when
  // there are 2 or 4, a contiguous eighth of the tiles (whole row blocks) of an unsliced product or of a parity class.
  int bx = L0 % nx, by = (L0 / nx) % ny, bz = L0 / (nx * ny);
  if (g.xcd) {
    const int T = nx * ny;
    const int L = L0, xcd = L & 7;
    int t;
    if (!PARITY && nz == 1) {
      // g.xcd = gx: the XCDs as gx column groups x gy = 8 / gx row groups, each a contiguous range of column tiles x a
      // contiguous range of row blocks -- A crosses the fabric gx times, B gy times, and an XCD's share of B stays in its L2
      const int gy = 8 / g.xcd, cn = nx / g.xcd, j = L >> 3;
      t = ((xcd / g.xcd) * (ny / gy) + j / cn) * nx + (xcd % g.xcd) * cn + j % cn;
    } else if (PARITY) {
      const int Lt = L - bz * T;  // (T % 8 == 0)
      t = (Lt & 7) * (T >> 3) + (Lt >> 3);
    } else if (nz >= 8) {  // (nz % 8 == 0)
      const int j = L >> 3;
      bz = xcd + 8 * (j / T);
      t = j % T;
    } else {  // nz = 2 or 4, T % (8 / nz) == 0
      const int q = 8 / nz;
      bz = xcd / q;
      t = (xcd % q) * (T / q) + (L >> 3);
    }
    bx = t % nx;
    by = t / nx;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int kb = PARITY ? 0 : bz * g.k_per_slice;
  const int ke = PARITY ? g.K : ((kb + g.k_per_slice < g.K) ? kb + g.k_per_slice : g.K);
  float* __restrict__ C = g.C + (PARITY ? 0 : (size_t)bz * g.slice_stride);
  const int par_y = PARITY ? (bz >> 1) : 0, par_x = PARITY ? (bz & 1) : 0;
  const ConvGeom cg = g.cg;
  const bf16r* zero = reinterpret_cast<const bf16r*>(g_p3_zero);

  // ---- LDS-DMA geometry.  A plane tile is BM / 16 one-KiB pieces (KC: 16 rows of 64 bytes; KM: 1024 / (2 BM) k rows).  The waves
  // move pieces w, w + 8, ... of every plane of both operands: one address computation per operand piece serves three DMA
  // instructions (the planes differ by a uniform stride); everything that does not depend on the K step is kept in
  // registers (inv per piece).
  constexpr int PPA = BM / 16, PPB = BN / 16;         // pieces per plane
  constexpr int NI = NWT;                              // the waves that move a tile: all eight, or (KALT) the group that consumes it
  constexpr int UA = (PPA + NI - 1) / NI, UB = (PPB + NI - 1) / NI;
  struct Inv { long long off; int i0, i1, i2, i3; };
  // lane geometry inside a piece: KC -> (tile row r, source chunk offset c8 in elements); KM -> (k row kr, column offset col)
  auto kc_lane = [&](int pc, int* r, int* c8) __attribute__((always_inline)) {
    *r = pc * 16 + (lane >> 2);
    *c8 = ((lane & 3) ^ p3_h((lane >> 4) & 3)) * 8;
  };
  auto km_lane = [&](int pc, int rows, int* kr, int* col) __attribute__((always_inline)) {
    const int CPR = rows / 16;
    const int off = pc * 1024 + lane * 16, inrow = off % (rows * 2);
    *kr = off / (rows * 2);
    *col = (((inrow >> 5) ^ p3_km_swz(*kr, CPR)) << 4) + ((inrow >> 4) & 1) * 8;
  };
  auto a_inv = [&](int pc) __attribute__((always_inline)) -> Inv {
    Inv v{0, 0, 0, 0, 0};
    if constexpr (A_IS_KM) {
      int kr, col;
      km_lane(pc, BM, &kr, &col);
      v.off = (long long)kr * g.lda + m0 + col;
    } else {
      int r, c8;
      kc_lane(pc, &r, &c8);
      const int m = m0 + r;
      if constexpr (AF == A_KC) {
        v.off = (long long)m * g.lda + c8;
      } else {  // row = output pixel (A_G1) / pixel of this parity class (A_G3)
        const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = m >> cg.lOHW;
        v.i0 = AF == A_G1 ? 2 * oy - 1 : oy + par_y;
        v.i1 = AF == A_G1 ? 2 * ox - 1 : ox + par_x;
        v.i2 = b * cg.IH * cg.IW;
        v.i3 = c8;
      }
    }
    return v;
  };
  auto b_inv = [&](int pc) __attribute__((always_inline)) -> Inv {
    Inv v{0, 0, 0, 0, 0};
    if constexpr (!B_IS_KM) {
      int r, c8;
      kc_lane(pc, &r, &c8);
      v.off = (long long)(n0 + r) * g.ldb + c8;
    } else {
      int kr, col;
      km_lane(pc, BN, &kr, &col);
      const int j = n0 + col;
      if constexpr (BF == B_KM) {
        v.off = (long long)kr * g.ldb + j;
      } else if constexpr (BF == B_G2) {  // j = (tap, channel) of the patch: fixed per lane
        const int tap = j >> g.lCc;
        v.i0 = (tap >> 2) - 1;
        v.i1 = (tap & 3) - 1;
        v.i2 = j & ((1 << g.lCc) - 1);
        v.i3 = kr;
      } else {  // B_G3W
        v.i2 = j;
        v.i3 = kr;
      }
    }
    return v;
  };
  // element offset of this lane's 16 bytes inside a plane at K step k0 (-1: outside the image -> zeros)
  auto a_off = [&](const Inv& v, int k0, bool* ok) __attribute__((always_inline)) -> long long {
    *ok = true;
    if constexpr (AF == A_KC) return v.off + k0;
    else if constexpr (AF == A_KM) return v.off + (long long)k0 * g.lda;
    else {
      const int tap = k0 >> g.lCc, ch = k0 - (tap << g.lCc);  // uniform
      const int iy = v.i0 + (AF == A_G1 ? (tap >> 2) : -(tap >> 1)), ix = v.i1 + (AF == A_G1 ? (tap & 3) : -(tap & 1));
      *ok = (unsigned)iy < (unsigned)cg.IH && (unsigned)ix < (unsigned)cg.IW;
      return ((long long)(v.i2 + iy * cg.IW + ix) << g.lCc) + ch + v.i3;
    }
  };
  auto b_off = [&](const Inv& v, int k0, bool* ok) __attribute__((always_inline)) -> long long {
    *ok = true;
    if constexpr (BF == B_KC) return v.off + k0;
    else if constexpr (BF == B_KM) return v.off + (long long)k0 * g.ldb;
    else if constexpr (BF == B_G2) {  // k = output pixel (the contraction index)
      const int k = k0 + v.i3;
      const int ox = k & ((1 << cg.lOW) - 1), oy = (k & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = k >> cg.lOHW;
      const int iy = 2 * oy + v.i0, ix = 2 * ox + v.i1;
      *ok = (unsigned)iy < (unsigned)cg.IH && (unsigned)ix < (unsigned)cg.IW;
      return ((long long)((b * cg.IH + iy) * cg.IW + ix) << g.lCc) + v.i2;
    } else {  // B_G3W: row k = (tap, c) of the weight [C_in][(ky, kx, oc)]: the (uniform) tap picks the column block
      const int tap = k0 >> g.lCc;
      const int ky = 1 - par_y + 2 * (tap >> 1), kx = 1 - par_x + 2 * (tap & 1);
      return (long long)(k0 - (tap << g.lCc) + v.i3) * g.ldb + (long long)(ky * 4 + kx) * g.N + v.i2;
    }
  };
  const int grp = wave >> 2, idx = KALT ? (wave & 3) : wave;  // (grp: the ping-pong group, see the K loop)
  Inv inva[UA], invb[UB];
#pragma unroll
  for (int i = 0; i < UA; ++i) inva[i] = a_inv((idx + NI * i) % PPA);
#pragma unroll
  for (int i = 0; i < UB; ++i) invb[i] = b_inv((idx + NI * i) % PPB);
  auto dma3 = [&](const bf16r* base, long long ps, long long off, bool ok, int dst, int plane_bytes) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      // (outside the image: the 16 zero bytes of g_p3_zero, selected as an OFFSET so that the request stays one instruction)
      const bf16r* plane = base + (size_t)q * ps;
      const long long zoff = (long long)((uintptr_t)zero - (uintptr_t)plane) >> 1;
      const bf16r* src = plane + (ok ? off : zoff);
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(lds + dst + q * plane_bytes), 16, 0, 0);
    }
  };
#ifdef MV_P3_DBG
  const int dbg = g.dbg;
#else
  constexpr int dbg = 0;
#endif
  auto issue = [&](int buf, int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      const int pc = idx + NI * i;  // wave-uniform
      if ((PPA % NI == 0 || pc < PPA) && !(dbg & 2)) {
        bool ok;
        const long long off = a_off(inva[i], k0, &ok);
        dma3(g.A, g.psa, off, ok, buf * SB + pc * 1024, PLA);
      }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int pc = idx + NI * i;
      if ((PPB % NI == 0 || pc < PPB) && !(dbg & 4)) {
        bool ok;
        const long long off = b_off(invb[i], k0, &ok);
        dma3(g.B, g.psb, off, ok, buf * SB + 3 * PLA + pc * 1024, PLB);
      }
    }
  };

  // ---- fragments: lane l holds (tile row l & 15, k = 8 (l >> 4) .. + 7) of every 16-row block
  const int wm = (idx / WC) * WM, wn = (idx % WC) * WN;
  const int l15 = lane & 15, l4 = lane >> 4;
  // KC operand: ds_read_b128 at [row][chunk ^ h]: the lane part is the same for every 16-row block, plane and buffer
  const int kc_lane_off = l15 * 64 + ((l4 ^ p3_h(l15 >> 2)) << 4);
  // KM operand: two transposing reads per fragment (k rows 8 l4 + 4 t + (l15 >> 2), t = 0, 1); the lane part depends on the
  // 16-row block through the XOR swizzle, so one address register per (block, t), kept over the whole K loop
  unsigned kmA[A_IS_KM ? TM : 1][2], kmB[B_IS_KM ? TN : 1][2];
  auto km_lane_off = [&](int rows, int blk16, int t) __attribute__((always_inline)) -> unsigned {
    const int ROWB = rows * 2, CPR = rows / 16;
    const int kr = 4 * (2 * l4 + t) + (l15 >> 2);
    const int x = p3_km_swz(kr, CPR);
    return lds0 + kr * ROWB + ((blk16 ^ x) << 5) + (l15 & 3) * 8;
  };
  if constexpr (A_IS_KM) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int t = 0; t < 2; ++t) kmA[a][t] = km_lane_off(BM, (wm >> 4) + a, t);
  }
  if constexpr (B_IS_KM) {
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int t = 0; t < 2; ++t) kmB[b][t] = km_lane_off(BN, (wn >> 4) + b, t) + 3 * PLA;
  }
  // inline asm: behind the builtin hipcc waits for EVERY LDS-DMA in flight (s_waitcnt vmcnt(0)) before the read, which would
  // drain the tiles being prefetched; the wait for these reads is the explicit lgkmcnt(0) of sync()
#define MV_TR_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto km_pair = [&](unsigned a0, unsigned a1, auto qoff) __attribute__((always_inline)) -> p3_bf16x8 {
    p3_s16x4 h0, h1;
    MV_TR_READ(h0, a0, decltype(qoff)::value);
    MV_TR_READ(h1, a1, decltype(qoff)::value);
    p3_s16x8 v;
    v[0] = h0[0]; v[1] = h0[1]; v[2] = h0[2]; v[3] = h0[3];
    v[4] = h1[0]; v[5] = h1[1]; v[6] = h1[2]; v[7] = h1[3];
    return __builtin_bit_cast(p3_bf16x8, v);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- the K loop: two wave groups in PING-PONG.  A workgroup's waves are dealt to the four SIMDs cyclically, so wave w and
  // wave w + 4 share a SIMD; group 0 = waves 0-3, group 1 = waves 4-7.  A step of a wave is two phases, each closed by a
  // workgroup barrier: L (read the fragments of tile s from LDS, request this wave's share of tile s + 2 by LDS-DMA) and C
  // (the 6 x TM x TN MFMAs of tile s).  Group 1 runs ONE PHASE BEHIND group 0 (an extra barrier at its start), so on every SIMD
  // one wave multiplies while its partner loads: the LDS reads, the DMA requests (a wave stalls ~45 cycles in each while the
  // CU's address path is busy) and the MFMAs of a CU overlap instead of following each other -- with all eight waves in the
  // same phase they did not (tools/p3_dbg_sweep.sh: DMA 30 + reads 13 + barriers 3 + MFMA 23 us took 60 us for db1).
  //   interval 2s    : group 0 in L_s      group 1 in C_{s-1}
  //   interval 2s + 1: group 0 in C_s      group 1 in L_s
  // Three LDS buffers: tile s + 2 goes where tile s - 1 was; its last reader is group 1 in L_{s-1} = interval 2s - 1, every L
  // phase ends with lgkmcnt(0) before its barrier, and the earliest writer is group 0 in L_s = interval 2s.  Tile s + 1 is
  // complete before its first reader (group 0, interval 2s + 2): a wave leaves at most its newest tile in flight (counted
  // vmcnt) at the end of C_s (group 0) / L_s (group 1), both before the barrier that closes interval 2s + 1.
  struct Frags {
    p3_bf16x8 a[TM][3], b[TN][3];
  };
  auto load = [&](Frags& f, auto bufc) __attribute__((always_inline)) {
    constexpr int BUFOFF = decltype(bufc)::value * SB;
    if constexpr (!A_IS_KM) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          f.a[a][q] = *reinterpret_cast<const p3_bf16x8*>(lds + BUFOFF + q * PLA + ((wm >> 4) + a) * 1024 + kc_lane_off);
    } else {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const unsigned a0 = kmA[a][0] + BUFOFF, a1 = kmA[a][1] + BUFOFF;
        f.a[a][0] = km_pair(a0, a1, std::integral_constant<int, 0>{});
        f.a[a][1] = km_pair(a0, a1, std::integral_constant<int, PLA>{});
        f.a[a][2] = km_pair(a0, a1, std::integral_constant<int, 2 * PLA>{});
      }
    }
    if constexpr (!B_IS_KM) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          f.b[b][q] = *reinterpret_cast<const p3_bf16x8*>(lds + BUFOFF + 3 * PLA + q * PLB + ((wn >> 4) + b) * 1024 + kc_lane_off);
    } else {
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const unsigned b0 = kmB[b][0] + BUFOFF, b1 = kmB[b][1] + BUFOFF;
        f.b[b][0] = km_pair(b0, b1, std::integral_constant<int, 0>{});
        f.b[b][1] = km_pair(b0, b1, std::integral_constant<int, PLB>{});
        f.b[b][2] = km_pair(b0, b1, std::integral_constant<int, 2 * PLB>{});
      }
    }
  };
  // The L phase: the fragment reads (LDS-bound: 72 KB per group and phase) and the DMA requests (bound by the CU's address
  // path) use different units, but a wave issues in order -- written one after the other they took 436 + 448 cycles of a
  // 1100-cycle phase (tools/p3_phase_times.py) against ~900 for the MFMA phase of the partner group.  So they alternate: one
  // request, then the reads of one 16-row block, ...; sched_barrier keeps hipcc from re-clustering the reads.
  // request number d of this wave for the tile at the given offsets: (operand, piece, plane)
  auto dma1 = [&](int d, int fill_buf, const bool* oka, const long long* offa, const bool* okb, const long long* offb) __attribute__((always_inline)) {
    if (d < 3 * UA) {
      const int i = d / 3, q = d % 3, pc = idx + NI * i;
      if ((PPA % NI == 0 || pc < PPA) && !(dbg & 2)) {
        const bf16r* plane = g.A + (size_t)q * g.psa;
        const long long zoff = (long long)((uintptr_t)zero - (uintptr_t)plane) >> 1;
        __builtin_amdgcn_global_load_lds(plane + (oka[i] ? offa[i] : zoff),
                                         (__attribute__((address_space(3))) void*)(lds + fill_buf * SB + pc * 1024 + q * PLA), 16, 0, 0);
      }
    } else if (d < 3 * (UA + UB)) {
      const int e = d - 3 * UA, i = e / 3, q = e % 3, pc = idx + NI * i;
      if ((PPB % NI == 0 || pc < PPB) && !(dbg & 4)) {
        const bf16r* plane = g.B + (size_t)q * g.psb;
        const long long zoff = (long long)((uintptr_t)zero - (uintptr_t)plane) >> 1;
        __builtin_amdgcn_global_load_lds(plane + (okb[i] ? offb[i] : zoff),
                                         (__attribute__((address_space(3))) void*)(lds + fill_buf * SB + 3 * PLA + pc * 1024 + q * PLB), 16, 0, 0);
      }
    }
  };
  auto load_and_issue = [&](Frags& f, auto bufc, int fill_buf, int k0) __attribute__((always_inline)) {
    constexpr int BUFOFF = decltype(bufc)::value * SB;
    // addresses of this wave's pieces first (VALU only)
    bool oka[UA], okb[UB];
    long long offa[UA], offb[UB];
#pragma unroll
    for (int i = 0; i < UA; ++i) offa[i] = a_off(inva[i], k0, &oka[i]);
#pragma unroll
    for (int i = 0; i < UB; ++i) offb[i] = b_off(invb[i], k0, &okb[i]);
    constexpr int ND = 3 * (UA + UB), NPART = TM + TN;
    constexpr int PER = (ND + NPART - 1) / NPART;  // requests in front of each block's reads
    int d = 0;
#pragma unroll
    for (int part = 0; part < NPART; ++part) {
#pragma unroll
      for (int u = 0; u < PER; ++u)
        if (d < ND) dma1(d++, fill_buf, oka, offa, okb, offb);
      if (!(dbg & 64)) {
        if (part < TM) {
          const int a = part;
          if constexpr (!A_IS_KM) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
              f.a[a][q] = *reinterpret_cast<const p3_bf16x8*>(lds + BUFOFF + q * PLA + ((wm >> 4) + a) * 1024 + kc_lane_off);
          } else {
            const unsigned a0 = kmA[a][0] + BUFOFF, a1 = kmA[a][1] + BUFOFF;
            f.a[a][0] = km_pair(a0, a1, std::integral_constant<int, 0>{});
            f.a[a][1] = km_pair(a0, a1, std::integral_constant<int, PLA>{});
            f.a[a][2] = km_pair(a0, a1, std::integral_constant<int, 2 * PLA>{});
          }
        } else {
          const int b = part - TM;
          if constexpr (!B_IS_KM) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
              f.b[b][q] = *reinterpret_cast<const p3_bf16x8*>(lds + BUFOFF + 3 * PLA + q * PLB + ((wn >> 4) + b) * 1024 + kc_lane_off);
          } else {
            const unsigned b0 = kmB[b][0] + BUFOFF, b1 = kmB[b][1] + BUFOFF;
            f.b[b][0] = km_pair(b0, b1, std::integral_constant<int, 0>{});
            f.b[b][1] = km_pair(b0, b1, std::integral_constant<int, PLB>{});
            f.b[b][2] = km_pair(b0, b1, std::integral_constant<int, 2 * PLB>{});
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto mma = [&](const Frags& f) __attribute__((always_inline)) {
    // piece pairs from the smallest products up; operands swapped (B fragment first) so that a lane's four accumulator
    // values are four consecutive columns of one output row
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    if (dbg & 8) {  // (keep the fragments alive without the MFMAs)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) asm volatile("" ::"v"(f.a[a][q]));
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 3; ++q) asm volatile("" ::"v"(f.b[b][q]));
      return;
    }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[b][PB[t]], f.a[a][PA[t]], acc[a][b], 0, 0, 0);
  };
  // DMA instructions this wave issues per tile (what its counted wait leaves in flight)
  const int per_tile = 3 * ((PPA % NI == 0 ? UA : (idx < PPA ? 1 : 0)) + (PPB % NI == 0 ? UB : (idx < PPB ? 1 : 0)));
  auto barrier = [&]() __attribute__((always_inline)) {
    if (!(dbg & 32)) asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // (MFMAs touch no memory: the clobber alone would not keep them behind the barrier)
  };
  auto wait_dma = [&](bool all) __attribute__((always_inline)) {  // leave only the newest tile's requests in flight
    if (dbg & 1) return;
    if (all || per_tile == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (per_tile == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (per_tile == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const int nsteps = (ke - kb + 31) >> 5;
  auto tile_k = [&](int t) __attribute__((always_inline)) { return (t < nsteps && !(dbg & 16)) ? kb + 32 * t : kb; };  // (past the end: tile 0 again, never consumed)
  static_assert(NST == 3, "the ping-pong loop cycles three LDS buffers");
  Frags f;
  if constexpr (KALT) {
    issue(grp, tile_k(grp));  // each group fetches (and later consumes) its own tiles: grp, grp + 2, ...
  } else {
    issue(0, tile_k(0));
    issue(1, tile_k(1));
  }
  wait_dma(true);
  barrier();
  if (grp == 1) barrier();  // group 1 starts one phase late
#ifdef MV_P3_DBG
  // time stamps (s_memtime) of the phases of step 8 in waves 0 and 4 of workgroup (0, 0, 0): tools/p3_phase_times.py
#define MV_P3_STAMP(i) do { if (stamp) ts[i] = __builtin_readcyclecounter(); } while (0)
  unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#else
#define MV_P3_STAMP(i) do {} while (0)
#endif
  auto step = [&](int t, auto buf, auto buf_fill) __attribute__((always_inline)) {
#ifdef MV_P3_DBG
    const bool stamp = g.dbgbuf && t == 8 && L0 == 0 && (wave & 3) == 0;
#endif
    // L: fragments of tile t, requests of tile t + 2
    MV_P3_STAMP(0);
    load_and_issue(f, buf, decltype(buf_fill)::value, tile_k(t + 2));
    MV_P3_STAMP(1);
    MV_P3_STAMP(2);
    if (!KALT && grp == 1) wait_dma(false);
    MV_P3_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MV_P3_STAMP(4);
    barrier();
    MV_P3_STAMP(5);
    // C (MV_P3_PRIO: with priority over the partner wave's load phase on this SIMD)
#if MV_P3_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    mma(f);
#if MV_P3_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    MV_P3_STAMP(6);
    if (KALT) wait_dma(true);  // the group's next tile (requested in this step's L phase) before the barrier its L phase follows
    else if (grp == 0) wait_dma(false);
    MV_P3_STAMP(7);
    barrier();
    MV_P3_STAMP(8);
#ifdef MV_P3_DBG
    if (stamp && lane == 0)
      for (int i = 0; i < 9; ++i) g.dbgbuf[grp * 16 + i] = ts[i];
#endif
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  if constexpr (KALT) {
    // interval i: group i & 1 in L_i (fragments of tile i, requests of tile i + 2 into the buffer tile i - 1 left at the previous
    // barrier), the other group in C_{i-1}.  A group's tiles cycle the three buffers with period three of ITS steps.
    if (grp == 0) {
      for (int t = 0; t < nsteps; t += 6) {
        step(t, B0{}, B2{});
        if (t + 2 < nsteps) step(t + 2, B2{}, B1{});
        if (t + 4 < nsteps) step(t + 4, B1{}, B0{});
      }
    } else {
      for (int t = 1; t < nsteps; t += 6) {
        step(t, B1{}, B0{});
        if (t + 2 < nsteps) step(t + 2, B0{}, B2{});
        if (t + 4 < nsteps) step(t + 4, B2{}, B1{});
      }
    }
    if ((grp == 0) == ((nsteps & 1) == 0)) barrier();  // (the other group's last phase)
  } else {
    for (int t = 0; t < nsteps; t += 3) {
      step(t, B0{}, B2{});
      if (t + 1 < nsteps) step(t + 1, B1{}, B0{});
      if (t + 2 < nsteps) step(t + 2, B2{}, B1{});
    }
    if (grp == 0) barrier();  // (group 1's last phase)
  }
#undef MV_TR_READ

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the (unused) DMA of the last step
  // ---- KALT: the two groups hold partial sums of the SAME wave tiles (wave w and w + 4).  Each wave keeps one half of its
  // rows (group 0 the upper TM / 2 blocks, group 1 the lower), hands the other half to its partner through LDS and adds what
  // it receives: own + partner on both sides, so the result does not depend on which group computed which K steps' sum first
  // beyond the (fixed) even / odd partition.
  constexpr int TMK = KALT ? TM / 2 : TM;  // row blocks this wave writes
  if constexpr (KALT) {
    static_assert(!KALT || (TM % 2 == 0 && 8 * (TM / 2) * TN * 1024 <= NST * SB), "exchange area");
    barrier();  // every wave has left the K loop: the tile buffers are free
    f32x4* xch = reinterpret_cast<f32x4*>(lds) + (size_t)wave * (TMK * TN * 64) + lane;
    const f32x4* xin = reinterpret_cast<const f32x4*>(lds) + (size_t)(wave ^ 4) * (TMK * TN * 64) + lane;
    auto swap_half = [&](auto keep0, auto send0) __attribute__((always_inline)) {
      constexpr int K0 = decltype(keep0)::value, S0 = decltype(send0)::value;
#pragma unroll
      for (int a = 0; a < TMK; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) xch[(a * TN + b) * 64] = acc[S0 + a][b];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();
#pragma unroll
      for (int a = 0; a < TMK; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = acc[K0 + a][b] + xin[(a * TN + b) * 64];  // (kept half moved to blocks 0 .. TMK - 1)
    };
    if (grp == 0) swap_half(std::integral_constant<int, 0>{}, std::integral_constant<int, TM / 2>{});
    else swap_half(std::integral_constant<int, TM / 2>{}, std::integral_constant<int, 0>{});
  }
  const int row_half = KALT ? grp * (TM / 2) * 16 : 0;
  // ---- epilogue: lane holds row l15, columns 4 * l4 + r of every 16 x 16 tile
  float cs[TN][4];
#pragma unroll
  for (int a = 0; a < TMK; ++a) {
    int m = m0 + wm + row_half + a * 16 + l15;
    if (PARITY) {  // row of the parity class -> its pixel of the (2 IH) x (2 IW) output
      const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, bb = m >> cg.lOHW;
      m = (bb * 2 * cg.IH + 2 * oy + par_y) * 2 * cg.IW + 2 * ox + par_x;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + wn + b * 16 + l4 * 4;
      f32x4 v = acc[a][b];
      const size_t o = (size_t)m * g.ldc + n;
      if (g.bias) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if (g.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (g.mask) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(g.mask + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (mk[r] > 0.f) ? v[r] : 0.f;
      }
      if (g.C) *reinterpret_cast<f32x4*>(C + o) = v;
      if (g.Cp) store_planes4(g.Cp, g.psc, o, v[0], v[1], v[2], v[3]);
      if (g.colpart) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[b][r] = (a == 0) ? v[r] : cs[b][r] + v[r];
      }
    }
  }
  // ---- column sums of the tile (the bias gradient of the layer whose backward-data this is: its f32 result then need not be
  // written at all).  Rows of a wave: in-lane over its 16-row blocks, then the 16 lanes of a DPP row; waves (and, with alternate
  // K steps, the two groups' row halves): through LDS in a fixed order; row tiles / parity classes: by the caller's deferred sum.
  if (g.colpart) {
    constexpr int NS = KALT ? 2 * WR : WR;  // partial sums per column
    barrier();  // (every wave has left the K loop / the exchange: LDS is free)
    float* red = reinterpret_cast<float*>(lds);
    const int slot = KALT ? (idx / WC) * 2 + grp : idx / WC;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = p3_row16_sum(cs[b][r]);
        if (l15 == 0) red[slot * BN + wn + b * 16 + l4 * 4 + r] = t;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    if (tid < BN) {
      float t = red[tid];
#pragma unroll
      for (int w = 1; w < NS; ++w) t += red[w * BN + tid];
      g.colpart[(size_t)((PARITY ? bz * ny : 0) + m0 / BM) * g.N + n0 + tid] = t;
    }
  }
}

template <int BM, int BN, int WR, int AF, int BF, int NST = 3, bool KALT = false>
__global__ __launch_bounds__(512) void k_gemm_p3(const P3Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[p3_lds_bytes<BM, BN>(NST)];
  p3_body<BM, BN, WR, AF, BF, NST, KALT>(g, lds, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), (int)gridDim.x,
                                         (int)gridDim.y, (int)gridDim.z);
}

// Two INDEPENDENT contractions in one launch (a layer's weight gradient and its backward-data: both read the same incoming
// gradient, neither reads the other's result): workgroups 0 .. n1 - 1 are the first one's, the rest the second's.  A CU that
// finishes a workgroup of the first takes one of the second -- the epilogue of one (57 MB of stores for db1) and the prologue of
// the other (two tiles' DMA latency) run under the other's K loop instead of at a kernel boundary where the whole chip waits.
template <int BM1, int BN1, int WR1, int AF1, int BF1, bool KALT1, int BM2, int BN2, int WR2, int AF2, int BF2, bool KALT2>
__global__ __launch_bounds__(512) void k_gemm_p3_pair(const P3Args g1, const P3Args g2, const int n1, const int nx1, const int ny1,
                                                      const int nz1, const int nx2, const int ny2, const int nz2) {
  constexpr int LB = p3_lds_bytes<BM1, BN1>() > p3_lds_bytes<BM2, BN2>() ? p3_lds_bytes<BM1, BN1>() : p3_lds_bytes<BM2, BN2>();
  __shared__ __attribute__((aligned(16))) unsigned char lds[LB];
  const int L = blockIdx.x;
  if (L < n1) p3_body<BM1, BN1, WR1, AF1, BF1, 3, KALT1>(g1, lds, L, nx1, ny1, nz1);
  else p3_body<BM2, BN2, WR2, AF2, BF2, 3, KALT2>(g2, lds, L - n1, nx2, ny2, nz2);
}

// ---- planes of an existing f32 tensor (weights after the optimizer step; activations whose producer does not emit them)
__global__ __launch_bounds__(256) void k_split3(const SplitJobs jobs) { split3_body(jobs, (int)blockIdx.x, 256); }

static int split_jobs_build(SplitJobs* jobs, int njobs, const float* const* src, uint16_t* const* planes, const int64_t* n) {
  if (njobs < 1 || njobs > kMaxSplitJobs || !src || !planes || !n) return fail(MVAE_E_BADARG, "1 .. 12 jobs%s", "");
  jobs->njobs = njobs;
  jobs->blk0[0] = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!src[j] || !planes[j] || n[j] < 4 || (n[j] & 3)) return fail(MVAE_E_BADARG, "null pointer / n not a multiple of 4%s", "");
    if (!aligned16(src[j]) || ((uintptr_t)planes[j] & 7)) return fail(MVAE_E_ALIGN, "mvae_split3_planes: 16-byte aligned source, 8-byte aligned planes%s", "");
    jobs->src[j] = src[j];
    jobs->dst[j] = reinterpret_cast<bf16r*>(planes[j]);
    jobs->n[j] = n[j];
    const long long want = (n[j] / 4 + 256 * 4 - 1) / (256 * 4);  // ~4 vectors per thread
    jobs->blk0[j + 1] = jobs->blk0[j] + (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  }
  return 0;
}

// Queued planes: mvae_split3_planes_queue only records the jobs (host side, per calling thread); the next launch that can carry
// them as extra workgroups takes them (p3_take_splitjobs: the conv latent forward -- a latency-bound launch of one small
// workgroup per batch row, whose CUs have bandwidth to spare), and ANY plane contraction launched before that performs them
// first (p3_flush_split), so a consumer never sees planes that were not written.
static thread_local SplitJobs g_split;
static thread_local bool g_split_pending = false;
void p3_flush_split(hipStream_t s) {
  if (!g_split_pending) return;
  g_split_pending = false;
  hipLaunchKernelGGL(k_split3, dim3((unsigned)g_split.blk0[g_split.njobs]), dim3(256), 0, s, g_split);
}
bool p3_take_splitjobs(SplitJobs4* out) {
  if (!g_split_pending || g_split.njobs > kRideSplitJobs) return false;
  g_split_pending = false;
  out->njobs = g_split.njobs;
  for (int j = 0; j < g_split.njobs; ++j) {
    out->src[j] = g_split.src[j];
    out->dst[j] = g_split.dst[j];
    out->n[j] = g_split.n[j];
  }
  for (int j = 0; j <= g_split.njobs; ++j) out->blk0[j] = g_split.blk0[j];
  return true;
}
extern "C" int mvae_split3_planes_queue(int njobs, const float* const* src, uint16_t* const* planes, const int64_t* n, void* stream) {
  p3_flush_split((hipStream_t)stream);  // (a queue nobody took: performed now, in order)
  int rc = split_jobs_build(&g_split, njobs, src, planes, n);
  if (rc) return rc;
  g_split_pending = true;
  return 0;
}
extern "C" int mvae_split3_planes_flush(void* stream) {
  p3_flush_split((hipStream_t)stream);
  LAUNCH_CHECK("split3 planes launch");
  return 0;
}

extern "C" int mvae_split3_planes(int njobs, const float* const* src, uint16_t* const* planes, const int64_t* n, void* stream) {
  p3_flush_split((hipStream_t)stream);
  SplitJobs jobs;
  int rc = split_jobs_build(&jobs, njobs, src, planes, n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_split3, dim3((unsigned)jobs.blk0[njobs]), dim3(256), 0, (hipStream_t)stream, jobs);
  LAUNCH_CHECK("split3 planes launch");
  return 0;
}

// ---- entry points -------------------------------------------------------------------------------------------------------
static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}
static bool planes_ok(const void* p, long long ld, long long ps) { return p && ((uintptr_t)p & 15) == 0 && (ld & 7) == 0 && (ps & 7) == 0; }

#ifdef MV_P3_DBG
static unsigned long long* g_p3_dbgbuf = nullptr;
extern "C" int mvae_p3_debug_stamps(unsigned long long* out32) {  // (debug builds) the stamps of the last launch
  if (!g_p3_dbgbuf) return -1;
  return (int)hipMemcpy(out32, g_p3_dbgbuf, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif
#ifndef MV_P3_STAGES
#define MV_P3_STAGES 3
#endif
void p3_sum_slices(const float* part, float* out, int64_t n, int slices, hipStream_t s);      // mvae_conv.hip (honours deferral)
void p3_sum_slices_now(const float* part, float* out, int64_t n, int slices, hipStream_t s);  // mvae_conv.hip (immediate)

// ---- grouped launches (mvae_p3_group): between group(1) and group(0) the plane contractions are QUEUED; group(0) launches two
// queued ones that form a known pair as ONE kernel (k_gemm_p3_pair), anything else one after the other, then the slice sums that
// were waiting for them.
enum { P3_OTHER = 0, P3_WGRAD = 1, P3_DGRAD_CONV = 2, P3_NN = 3, P3_DGRAD_CONVT = 4 };
template <int BM, int BN, int WR, int AF, int BF, bool KALT>
constexpr int p3_cfg_id() {
  if (BM == 128 && BN == 128 && AF == A_KM && BF == B_G2 && KALT) return P3_WGRAD;
  if (BM == 128 && BN == 128 && AF == A_G1 && BF == B_KC && !KALT) return P3_DGRAD_CONV;
  if (BM == 128 && BN == 128 && AF == A_KC && BF == B_KM && !KALT) return P3_NN;
  if (BM == 128 && BN == 64 && AF == A_G3 && BF == B_G3W && KALT) return P3_DGRAD_CONVT;
  return P3_OTHER;
}
struct P3Queued {
  int id;
  P3Args a;
  dim3 grid;
  void (*single)(const P3Args&, dim3, hipStream_t);
};
struct P3Post { const float* part; float* out; int64_t n; int slices; bool deferrable; float* colws; };  // colws != NULL: a column sum of part [slices, n]
static thread_local bool g_p3_group = false;
static thread_local int g_p3_nq = 0, g_p3_npost = 0;
static thread_local P3Queued g_p3_q[2];
constexpr int kP3PostMax = 4;  // a pair queues at most: the weight gradient's slice sum + the data gradient's slice sum + a column sum
static thread_local P3Post g_p3_post[kP3PostMax];

template <int BM, int BN, int WR, int AF, int BF, bool KALT>
static void p3_single(const P3Args& a, dim3 grid, hipStream_t s) {
  p3_flush_split(s);
  hipLaunchKernelGGL((k_gemm_p3<BM, BN, WR, AF, BF, MV_P3_STAGES, KALT>), grid, dim3(512), 0, s, a);
}
template <int BM, int BN, int WR, int AF, int BF, bool KALT>
static void p3_submit(const P3Args& a, dim3 grid, hipStream_t s) {
  if (g_p3_group && g_p3_nq < 2) {
    g_p3_q[g_p3_nq++] = P3Queued{p3_cfg_id<BM, BN, WR, AF, BF, KALT>(), a, grid, &p3_single<BM, BN, WR, AF, BF, KALT>};
    return;
  }
  p3_single<BM, BN, WR, AF, BF, KALT>(a, grid, s);
}
// the immediate slice sum of a split-K backward-data result: after the (possibly queued) launch that writes the slices
static void p3_sum_after(const float* part, float* out, int64_t n, int slices, hipStream_t s) {
  if (g_p3_group && g_p3_npost < kP3PostMax) {
    g_p3_post[g_p3_npost++] = P3Post{part, out, n, slices, false, nullptr};
    return;
  }
  p3_sum_slices_now(part, out, n, slices, s);
}
// a deferrable slice sum (final gradients: bias column sums) of a result whose launch may still be queued
static void p3_sum_deferrable_after(const float* part, float* out, int64_t n, int slices, hipStream_t s) {
  if (g_p3_group && g_p3_npost < kP3PostMax) {
    g_p3_post[g_p3_npost++] = P3Post{part, out, n, slices, true, nullptr};
    return;
  }
  p3_sum_slices(part, out, n, slices, s);
}
template <int BM2, int BN2, int WR2, int AF2, int BF2, bool KALT2>
static void p3_launch_pair(const P3Queued& w, const P3Queued& d, hipStream_t s) {  // w: the weight gradient (P3_WGRAD)
  const int n1 = (int)(w.grid.x * w.grid.y * w.grid.z), n2 = (int)(d.grid.x * d.grid.y * d.grid.z);
  p3_flush_split(s);
  hipLaunchKernelGGL((k_gemm_p3_pair<128, 128, 2, A_KM, B_G2, true, BM2, BN2, WR2, AF2, BF2, KALT2>), dim3((unsigned)(n1 + n2)),
                     dim3(512), 0, s, w.a, d.a, n1, (int)w.grid.x, (int)w.grid.y, (int)w.grid.z, (int)d.grid.x, (int)d.grid.y,
                     (int)d.grid.z);
}
void p3_colsum_deferrable(const float* G, float* out, int64_t M, int N, float* ws, hipStream_t s);  // mvae_conv.hip
// the (deferrable) column sum of per-tile partial sums [rows, n] whose launch may still be queued
static void p3_colsum_after(float* part, float* out, int64_t rows, int n, float* ws, hipStream_t s) {
  if (g_p3_group && g_p3_npost < kP3PostMax) {
    g_p3_post[g_p3_npost++] = P3Post{part, out, n, (int)rows, true, ws};
    return;
  }
  p3_colsum_deferrable(part, out, rows, n, ws, s);
}
extern "C" int64_t mvae_conv_transpose_k4s2p1_nhwc_p3_colsum_floats(int B, int IH, int IW, int OC) {
  const int64_t rows = 4 * (((int64_t)B * IH * IW) / 128);
  return (rows + (rows + 511) / 512) * OC;
}
extern "C" int mvae_p3_group(int on, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (on) {
    g_p3_group = true;
    g_p3_nq = g_p3_npost = 0;
    return 0;
  }
  g_p3_group = false;
  static const bool pair_off = getenv("MVAE_P3_NO_PAIR") != nullptr;
  bool paired = false;
  if (g_p3_nq == 2 && !pair_off) {
    const int wi = g_p3_q[0].id == P3_WGRAD ? 0 : (g_p3_q[1].id == P3_WGRAD ? 1 : -1);
    if (wi >= 0 && (((g_p3_q[wi].grid.x * g_p3_q[wi].grid.y * g_p3_q[wi].grid.z) & 7) == 0)) {
      const P3Queued &w = g_p3_q[wi], &d = g_p3_q[1 - wi];
      paired = true;
      if (d.id == P3_DGRAD_CONV) p3_launch_pair<128, 128, 2, A_G1, B_KC, false>(w, d, s);
      else if (d.id == P3_NN) p3_launch_pair<128, 128, 2, A_KC, B_KM, false>(w, d, s);
      else if (d.id == P3_DGRAD_CONVT) p3_launch_pair<128, 64, 2, A_G3, B_G3W, true>(w, d, s);
      else paired = false;
    }
  }
  if (!paired)
    for (int i = 0; i < g_p3_nq; ++i) g_p3_q[i].single(g_p3_q[i].a, g_p3_q[i].grid, s);
  for (int i = 0; i < g_p3_npost; ++i) {
    if (g_p3_post[i].colws) p3_colsum_deferrable(g_p3_post[i].part, g_p3_post[i].out, g_p3_post[i].slices, (int)g_p3_post[i].n, g_p3_post[i].colws, s);
    else if (g_p3_post[i].deferrable) p3_sum_slices(g_p3_post[i].part, g_p3_post[i].out, g_p3_post[i].n, g_p3_post[i].slices, s);
    else p3_sum_slices_now(g_p3_post[i].part, g_p3_post[i].out, g_p3_post[i].n, g_p3_post[i].slices, s);
  }
  g_p3_nq = g_p3_npost = 0;
  LAUNCH_CHECK("grouped plane contraction launch");
  return 0;
}

template <int BM, int BN, int WR, int AF, int BF>
static void launch_p3(const P3Args& a0, int zdim, hipStream_t s) {
  // alternating K steps where they measured faster (tools/bench_p3.py, us with / without: weight gradients 47.3 / 48.7, 48.4 /
  // 53.9, transposed convolution 30.5 / 34.3) -- not for the gathered / plain KC forms (56.0 / 49.9, 34.8 / 32.9, 49.7 / 48.0);
  // MVAE_P3_KALT=0 / 1 forces one form for all
  static const int kalt_env = getenv("MVAE_P3_KALT") ? atoi(getenv("MVAE_P3_KALT")) : -1;
  const bool kalt = kalt_env >= 0 ? kalt_env != 0 : (AF == A_KM || AF == A_G3);
  P3Args a = a0;
#ifdef MV_P3_DBG
  const char* e = getenv("MV_P3_DBG");
  a.dbg = e ? atoi(e) : 0;
  static unsigned long long* dbuf = nullptr;
  if (!dbuf) { (void)hipMalloc(&dbuf, 32 * sizeof(unsigned long long)); (void)hipMemset(dbuf, 0, 32 * sizeof(unsigned long long)); }
  a.dbgbuf = dbuf;
  g_p3_dbgbuf = dbuf;
#endif
  dim3 grid(a.N / BN, a.M / BM, zdim);
  static const bool xcd_off = getenv("MVAE_P3_NO_XCD") != nullptr;
  const int T = (int)(grid.x * grid.y);
  const bool per_class = AF == A_G3 || zdim == 1;
  a.xcd = !xcd_off && (per_class ? (T % 8 == 0) : (zdim >= 8 ? (zdim % 8 == 0) : ((zdim == 2 || zdim == 4) && T % (8 / zdim) == 0)));
  if (a.xcd && AF != A_G3 && zdim == 1) {  // the gx that minimises the fabric bytes gx |A| + (8 / gx) |B| (a gathered image is K / 4 wide)
    double best = 0;
    a.xcd = 0;
    for (int gx = 1; gx <= 8; gx *= 2) {
      if (grid.x % gx || grid.y % (8 / gx)) continue;
      const double cost = (double)gx * a.M * (AF == A_G1 ? 0.25 : 1.0) + (8.0 / gx) * a.N;
      if (a.xcd == 0 || cost < best) { a.xcd = gx; best = cost; }
    }
  }
  if (kalt) p3_submit<BM, BN, 2, AF, BF, true>(a, grid, s);
  else p3_submit<BM, BN, WR, AF, BF, false>(a, grid, s);
}

// Which shapes the plane kernels take (the callers fall back to the f32-operand kernels otherwise): whole tiles only.
//   form 0  conv backward-data   (M = B OH OW output pixels, N = OC, K = 16 C)        mvae_conv_k4s2p1_nhwc_p3
//   form 1  NN product           (M rows, N columns, K)                               mvae_gemm_nn_p3
//   form 2  transposed conv      (M = B IH IW pixels per parity class, N = OC, K = 4 C)  mvae_conv_transpose_k4s2p1_nhwc_p3
//   form 3  conv weight gradient (M = B OH OW contracted rows, N = OC, K = 16 C columns) mvae_conv_k4s2p1_nhwc_wgrad_p3
extern "C" int mvae_p3_supported(int form, int64_t M, int N, int K, int Cc) {
  if (M < 256 || (M & 127) || M > 0x7fffffff || N < 64 || K < 32) return 0;
  const bool cpow = Cc >= 16 && ilog2_exact(Cc) >= 0;
  switch (form) {
    case 0: return (N % 128 == 0) && (K % 32 == 0) && cpow && Cc >= 32;
    case 1: return (N % 128 == 0) && (K % 32 == 0);
    case 2: return (N % 64 == 0) && (K % 32 == 0) && cpow && Cc >= 32;
    case 3: return (N % 128 == 0) && (K % 128 == 0) && (M % 256 == 0) && cpow;
    default: return 0;
  }
}

// Split-K slices of the gathered backward-data contraction: whole 128 x 128 tiles, >= ~256 workgroups, <= 8 slices of whole K steps
static int p3_conv_slices(int64_t M, int OC, int K, bool has_mask, int* kps) {
  *kps = K;
  if (has_mask) return 1;
  const int64_t tiles = (M / 128) * (OC / 128);
  if (tiles >= 192 || K < 1024) return 1;
  // (128, not 256: the sliced backward-data shares its launch with a weight gradient -- dt0: 4 slices of 32 K steps instead of 8
  // of 16, half the partial results: 0.656 -> 0.650 ms per step)
  static const int target = getenv("MVAE_P3_DGRAD_WGS") ? atoi(getenv("MVAE_P3_DGRAD_WGS")) : 128;
  int slices = (int)((target + tiles - 1) / tiles);
  if (slices > 8) slices = 8;
  *kps = ((K + slices - 1) / slices + 31) & ~31;
  return (K + *kps - 1) / *kps;
}
extern "C" int64_t mvae_conv_k4s2p1_nhwc_p3_workspace_floats(int B, int Cc, int IH, int IW, int OC, int has_mask) {
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  int kps;
  const int slices = p3_conv_slices(M, OC, 16 * Cc, has_mask != 0, &kps);  // (has_mask: any epilogue -- mask, bias or ReLU)
  return slices > 1 ? (int64_t)slices * M * OC : 0;
}


static int p3_geom(ConvGeom* g, int* lCc, int B, int Cc, int IH, int IW, bool out_is_half) {
  const int l = ilog2_exact(Cc);
  if (B < 1 || l < 0 || IH < 2 || IW < 2 || (IH & (IH - 1)) || (IW & (IW - 1)))
    return fail(MVAE_E_UNSUPPORTED, "plane contractions need power-of-two channels and extents%s (%lld)", "", Cc);
  int lOW = 0, lOH = 0;
  const int OW = out_is_half ? IW / 2 : IW, OH = out_is_half ? IH / 2 : IH;
  while ((1 << lOW) < OW) ++lOW;
  while ((1 << lOH) < OH) ++lOH;
  *g = ConvGeom{Cc, IH, IW, lOW, lOW + lOH};
  *lCc = l;
  return 0;
}

// y[(b,oy,ox), oc] = sum_{ky,kx,c} src[b, 2oy-1+ky, 2ox-1+kx, c] Wt[oc, (ky,kx,c)], zeroed where mask <= 0: the backward-data of a
// ConvTranspose2d (conv_vae.py:52-55) as mvae_conv_k4s2p1_nhwc computes it, on the planes of src [B*IH*IW, C] and of Wt
// [OC, 16 C]; y f32 (+ its planes when y_planes != NULL).
extern "C" int mvae_conv_k4s2p1_nhwc_p3(const uint16_t* src_planes, int64_t src_ps, const uint16_t* Wt_planes, int64_t w_ps,
                                        const float* mask, const float* bias, int relu, float* y, uint16_t* y_planes,
                                        int64_t y_ps, float* colsum_out, float* colsum_part, int B, int Cc, int IH, int IW,
                                        int OC, float* workspace, void* stream) {
  if (!src_planes || !Wt_planes) return fail(MVAE_E_BADARG, "null pointer%s", "");
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  const int K = 16 * Cc;
  if (!mvae_p3_supported(0, M, OC, K, Cc)) return fail(MVAE_E_UNSUPPORTED, "mvae_conv_k4s2p1_nhwc_p3: whole 128 x 128 tiles only%s", "");
  P3Args a{};
  int rc = p3_geom(&a.cg, &a.lCc, B, Cc, IH, IW, true);
  if (rc) return rc;
  if (!planes_ok(src_planes, Cc, src_ps) || !planes_ok(Wt_planes, K, w_ps) || (y && !aligned16(y)) || (mask && !aligned16(mask)) ||
      (bias && !aligned16(bias)) || (y_planes && !planes_ok(y_planes, OC, y_ps)))
    return fail(MVAE_E_ALIGN, "plane operands must be 16-byte aligned%s", "");
  const bool fwd = bias != nullptr || relu != 0;  // a layer's forward pass: the epilogue needs the whole sum, no K slices
  int kps;
  const int slices = (workspace && !fwd) ? p3_conv_slices(M, OC, K, mask != nullptr, &kps) : 1;
  a.bias = bias; a.relu = relu;
  // y == NULL: the K slices stay in the workspace for a consumer that adds them itself (mvae_conv_latent_backward's dt0), or
  // (one slice) only the planes and the column sums of the result are wanted
  if (!y && slices < 2 && !(y_planes && colsum_out)) return fail(MVAE_E_BADARG, "y may only be NULL with K slices or planes + column sums%s", "");
  if ((colsum_out == nullptr) != (colsum_part == nullptr) || (colsum_out && slices > 1))
    return fail(MVAE_E_BADARG, "colsum_out and colsum_part go together (unsliced calls only)%s", "");
  a.A = src_planes; a.lda = Cc; a.psa = src_ps;
  a.B = Wt_planes; a.ldb = K; a.psb = w_ps;
  a.ldc = OC; a.M = (int)M; a.N = OC; a.K = K;
  if (slices > 1) {
    if (!aligned16(workspace)) return fail(MVAE_E_ALIGN, "workspace must be 16-byte aligned%s", "");
    a.C = workspace; a.Cp = nullptr; a.psc = 0; a.mask = nullptr; a.k_per_slice = kps; a.slice_stride = M * OC;
    launch_p3<128, 128, 2, A_G1, B_KC>(a, slices, (hipStream_t)stream);
    if (y_planes) return fail(MVAE_E_UNSUPPORTED, "planes of a split-K result are not produced%s", "");
    if (y) p3_sum_after(workspace, y, M * OC, slices, (hipStream_t)stream);  // (an intermediate: the next launch reads it)
  } else {
    a.C = y; a.Cp = y_planes; a.psc = y_ps; a.mask = mask; a.k_per_slice = K; a.slice_stride = 0; a.colpart = colsum_part;
    // 128 x 64 tiles where 128 x 128 ones would leave CUs without a workgroup (the forward layers: 128 tiles each)
    if ((M / 128) * (OC / 128) >= 192) launch_p3<128, 128, 2, A_G1, B_KC>(a, 1, (hipStream_t)stream);
    else launch_p3<128, 64, 4, A_G1, B_KC>(a, 1, (hipStream_t)stream);
    if (colsum_out) p3_sum_deferrable_after(colsum_part, colsum_out, OC, (int)(M / 128), (hipStream_t)stream);
  }
  LAUNCH_CHECK("plane conv launch");
  return 0;
}

// out[M, N] = G[M, K] W[K, N] on the planes of G (K contiguous) and W (N contiguous): the product of a Conv2d backward-data
// that mvae_col2im_k4s2p1 folds (mvae_gemm_nn).
extern "C" int mvae_gemm_nn_p3(const uint16_t* G_planes, int64_t g_ps, const uint16_t* W_planes, int64_t w_ps, float* out,
                               int64_t M, int K, int N, void* stream) {
  if (!G_planes || !W_planes || !out) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (!mvae_p3_supported(1, M, N, K, 0)) return fail(MVAE_E_UNSUPPORTED, "mvae_gemm_nn_p3: whole 128 x 128 tiles only%s", "");
  if (!planes_ok(G_planes, K, g_ps) || !planes_ok(W_planes, N, w_ps) || !aligned16(out))
    return fail(MVAE_E_ALIGN, "plane operands must be 16-byte aligned%s", "");
  P3Args a{};
  a.A = G_planes; a.lda = K; a.psa = g_ps;
  a.B = W_planes; a.ldb = N; a.psb = w_ps;
  a.C = out; a.ldc = N; a.Cp = nullptr; a.mask = nullptr;
  a.M = (int)M; a.N = N; a.K = K; a.k_per_slice = K; a.slice_stride = 0;
  launch_p3<128, 128, 2, A_KC, B_KM>(a, 1, (hipStream_t)stream);
  LAUNCH_CHECK("plane NN product launch");
  return 0;
}

// The transposed convolution per output parity class (mvae_conv_transpose_k4s2p1_nhwc) on planes: src [B*IH*IW, C], Wt [C, 16 OC]
// (columns (ky, kx, oc)); y [B * 2IH * 2IW, OC] f32 (+ planes), zeroed where mask <= 0.  A Conv2d's backward-data.
extern "C" int mvae_conv_transpose_k4s2p1_nhwc_p3(const uint16_t* src_planes, int64_t src_ps, const uint16_t* Wt_planes,
                                                  int64_t w_ps, const float* mask, const float* bias, int relu, float* y,
                                                  uint16_t* y_planes, int64_t y_ps, float* colsum_out, float* colsum_ws, int B,
                                                  int Cc, int IH, int IW, int OC, void* stream) {
  if (!src_planes || !Wt_planes) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (!y && !(y_planes && colsum_out)) return fail(MVAE_E_BADARG, "y may only be NULL with planes + column sums%s", "");
  if ((colsum_out == nullptr) != (colsum_ws == nullptr)) return fail(MVAE_E_BADARG, "colsum_out and colsum_ws go together%s", "");
  const int64_t M = (int64_t)B * IH * IW;
  const int K = 4 * Cc;
  if (!mvae_p3_supported(2, M, OC, K, Cc)) return fail(MVAE_E_UNSUPPORTED, "mvae_conv_transpose_k4s2p1_nhwc_p3: whole tiles only%s", "");
  P3Args a{};
  int rc = p3_geom(&a.cg, &a.lCc, B, Cc, IH, IW, false);
  if (rc) return rc;
  if (!planes_ok(src_planes, Cc, src_ps) || !planes_ok(Wt_planes, 16 * OC, w_ps) || (y && !aligned16(y)) || (mask && !aligned16(mask)) ||
      (colsum_ws && !aligned16(colsum_ws)) ||
      (bias && !aligned16(bias)) || (y_planes && !planes_ok(y_planes, OC, y_ps)) || 4 * M > 0x7fffffff)
    return fail(MVAE_E_ALIGN, "plane operands must be 16-byte aligned%s", "");
  a.A = src_planes; a.lda = Cc; a.psa = src_ps;
  a.B = Wt_planes; a.ldb = (long long)16 * OC; a.psb = w_ps;
  a.C = y; a.ldc = OC; a.Cp = y_planes; a.psc = y_ps; a.mask = mask; a.bias = bias; a.relu = relu;
  a.M = (int)M; a.N = OC; a.K = K; a.k_per_slice = K; a.slice_stride = 0;
  // 128 x 128 tiles only where they still give every CU a workgroup
  a.colpart = colsum_ws;  // [4 parity classes x M / 128 row tiles][OC], then the column sum's own slice partials
  if (OC % 128 == 0 && (M / 128) * (OC / 128) * 4 >= 256) launch_p3<128, 128, 2, A_G3, B_G3W>(a, 4, (hipStream_t)stream);
  else launch_p3<128, 64, 4, A_G3, B_G3W>(a, 4, (hipStream_t)stream);
  if (colsum_out) {
    const int64_t rows = 4 * (M / 128);
    p3_colsum_after(colsum_ws, colsum_out, rows, OC, colsum_ws + rows * OC, (hipStream_t)stream);
  }
  LAUNCH_CHECK("plane transposed conv launch");
  return 0;
}

// dWt[oc, (ky,kx,c)] = sum_{b,oy,ox} dy[(b,oy,ox), oc] src[b, 2oy-1+ky, 2ox-1+kx, c] on the planes of dy [B*OH*OW, OC] and src
// [B*IH*IW, C] (mvae_conv_k4s2p1_nhwc_wgrad).  The rows are cut into slices (>= 256 workgroups) whose partial products
// are added in index order; workspace = mvae_conv_k4s2p1_nhwc_wgrad_p3_workspace_floats floats.
static int p3_wgrad_slices(int64_t M, int NP, int NQ, int* kps) {
  const int wg = (NP / 128) * (NQ / 128);
  // 128 workgroups, not one per CU: a weight gradient leaves in ONE launch with its layer's backward-data (mvae_p3_group), whose
  // 256-512 workgroups fill the other CUs; half the row slices are half the partial results to write and to add (67 -> 34 MB per
  // step) and twice the K loop per workgroup.  Step, same box: 0.657 (256) / 0.663 (192) / 0.644 (128) / 0.700 (64) ms.
  static const int target = getenv("MVAE_P3_WGRAD_WGS") ? atoi(getenv("MVAE_P3_WGRAD_WGS")) : 128;
  int slices = (target + wg - 1) / wg;
  const int max_slices = (int)(M / 256);
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  *kps = (int)((((M + slices - 1) / slices) + 31) & ~(int64_t)31);
  return (int)((M + *kps - 1) / *kps);
}
extern "C" int64_t mvae_conv_k4s2p1_nhwc_wgrad_p3_workspace_floats(int B, int Cc, int IH, int IW, int OC) {
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  int kps;
  const int slices = p3_wgrad_slices(M, OC, 16 * Cc, &kps);
  return slices > 1 ? (int64_t)slices * OC * 16 * Cc : 0;
}
extern "C" int mvae_conv_k4s2p1_nhwc_wgrad_p3(const uint16_t* dy_planes, int64_t dy_ps, const uint16_t* src_planes, int64_t src_ps,
                                              float* dWt, int B, int Cc, int IH, int IW, int OC, float* workspace, void* stream) {
  if (!dy_planes || !src_planes || !dWt) return fail(MVAE_E_BADARG, "null pointer%s", "");
  const int64_t M = (int64_t)B * (IH / 2) * (IW / 2);
  const int NQ = 16 * Cc;
  if (!mvae_p3_supported(3, M, OC, NQ, Cc))
    return fail(MVAE_E_UNSUPPORTED, "mvae_conv_k4s2p1_nhwc_wgrad_p3: whole 128 x 128 tiles, rows a multiple of 256%s", "");
  P3Args a{};
  int rc = p3_geom(&a.cg, &a.lCc, B, Cc, IH, IW, true);
  if (rc) return rc;
  if (!planes_ok(dy_planes, OC, dy_ps) || !planes_ok(src_planes, Cc, src_ps) || !aligned16(dWt))
    return fail(MVAE_E_ALIGN, "plane operands must be 16-byte aligned%s", "");
  int kps;
  const int slices = p3_wgrad_slices(M, OC, NQ, &kps);
  if (slices > 1 && (!workspace || !aligned16(workspace))) return fail(MVAE_E_BADARG, "the weight gradient needs its workspace%s", "");
  const int64_t n = (int64_t)OC * NQ;
  a.A = dy_planes; a.lda = OC; a.psa = dy_ps;
  a.B = src_planes; a.ldb = Cc; a.psb = src_ps;
  a.C = slices > 1 ? workspace : dWt; a.ldc = NQ; a.Cp = nullptr; a.mask = nullptr;
  a.M = OC; a.N = NQ; a.K = (int)M; a.k_per_slice = kps; a.slice_stride = n;
  launch_p3<128, 128, 2, A_KM, B_G2>(a, slices, (hipStream_t)stream);
  // inside mvae_p3_group(1) the contraction above is only QUEUED: its slice sum joins the group's post queue (and runs, or is
  // deferred, after group(0) has launched the contraction) -- summed right here it would read the workspace before it is written
  if (slices > 1) p3_sum_deferrable_after(workspace, dWt, n, slices, (hipStream_t)stream);
  LAUNCH_CHECK("plane weight gradient launch");
  return 0;
}
