// mvae_f32pp.hip -- the FORWARD contractions of the conv architecture (conv_vae.py:57-79) on the exact f32-input MFMA, in the
// ping-pong structure of mvae_p3.hip.
//
// In contraction mode 2 (and 0) every forward contraction -- whose output decides a ReLU mask or is the logits -- multiplies
// on v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered f32 accumulation.  k_gemm_tiled (mvae_conv.hip) does that at
// 50-68 % of the f32 MFMA peak: per K step its eight waves stage through registers, meet at a barrier, read fragments, multiply,
// meet again -- all in the same phase, so the MFMA pipe idles through every staging phase.  Here:
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), three LDS stages, no staging registers;
//   * two wave groups in ping-pong (waves w and w + 4 share a SIMD): one group reads fragments and requests DMA while the other
//     multiplies, phases closed by workgroup barriers, group 1 one phase behind (the scheme and its hazards: mvae_p3.hip);
//   * LDS images: K-contiguous operand [rows][32 k] f32 = 128-byte rows (full cache lines per DMA row), 16-byte chunk c of row
//     r at chunk c ^ ((r >> 1) & 7) -- conflict-free for ds_read_b128 under its 16-lane service groups; contraction-major
//     operand [32 k][cols] (the weight of an NN product / of the transposed convolution), 64-byte chunk c of k row kr at
//     c ^ ((kr >> 2) & mask), read one float per lane (ds_read_b32: 4 k rows x 16 columns per instruction, the two rows of a
//     32-lane service group in different bank halves).
// ARITHMETIC ORDER = k_gemm_tiled's: per output element the same sequence of MFMA steps (K step 32; kk = 0, 16; component j =
// 0..3; inside an MFMA the hardware's k = kk + 4 g + j, g = 0..3), so the results are BIT-IDENTICAL to that kernel's -- the
// forward pass does not change by a bit, only its duration (tests/test_conv_gpu.py::test_forward_pingpong_is_bit_identical).
// The epilogue (bias, ReLU, mask, optional bf16 planes of the result) is k_gemm_tiled's.
#include <type_traits>

#include "mvae_common.hpp"
#include "mvae_p3.hpp"

#ifndef MV_F32PP_PRIO
#define MV_F32PP_PRIO 1  // s_setprio(1) around the MFMA phase
#endif
enum { FA_KC = 0, FA_G1 = 1, FA_G3 = 2 };
enum { FB_KC = 0, FB_KM = 1, FB_G3W = 3 };

struct F32Args {
  const float* A; long long lda;
  const float* B; long long ldb;
  float* C; long long ldc;
  bf16r* Cp; long long psc;  // planes of the result (NULL: none)
  const float* bias;
  const float* mask;
  int relu;
  int M, N, K;
  ConvGeom cg;
  int lCc;
  int gx;  // XCD-aware tile order: the 8 XCDs as gx column groups x 8 / gx row groups (0: dispatch order)
  int dbg; // (-DMV_F32PP_DBG builds) bit 1: no DMA, bit 3: no MFMAs, bit 6: no fragment reads (env MV_F32PP_DBG)
};

__device__ __attribute__((aligned(16))) unsigned int g_f32pp_zero[4];
#ifdef MV_F32PP_DBG
// (debug builds) cycle stamps of the phases of K step 8 in waves 0 and 4 of workgroup 0: tools/f32pp_phase_times.py
__device__ unsigned long long g_f32pp_stamps[32];
extern "C" int mvae_f32pp_debug_stamps(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_f32pp_stamps), 32 * sizeof(unsigned long long));
}
#define MV_F32PP_STAMP(i) do { if (stamp) ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MV_F32PP_STAMP(i) do {} while (0)
#endif

template <int BM, int BN, int WR, int AF, int BF, int KS = 32>
__global__ __launch_bounds__(512) void k_gemm_f32pp(const F32Args g) {
  constexpr int NW = 8, WC = NW / WR, WM = BM / WR, WN = BN / WC, TM = WM / 16, TN = WN / 16;
  // KS = K extent of an LDS stage (32 or 64): with 64 a phase carries twice the MFMA work per barrier (the ~200-cycle barrier
  // skew is 20 % of a 128 x 64 x 32 phase); the order of MFMA steps per output element is the same
  constexpr int RB = KS * 4, LPR = RB / 16, RPP = 1024 / RB, NH = KS / 16;  // K-contiguous rows: bytes, lanes per row, rows per piece
  constexpr int PLA = BM * RB, PLB = BN * RB;    // bytes per operand tile
  constexpr int SB = PLA + PLB, NST = 3;
  constexpr int PPA = PLA / 1024, PPB = PLB / 1024;  // 1-KiB DMA pieces per tile
  constexpr int UA = PPA / NW, UB = PPB / NW;
  constexpr bool B_IS_KM = BF != FB_KC, PARITY = AF == FA_G3;
  static_assert(PPA % NW == 0 && PPB % NW == 0 && TM >= 1 && TN >= 1, "tile");
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * SB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  // XCD-aware tile order.  Workgroup number L of a launch runs on XCD L % 8 (observed; only speed depends on it) and each XCD has
  // its own L2.  In dispatch order the column tiles of one row block sit on 8 different XCDs, so every L2 fetches all of A
  // (e2 forward: 8 x 8.4 MB for 12.6 MB of operands).  With the XCDs arranged as gx column groups x gy = 8 / gx row groups, XCD
  // (xg, yg) runs the tiles (row block = yg + gy * rl, column tile = xg * cn + cl): A crosses the fabric gx times, the weight gy
  // times (the host picks the gx that minimises gx |A| + gy |W|).
  int tx = blockIdx.x, ty = blockIdx.y;
  if (g.gx > 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, j = lin >> 3;
    const int gy = 8 / g.gx, cn = gridDim.x / g.gx;
    tx = (xcd % g.gx) * cn + j % cn;
    ty = xcd / g.gx + gy * (j / cn);
  }
  const int m0 = ty * BM, n0 = tx * BN;
  const int par_y = PARITY ? (int)(blockIdx.z >> 1) : 0, par_x = PARITY ? (int)(blockIdx.z & 1) : 0;
  const ConvGeom cg = g.cg;
  const float* zero = reinterpret_cast<const float*>(g_f32pp_zero);

  // ---- LDS-DMA geometry: wave w moves pieces w, w + 8, ... of each operand tile; what does not depend on the K step is kept
  struct Inv { long long off; int i0, i1, i2, i3; };
  // K-contiguous rows of RB bytes: 16-byte chunk c of row r sits at slot c ^ swz(r), swz = (r >> 1) & 7 for 128-byte rows (two
  // rows per 256-byte bank row), r & 15 for 256-byte rows: conflict-free for ds_read_b128 under its 16-lane service groups
  auto kc_swz = [&](int r) __attribute__((always_inline)) { return KS == 32 ? ((r >> 1) & 7) : (r & 15); };
  auto kc_lane = [&](int pc, int* r, int* c4) __attribute__((always_inline)) {
    *r = pc * RPP + lane / LPR;
    *c4 = ((lane % LPR) ^ kc_swz(*r)) * 4;  // source chunk = slot ^ swz(r), in floats
  };
  auto a_inv = [&](int pc) __attribute__((always_inline)) -> Inv {
    Inv v{0, 0, 0, 0, 0};
    int r, c4;
    kc_lane(pc, &r, &c4);
    const int m = m0 + r;
    if constexpr (AF == FA_KC) {
      v.off = (long long)m * g.lda + c4;
    } else {
      const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, b = m >> cg.lOHW;
      v.i0 = AF == FA_G1 ? 2 * oy - 1 : oy + par_y;
      v.i1 = AF == FA_G1 ? 2 * ox - 1 : ox + par_x;
      v.i2 = b * cg.IH * cg.IW;
      v.i3 = c4;
    }
    return v;
  };
  auto b_inv = [&](int pc) __attribute__((always_inline)) -> Inv {
    Inv v{0, 0, 0, 0, 0};
    if constexpr (!B_IS_KM) {
      int r, c4;
      kc_lane(pc, &r, &c4);
      v.off = (long long)(n0 + r) * g.ldb + c4;
    } else {
      constexpr int ROWB = BN * 4, NCH = ROWB / 64;  // contraction-major tile [KS k][BN]: KS rows of BN floats
      const int off = pc * 1024 + lane * 16, kr = off / ROWB, inrow = off % ROWB;
      const int col = (((inrow >> 6) ^ ((kr >> 2) & (NCH - 1))) << 4) + ((inrow >> 4) & 3) * 4;
      if constexpr (BF == FB_KM) {
        v.off = (long long)kr * g.ldb + n0 + col;
      } else {
        v.i2 = n0 + col;
        v.i3 = kr;
      }
    }
    return v;
  };
  auto a_off = [&](const Inv& v, int k0, bool* ok) __attribute__((always_inline)) -> long long {
    *ok = true;
    if constexpr (AF == FA_KC) return v.off + k0;
    else {
      const int tap = k0 >> g.lCc, ch = k0 - (tap << g.lCc);  // uniform
      const int iy = v.i0 + (AF == FA_G1 ? (tap >> 2) : -(tap >> 1)), ix = v.i1 + (AF == FA_G1 ? (tap & 3) : -(tap & 1));
      *ok = (unsigned)iy < (unsigned)cg.IH && (unsigned)ix < (unsigned)cg.IW;
      return ((long long)(v.i2 + iy * cg.IW + ix) << g.lCc) + ch + v.i3;
    }
  };
  auto b_off = [&](const Inv& v, int k0, bool* ok) __attribute__((always_inline)) -> long long {
    *ok = true;
    if constexpr (BF == FB_KC) return v.off + k0;
    else if constexpr (BF == FB_KM) return v.off + (long long)k0 * g.ldb;
    else {  // FB_G3W: row k = (tap, c) of the weight [C_in][(ky, kx, oc)]: the (uniform) tap picks the column block
      const int tap = k0 >> g.lCc;
      const int ky = 1 - par_y + 2 * (tap >> 1), kx = 1 - par_x + 2 * (tap & 1);
      return (long long)(k0 - (tap << g.lCc) + v.i3) * g.ldb + (long long)(ky * 4 + kx) * g.N + v.i2;
    }
  };
  Inv inva[UA], invb[UB];
#pragma unroll
  for (int i = 0; i < UA; ++i) inva[i] = a_inv(wave + NW * i);
#pragma unroll
  for (int i = 0; i < UB; ++i) invb[i] = b_inv(wave + NW * i);
  // The address of a piece's 16 bytes per lane.  The f32-input MFMA of the partner wave leaves the load phase almost no issue
  // slots (tools/f32pp_phase_times.py: a load phase with NOTHING in it but its address arithmetic takes 1004 cycles next to the
  // partner's 1068-cycle MFMA phase, 464 without it: unlike the bf16 MFMA, the 8-pass f32 MFMA does not co-issue with another
  // wave's VALU work), so every VALU instruction of this phase is step time.  Hence: the full gather arithmetic (pixel, bounds,
  // 64-bit address: ~14 VALU per piece, several of them 64-bit) runs only when the K step enters a new TAP; within a tap the
  // source advances by KS floats, one select + one 64-bit add per piece.
  const float* pa[UA];
  bool oka[UA];
  const float* pb[UB];
  auto dma = [&](const float* src, int dst) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(lds + dst), 16, 0, 0);
  };
  auto issue = [&](int buf, int k0) __attribute__((always_inline)) {
    const bool fresh = AF == FA_KC ? true : ((k0 & ((1 << g.lCc) - 1)) == 0);  // uniform
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      if (fresh) {
        bool ok;
        const long long off = a_off(inva[i], k0, &ok);
        oka[i] = ok;
        pa[i] = ok ? g.A + off : zero;
      } else {
        pa[i] += oka[i] ? KS : 0;
      }
      dma(pa[i], buf * SB + (wave + NW * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      if (BF != FB_G3W || fresh) {
        bool ok;
        const long long off = b_off(invb[i], k0, &ok);
        pb[i] = g.B + off;
      } else {
        pb[i] += (long long)KS * g.ldb;  // (the same tap: KS rows further down the same column block)
      }
      dma(pb[i], buf * SB + PLA + (wave + NW * i) * 1024);
    }
  };

  // ---- fragments
  const int wm = (wave / WC) * WM, wn = (wave % WC) * WN;
  const int l15 = lane & 15, l4 = lane >> 4;
  struct Frags {
    f32x4 a[TM][NH];                         // [16-row block][kk / 16]: k = kk + 4 l4 + j, j = component
    f32x4 b[B_IS_KM ? 1 : TN][NH];           // K-contiguous B: the same
    float bk[B_IS_KM ? TN : 1][NH][4];       // contraction-major B: [block][kk / 16][j], one float per lane
  };
  const int kc_row_off = l15 * RB;
  const int kc_sw = kc_swz(l15);  // (16-row blocks start at multiples of 16: the block index does not enter the swizzle)
  auto load = [&](Frags& f, auto bufc) __attribute__((always_inline)) {
    constexpr int BUFOFF = decltype(bufc)::value * SB;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int h = 0; h < NH; ++h)
        f.a[a][h] = *reinterpret_cast<const f32x4*>(lds + BUFOFF + ((wm >> 4) + a) * 16 * RB + kc_row_off + (((4 * h + l4) ^ kc_sw) << 4));
    if constexpr (!B_IS_KM) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int h = 0; h < NH; ++h)
          f.b[b][h] = *reinterpret_cast<const f32x4*>(lds + BUFOFF + PLA + ((wn >> 4) + b) * 16 * RB + kc_row_off + (((4 * h + l4) ^ kc_sw) << 4));
    } else {
      constexpr int ROWB = BN * 4, NCH = ROWB / 64;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = 16 * h + 4 * l4 + j;
            const int pos = ((wn >> 4) + b) ^ ((4 * h + l4) & (NCH - 1));  // chunk position: chunk ^ ((k >> 2) & mask)
            f.bk[b][h][j] = *reinterpret_cast<const float*>(lds + BUFOFF + PLA + k * ROWB + (pos << 6) + l15 * 4);
          }
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto bval = [&](const Frags& f, int b, int h, int j) __attribute__((always_inline)) -> float {
    if constexpr (B_IS_KM) return f.bk[b][h][j];
    else return f.b[b][h][j];
  };
  auto mma = [&](const Frags& f) __attribute__((always_inline)) {
    // the order of k_gemm_tiled: kk outermost; small wave tiles take the component j outside the tile loops
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if constexpr (TM * TN <= 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = mfma16(bval(f, b, h, j), f.a[a][h][j], acc[a][b]);
      } else {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][b] = mfma16(bval(f, b, h, j), f.a[a][h][j], acc[a][b]);
      }
    }
  };

  // ---- the K loop: ping-pong of the two wave groups (mvae_p3.hip explains the scheme and its hazards)
  constexpr int per_tile = UA + UB;
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_dma = [&](bool all) __attribute__((always_inline)) {  // leave only the newest tile's requests in flight
    if (all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (per_tile == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (per_tile == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (per_tile == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (per_tile == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (per_tile == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const int nsteps = g.K / KS;
  auto tile_k = [&](int t) __attribute__((always_inline)) { return t < nsteps ? KS * t : 0; };  // (past the end: tile 0 again, never consumed)
  Frags f;
  issue(0, tile_k(0));
  issue(1, tile_k(1));
  wait_dma(true);
  barrier();
  if (grp == 1) barrier();  // group 1 starts one phase late
#ifdef MV_F32PP_DBG
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  auto step = [&](int t, auto buf, auto buf_fill) __attribute__((always_inline)) {
#ifdef MV_F32PP_DBG
    const bool stamp = t == 8 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && (wave & 3) == 0;
#endif
    MV_F32PP_STAMP(0);
#ifdef MV_F32PP_DBG
    if (!(g.dbg & 64)) load(f, buf);
    if (!(g.dbg & 2)) issue(decltype(buf_fill)::value, tile_k(t + 2));
#else
    load(f, buf);                                     // L: fragments of tile t, requests of tile t + 2
    issue(decltype(buf_fill)::value, tile_k(t + 2));
#endif
    MV_F32PP_STAMP(1);
    if (grp == 1) wait_dma(false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MV_F32PP_STAMP(2);
    barrier();
    MV_F32PP_STAMP(3);
#if MV_F32PP_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#ifdef MV_F32PP_DBG
    if (!(g.dbg & 8)) mma(f);
#else
    mma(f);                                           // C
#endif
#if MV_F32PP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    MV_F32PP_STAMP(4);
    if (grp == 0) wait_dma(false);
    MV_F32PP_STAMP(5);
    barrier();
    MV_F32PP_STAMP(6);
#ifdef MV_F32PP_DBG
    if (stamp && lane == 0)
      for (int i = 0; i < 7; ++i) g_f32pp_stamps[grp * 16 + i] = ts[i];
#endif
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  for (int t = 0; t < nsteps; t += 3) {
    step(t, B0{}, B2{});
    if (t + 1 < nsteps) step(t + 1, B1{}, B0{});
    if (t + 2 < nsteps) step(t + 2, B2{}, B1{});
  }
  if (grp == 0) barrier();  // (group 1's last phase)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the (unused) DMA of the last steps

  // ---- epilogue (k_gemm_tiled's): lane holds row l15, columns 4 * l4 + r of every 16 x 16 tile
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    int m = m0 + wm + a * 16 + l15;
    if (PARITY) {  // row of the parity class -> its pixel of the (2 IH) x (2 IW) output
      const int ox = m & ((1 << cg.lOW) - 1), oy = (m & ((1 << cg.lOHW) - 1)) >> cg.lOW, bb = m >> cg.lOHW;
      m = (bb * 2 * cg.IH + 2 * oy + par_y) * 2 * cg.IW + 2 * ox + par_x;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + wn + b * 16 + l4 * 4;
      f32x4 v = acc[a][b];
      const size_t o = (size_t)m * g.ldc + n;
      if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
      if (g.relu)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];  // torch.relu: NaN propagates
      if (g.mask) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(g.mask + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (mk[r] > 0.f) ? v[r] : 0.f;
      }
      *reinterpret_cast<f32x4*>(g.C + o) = v;
      if (g.Cp) store_planes4(g.Cp, g.psc, o, v[0], v[1], v[2], v[3]);
    }
  }
}

// Which kernel the exact-f32 contractions of whole-tile shapes take (process-wide, returns the previous value, < 0 queries):
// 1 (default) the ping-pong LDS-DMA kernel above, 0 the register-staged k_gemm_tiled.  Same bits either way.
#include <atomic>
static std::atomic<int> g_f32pp_on{1};
extern "C" int mvae_set_forward_kernel(int pingpong) {
  const int old = g_f32pp_on.load(std::memory_order_relaxed);
  if (pingpong >= 0) g_f32pp_on.store(pingpong ? 1 : 0, std::memory_order_relaxed);
  return old;
}

// ---- dispatch (called from mvae_conv.hip's launch_gemm_tiled): true if the shape was taken
// form: 0 NT plain (A [M, K], B [N, K]), 1 gathered conv (A = image, B [N, K]), 2 NN (A [M, K], B [K, N]),
//       3 transposed conv per parity class (A = image, B [C, 16 N]).  Whole tiles, K a multiple of 32.
static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}
template <int BM, int BN, int WR, int AF, int BF, int KS = 32>
static void launch_f32pp(const F32Args& a0, int zdim, hipStream_t s) {
  F32Args a = a0;
  const int nx = a.N / BN, ny = a.M / BM;
  // gx: fabric bytes ~ gx * |A| + (8 / gx) * |B| with |A| ~ M, |B| ~ N (same K, same element size); whole groups only
  static const bool xcd_off = getenv("MVAE_F32PP_NO_XCD") != nullptr;
  a.gx = 0;
  if (!xcd_off && (nx * ny) % 8 == 0) {
    double best = 0;
    for (int gx = 1; gx <= 8; gx *= 2) {
      if (nx % gx || ny % (8 / gx)) continue;
      const double cost = (double)gx * a.M * (AF == FA_KC ? 1.0 : 0.25) + (8.0 / gx) * a.N;  // (a gathered image is K / 4 wide)
      if (a.gx == 0 || cost < best) { a.gx = gx; best = cost; }
    }
  }
  dim3 grid(nx, ny, zdim);
#ifdef MV_F32PP_DBG
  a.dbg = getenv("MV_F32PP_DBG") ? atoi(getenv("MV_F32PP_DBG")) : 0;
#endif
  hipLaunchKernelGGL((k_gemm_f32pp<BM, BN, WR, AF, BF, KS>), grid, dim3(512), 0, s, a);
}
bool f32pp_try(int form, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, bf16r* Cp,
               long long psc, const float* bias, const float* mask, int relu, int M, int N, int K, ConvGeom cg, hipStream_t s) {
  if (!g_f32pp_on.load(std::memory_order_relaxed)) return false;
  if (M < 256 || (M & 127) || (N & 63) || N < 64 || (K & 31) || K < 64) return false;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)mask) & 15) || (ldc & 3) || (lda & 3) || (ldb & 3))
    return false;
  F32Args a{};
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.Cp = Cp; a.psc = psc;
  a.bias = bias; a.mask = mask; a.relu = relu; a.M = M; a.N = N; a.K = K; a.cg = cg; a.lCc = 0;
  if (form == 1 || form == 3) {
    a.lCc = ilog2_exact(cg.Cc);
    if (a.lCc < 5) return false;  // a K step of 32 channels inside one tap
  }
  // Tile choice (A/B override: MVAE_F32PP_TILE = 0 128 x 128 x 32 | 1 128 x 64 x 32 | 2 128 x 64 x 64).  Default 128 x 64 x 32: three
  // stages in 72 KB, so two workgroups share a CU and cover each other's prologue / epilogue.  Measured (tools/bench_f32pp.py,
  // TFLOP/s on [8192 x 4096 x 4096] / e2 / e1 shapes): 128 x 64 x 32 131 / 119 / 112, 128 x 128 x 32 130 / 67 / 64 (too few tiles
  // for the chip), 128 x 64 x 64 121 / 116 / 107, k_gemm_tiled 129 / 114 / 108 -- the sustained f32-MFMA rate of this chip is
  // ~130 TFLOP/s (83 % of the 157 at the 2.4 GHz boost clock), which every variant reaches on large shapes
  static const char* ov = getenv("MVAE_F32PP_TILE");
  const bool gather_k64 = (form != 1 && form != 3) || a.lCc >= 6;  // a K stage inside one tap
  int tile = 1;
  if (ov) tile = atoi(ov);
  if (tile == 0 && (N % 128)) tile = 1;
  if (tile == 2 && ((K % 64) || !gather_k64)) tile = 1;
#define MV_F32PP_CASE(FORM, AFV, BFV, Z)                                                         \
  case FORM:                                                                                     \
    if (tile == 0) launch_f32pp<128, 128, 2, AFV, BFV, 32>(a, Z, s);                             \
    else if (tile == 2) launch_f32pp<128, 64, 4, AFV, BFV, 64>(a, Z, s);                         \
    else launch_f32pp<128, 64, 4, AFV, BFV, 32>(a, Z, s);                                        \
    break;
  switch (form) {
    MV_F32PP_CASE(0, FA_KC, FB_KC, 1)
    MV_F32PP_CASE(1, FA_G1, FB_KC, 1)
    MV_F32PP_CASE(2, FA_KC, FB_KM, 1)
    MV_F32PP_CASE(3, FA_G3, FB_G3W, 4)
    default: return false;
  }
#undef MV_F32PP_CASE
  return true;
}
