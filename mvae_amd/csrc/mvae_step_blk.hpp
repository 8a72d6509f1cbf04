// mvae_step_blk.hpp -- the latent launches of the step for MANY SMALL components (BASELINE config [3], `6h2,6s2,6e2`:
// 18 components, 78 head columns, z_dim 48), as 16-row MFMA tiles.  Included by mvae_step.hip only.
//
// The one-row-per-workgroup latent kernels stream W_heads (125 KB) and W_d0 (77 KB) once per batch ROW: 128 workgroups
// read the same lines at the same moment and the launches took 10-11 us each.  Here a workgroup owns a 16-row block, the
// four small contractions (heads, first decoder layer, and their transposes) are v_mfma_f32_16x16x4_f32 tiles, and the
// launch structure becomes
//   2'  k_heads_comp   workgroup = (row block, GROUP of <= 4 consecutive components of one manifold kind whose head columns
//                      fit one 16-column tile): heads tile (K = H over 8 waves) -> wave 0: the components, one lane per
//                      (slot, row)
//   3'  k_fwd3m        workgroup = (row block, pair of logits column tiles): RECOMPUTES hd = relu(z W_d0^T + b) for its
//                      16 rows (K = Z <= 64, 25 MFMA column tiles over 8 waves) into LDS and contracts it with its two
//                      W_logits row blocks; BCE epilogue as in k_fwd23.  No launch boundary between hd and the logits.
//                      Extra workgroups at the front of its grid: the forward-mode dual records of launch 2's components.
//   5'  k_latent_bwd_blk   workgroup = (row block, 64-column group of dh): dz = dhd W_d0 (K = H), the records contracted
//                      with dz -> dheads, dh group = (dheads W_heads) [h > 0]; plus the dW_logits tile workgroups of
//                      launch 5.
// "NN" contractions (the operand's contraction index is its ROW index: W_d0 in dz, W_heads in dh) use an interleaved
// column labelling so that every operand request is a 16-byte row segment: lane (i, q) loads W[k][4i .. 4i+3] and feeds
// component tt to the MFMA of output tile tt, whose column j stands for n = 4j + tt -- four tiles cover 64 columns.
#pragma once

struct GroupTable {
  int ng;
  unsigned char first[kMaxComp];
  unsigned char count[kMaxComp];
};

// Greedy packing of consecutive same-kind components; false if a component alone needs more than one 16-column tile.
inline bool build_groups(const CompTable& t, GroupTable* g) {
  memset(g, 0, sizeof(*g));
  int i = 0;
  while (i < t.n) {
    int cols = 0, cnt = 0;
    const int kind = t.c[i].kind;
    while (i + cnt < t.n && cnt < 4 && t.c[i + cnt].kind == kind &&
           cols + t.c[i + cnt].true_dim + t.c[i + cnt].logvar_dim <= 16) {
      cols += t.c[i + cnt].true_dim + t.c[i + cnt].logvar_dim;
      ++cnt;
    }
    if (cnt == 0) return false;
    g->first[g->ng] = (unsigned char)i;
    g->count[g->ng] = (unsigned char)cnt;
    ++g->ng;
    i += cnt;
  }
  return true;
}

// descriptor of the k-th component of a group (k per lane, the four candidates fetched with uniform loads), with its
// columns re-based to the group's local tiles: heads [means | logvars], eps, z; radius_idx = k
__device__ __forceinline__ mvae_component_desc group_desc(const CompTable& t, int first, int cnt, int k, int Md) {
  const mvae_component_desc d0 = t.c[first], d1 = t.c[first + (cnt > 1 ? 1 : 0)], d2 = t.c[first + (cnt > 2 ? 2 : 0)],
                            d3 = t.c[first + (cnt > 3 ? 3 : 0)];
  mvae_component_desc r;
#define MV_SEL(f) r.f = (k == 0 ? d0.f : (k == 1 ? d1.f : (k == 2 ? d2.f : d3.f)))
  MV_SEL(kind); MV_SEL(true_dim); MV_SEL(mean_col); MV_SEL(logvar_col); MV_SEL(logvar_dim); MV_SEL(eps_col);
  MV_SEL(z_col);
#undef MV_SEL
  r.mean_col -= d0.mean_col;
  r.logvar_col = Md + (r.logvar_col - d0.logvar_col);
  r.eps_col -= d0.eps_col;
  r.z_col -= d0.z_col;
  r.radius_idx = k;
  return r;
}

// ---- 2': heads tile + components + dual records of one (row block, component group)
template <int DMAX>
__global__ __launch_bounds__(512) void k_heads_comp(CompTable t, GroupTable gt, const float* h, const float* Wh,
                                                    const float* bh, const float* eps, int eps_ld, const float* radii,
                                                    float* heads, int ldh, float* z, int ldz, float* z_user, float* kl,
                                                    float* kl_user, int B, int H, int NH, int Z, int Bv) {
  // Bv: valid rows (mvae_set_valid_rows); rows past it are padding and carry no KL term
  __shared__ float red[kW8][16][17];
  __shared__ __attribute__((aligned(16))) float heads_s[16][16];
  __shared__ __attribute__((aligned(16))) float eps_s[16][16];
  __shared__ float rad_s[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int MT = B >> 4;
  const int mt = (int)blockIdx.x % MT, grp = (int)blockIdx.x / MT;
  MV_SPAN_BEGIN(1);
  const int first = gt.first[grp], cnt = gt.count[grp];
  int Md = 0, Ml = 0;
  for (int k = 0; k < cnt; ++k) {  // uniform
    Md += t.c[first + k].true_dim;
    Ml += t.c[first + k].logvar_dim;
  }
  const int mean0 = t.c[first].mean_col, logvar0 = t.c[first].logvar_col, eps0 = t.c[first].eps_col,
            z0 = t.c[first].z_col;
  // head column behind tile column j (-1: unused)
  auto hc = [&](int j) { return j < Md ? mean0 + j : (j < Md + Ml ? logvar0 + (j - Md) : -1); };
  const int i = lane & 15, q = lane >> 4;
  const int nchunks = H >> 4;
  const int hci = hc(i);
  const float* hrow = h + (size_t)(mt * 16 + i) * H;
  const float* whrow = Wh + (size_t)(hci < 0 ? 0 : hci) * H;
  float4 ha[4], hb[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave + 8 * gq;
    const int k = ((c < nchunks ? c : 0) << 4) + (q << 2);
    ha[gq] = *reinterpret_cast<const float4*>(hrow + k);
    hb[gq] = *reinterpret_cast<const float4*>(whrow + k);  // unused tile columns re-read a valid row: never stored
  }
  const int hcn = hc(tid & 15);
  const float bhv = bh[hcn < 0 ? 0 : hcn];
  float epsv;
  {
    const int r = (tid >> 4) & 15, j = tid & 15;
    epsv = eps[(size_t)(mt * 16 + r) * eps_ld + eps0 + (j < Md ? j : 0)];
  }
  const float rad_r = radii[first + (tid < cnt ? tid : 0)];
  __builtin_amdgcn_sched_barrier(0);
  if (tid < 4) rad_s[tid] = rad_r;
  if (tid < 256) eps_s[tid >> 4][tid & 15] = (tid & 15) < Md ? epsv : 0.f;
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
    const float sv = reduce_tiles8(red, acc + acc2);
    if (tid < 256) {
      const int r = tid >> 4, n = tid & 15;
      const float v = hcn >= 0 ? sv + bhv : 0.f;
      heads_s[r][n] = v;
      if (hcn >= 0) heads[(size_t)(mt * 16 + r) * ldh + hcn] = v;
    }
  }
  lds_barrier();

  if (wave == 0) {  // the components: lane = slot * 16 + row
    const int sl = lane >> 4, r = lane & 15;
    const mvae_component_desc c = group_desc(t, first, cnt, sl, Md);
    if (sl < cnt) {
      const size_t row = (size_t)mt * 16 + r;
      float klv;
      comp_fwd_row<DMAX>(c, heads_s[r], eps_s[r], rad_s, z + row * ldz + z0, z_user ? z_user + row * Z + z0 : nullptr,
                         &klv, nullptr, nullptr, nullptr, nullptr);
      const float klm = (int)row < Bv ? klv : 0.f;
      kl[(size_t)(first + sl) * B + row] = klm;
      if (kl_user) kl_user[(size_t)(first + sl) * B + row] = klm;
    }
    MV_SPAN_END(1, 1);
    return;
  }
  // waves 1..7 are done: the forward-mode dual records of these components are computed by dedicated workgroups of the
  // NEXT launch (k_fwd3m), off every critical path -- inside this kernel their 3.5 us chains were its tail
}

// ---- 3': hd (recomputed per workgroup, K = Z <= 64) -> two logits tiles + BCE-with-logits
// The first n_dual = (B / 16) * (component groups) workgroups compute the forward-mode dual records of launch 2's
// components (heads / eps / radii read back from memory): ~5.5 us each on CUs the 200 tile workgroups leave free.
template <int DMAX>
__global__ __launch_bounds__(512) void k_fwd3m(CompTable t, GroupTable gt, const float* heads, int ldh,
                                               const float* eps, int eps_ld, const float* radii, int NH, float* duals,
                                               int n_dual, const float* z, int ldz, const float* Wd0, const float* bd0,
                                               const float* Wl, const float* bl, const float* x, float* hd, float* g,
                                               float* bce_part, float* logits_user, int B, int H, int D, int Z,
                                               float* hdF, int Bv) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // hd_s[16][H + 4]
  __shared__ float red[kW8][16][17];
  __shared__ float red2[kW8][16][17];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  MV_SPAN_BEGIN(2);
  const int ntD = D >> 4, ntP = (ntD + 1) >> 1, MT = B >> 4;
  if ((int)blockIdx.x < n_dual) {
    // ---- dual records of (row block mt, component group grp): item = (active direction of the group) * 16 + row
    float (*heads_s)[16] = reinterpret_cast<float (*)[16]>(&red[0][0][0]);   // [16][16]
    float (*eps_s)[16] = reinterpret_cast<float (*)[16]>(&red2[0][0][0]);    // [16][16]
    float* rad_s = &red2[1][0][0];
    const int mt = (int)blockIdx.x % MT, grp = (int)blockIdx.x / MT;
    const int first = gt.first[grp], cnt = gt.count[grp];
    int Md = 0, Ml = 0, gdirs = 0;
    for (int k = 0; k < cnt; ++k) {  // uniform
      Md += t.c[first + k].true_dim;
      Ml += t.c[first + k].logvar_dim;
      gdirs += t.dir_off[first + k + 1] - t.dir_off[first + k];
    }
    const int mean0 = t.c[first].mean_col, logvar0 = t.c[first].logvar_col, eps0 = t.c[first].eps_col;
    if (tid < 256) {
      const int r = tid >> 4, j = tid & 15;
      const int hcj = j < Md ? mean0 + j : (j < Md + Ml ? logvar0 + (j - Md) : -1);
      const float hv = heads[(size_t)(mt * 16 + r) * ldh + (hcj < 0 ? 0 : hcj)];
      const float ev = eps[(size_t)(mt * 16 + r) * eps_ld + eps0 + (j < Md ? j : 0)];
      heads_s[r][j] = hcj < 0 ? 0.f : hv;
      eps_s[r][j] = j < Md ? ev : 0.f;
    }
    if (tid < 4) rad_s[tid] = radii[first + (tid < cnt ? tid : 0)];
    __syncthreads();
    constexpr int AM = DMAX + 1, DS = DMAX + 2;
    for (int item = tid; item < gdirs * 16; item += 512) {
      const int r = item & 15, dd = item >> 4;
      int mine = 0, mydir = 0, base = 0, firstd = 0;
      for (int k = 0; k < cnt; ++k) {  // uniform loop, per-lane select
        const int nd = t.dir_off[first + k + 1] - t.dir_off[first + k];
        const int fd = t.first_dir[first + k];
        if (dd >= base && dd < base + nd) {
          mine = k;
          mydir = dd - base;
          firstd = fd;
        }
        base += nd;
      }
      const mvae_component_desc c = group_desc(t, first, cnt, mine, Md);
      float zd[AM];
      const float kld = comp_dual_dir<DMAX>(c, heads_s[r], eps_s[r], rad_s, mydir, zd);
      float* rec = duals + (((size_t)mt * 16 + r) * (NH + t.n) + firstd + mydir) * DS;
      const int A = ambient_dim(c.kind, c.true_dim);
      rec[0] = kld;
#pragma unroll
      for (int q2 = 0; q2 < AM; ++q2)
        if (q2 < A) rec[1 + q2] = zd[q2];
    }
    MV_SPAN_END(2, 2);
    return;
  }
  int pt = 0, mt = 0;
  {  // XCD-aware for the first 8 * floor(ntP / 8) pairs, plain order for the rest (no padding workgroups), as k_fwd23
    const int L = (int)blockIdx.x - n_dual;
    const int full = (ntP >> 3) << 3;
    if (L < full * MT) (void)xcd_tile(full, MT, &pt, &mt, L);
    else {
      const int idx = L - full * MT;
      pt = full + idx / MT;
      mt = idx - (pt - full) * MT;
    }
  }
  const int nt = pt * 2;
  const bool two = nt + 1 < ntD;
  const bool lead = pt == mt % ntP;
  const int ld = H + 4;
  float* hd_s = dyn;
  const int i = lane & 15, q = lane >> 4;
  const int nchunks = H >> 4;

  // requests in the order of use: z block and W_d0 (first phase), then the W_logits blocks and the epilogue operands
  const float* zrow = z + (size_t)(mt * 16 + i) * ldz;
  float4 za[4], wb[4][4];
  float bdv[4];
  // (16-wide K chunks past z_dim are not requested at all -- uniform guards: at ~36 cycles per wave-level request the 30
  // requests per wave of this phase are the kernel's first ~3.5 us, and for z_dim 48 a quarter of them fetched nothing used)
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const int k = kc * 16 + q * 4;
    za[kc] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kc * 16 < Z) za[kc] = *reinterpret_cast<const float4*>(zrow + (k < Z ? k : 0));
  }
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // (scalar guards below: chunks this wave does not have)
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave + 8 * gq;
    const int cc = c < nchunks ? c : 0;
    const float* wrow = Wd0 + (size_t)(cc * 16 + i) * Z;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int k = kc * 16 + q * 4;
      wb[gq][kc] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kc * 16 < Z && wave_s + 8 * gq < nchunks) wb[gq][kc] = *reinterpret_cast<const float4*>(wrow + (k < Z ? k : 0));
    }
    bdv[gq] = bd0[cc * 16 + i];
  }
  float4 w1[4], w2[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave + 8 * gq;
    const int k = ((c < nchunks ? c : 0) << 4) + (q << 2);
    w1[gq] = w2[gq] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave_s + 8 * gq < nchunks) {
      w1[gq] = *reinterpret_cast<const float4*>(Wl + (size_t)(nt * 16 + i) * H + k);
      w2[gq] = *reinterpret_cast<const float4*>(Wl + (size_t)((two ? nt + 1 : nt) * 16 + i) * H + k);
    }
  }
  const int nt_ep = (tid < 256 || !two) ? nt : nt + 1;
  const int m_ep = mt * 16 + ((tid & 255) >> 4), n_ep = nt_ep * 16 + (tid & 15);
  const float tv = x[(size_t)m_ep * D + n_ep];
  const float bias = bl[n_ep];
  __builtin_amdgcn_sched_barrier(0);

  // ---- hd tiles: this wave's column tiles c = wave, wave + 8, ...
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
    if (kc * 16 + q * 4 >= Z) za[kc] = make_float4(0.f, 0.f, 0.f, 0.f);  // clamped request: the columns do not exist
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave + 8 * gq;
    if (c < nchunks) {  // wave-uniform
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        if (kc * 16 < Z) {  // uniform
          a = mfma16(za[kc].x, wb[gq][kc].x, a);
          a2 = mfma16(za[kc].y, wb[gq][kc].y, a2);
          a = mfma16(za[kc].z, wb[gq][kc].z, a);
          a2 = mfma16(za[kc].w, wb[gq][kc].w, a2);
        }
      a += a2;
      const int col = (c << 4) + i;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        float v = a[r4] + bdv[gq];
        v = v < 0.f ? 0.f : v;  // torch.relu: NaN propagates
        hd_s[(q * 4 + r4) * ld + col] = v;
      }
    }
  }
  lds_barrier();
  if (lead) {
    if (hdF) {  // (uniform) fragment order only: every later reader (launches 4 and 5) takes fragments
      // (every pair workgroup storing one tile instead of the lead workgroup all 25: no faster, 36.65 vs 36.74 us per step)
      for (int e = tid; e < nchunks * 64; e += 512) {
        const int c = e >> 6, l = e & 63;
        const float* src = hd_s + (4 * (l >> 4)) * ld + 16 * c + (l & 15);
        f32x4 v = {src[0], src[ld], src[2 * ld], src[3 * ld]};
#pragma unroll
        for (int r4i = 0; r4i < 4; ++r4i) v[r4i] = mt * 16 + 4 * (l >> 4) + r4i < Bv ? v[r4i] : 0.f;  // (padding rows: zero)
        reinterpret_cast<f32x4*>(hdF)[((size_t)(c * MT + mt) << 6) + l] = v;
      }
    } else {
      for (int e4 = tid; e4 < 4 * H; e4 += 512) {
        const int r = e4 / (H >> 2), c4 = e4 - r * (H >> 2);
        *reinterpret_cast<float4*>(hd + ((size_t)mt * 16 + r) * H + 4 * c4) =
            *reinterpret_cast<const float4*>(hd_s + r * ld + 4 * c4);
      }
    }
  }
  // ---- the two logits tiles
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, bcc = {0.f, 0.f, 0.f, 0.f}, bcc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave + 8 * gq;
    if (c < nchunks) {
      const int k = (c << 4) + (q << 2);
      const float4 av = *reinterpret_cast<const float4*>(hd_s + i * ld + k);
      acc = mfma16(av.x, w1[gq].x, acc);
      bcc = mfma16(av.x, w2[gq].x, bcc);
      acc2 = mfma16(av.y, w1[gq].y, acc2);
      bcc2 = mfma16(av.y, w2[gq].y, bcc2);
      acc = mfma16(av.z, w1[gq].z, acc);
      bcc = mfma16(av.z, w2[gq].z, bcc);
      acc2 = mfma16(av.w, w1[gq].w, acc2);
      bcc2 = mfma16(av.w, w2[gq].w, bcc2);
    }
  }
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
  if (tid >= 256 && !two) return;
  const float y = sv + bias;
  // F.binary_cross_entropy_with_logits (image_reconstruction.py:81-82): (1-t)*y - log_sigmoid(y)
  const float e = expf(-fabsf(y));
  const float log_sig = fminf(y, 0.f) - mvf::log1p_pos(e);
  float loss = (1.f - tv) * y - log_sig;
  const float sig = (y >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
  loss += __shfl_xor(loss, 8, 16);
  loss += __shfl_xor(loss, 4, 16);
  loss += __shfl_xor(loss, 2, 16);
  loss += __shfl_xor(loss, 1, 16);
  const bool vrow = m_ep < Bv;  // (a padding row: no reconstruction term, and through g = 0 no gradient behind it)
  g[(size_t)m_ep * D + n_ep] = vrow ? sig - tv : 0.f;
  if (logits_user) logits_user[(size_t)m_ep * D + n_ep] = y;
  if ((tid & 15) == 0) bce_part[(size_t)nt_ep * B + m_ep] = vrow ? loss : 0.f;
  MV_SPAN_END(2, 1);
}

// ---- 5': backward through the first decoder layer, the components (dual records) and the heads, per 16-row block;
// dW_logits tiles as in k_latent_bwd
constexpr int kBlkDirs = kHeadsMax + kMaxComp;  // active input directions of a row (<= heads_dim + ncomp)

// One entry per RECORD of a row of `duals` (component-major: d mean directions, logvar_dim logvar directions, one
// radius / curvature direction -- written by the forward launch only while that radius is trainable), put into the
// workspace by the host (mvae_create / mvae_set_radius_trainable):
//   x = 1 if the direction is active, y = head column that receives the gradient or -(1 + component) for a radius,
//   z = first z column of the component, w = its ambient dimension
inline int fill_dirtab(const CompTable& t, int4* tab) {
  int n = 0;
  for (int ci = 0; ci < t.n; ++ci) {
    const mvae_component_desc& c = t.c[ci];
    const int nh = c.true_dim + c.logvar_dim;
    for (int dir = 0; dir <= nh; ++dir, ++n) {
      tab[n].x = (dir < nh || (t.trainable[ci] & 1)) ? 1 : 0;
      tab[n].y = dir < c.true_dim ? c.mean_col + dir : (dir < nh ? c.logvar_col + (dir - c.true_dim) : -(1 + ci));
      tab[n].z = c.z_col;
      tab[n].w = ambient_dim(c.kind, c.true_dim);
    }
  }
  return n;
}

// TT = interleaved column tiles of dz: lane (i, q) loads W_d0[k][TT i .. TT i + TT - 1] (one 4 TT-byte request) and
// tile tt's column j stands for z column TT j + tt; TT = 3 covers z_dim <= 48 (config [3]: exactly), TT = 4 up to 64.
// hdF != NULL (with dzp, dhF, dheadsF: the fragment-order form): the launch has 512-thread workgroups -- the dW_logits tiles
// eight per workgroup from g (row-major, fragment row sequence) and hd's fragment-order copy, so that tiles + row workgroups
// are fewer than the chip has CUs (a row workgroup that shares its CU with a tile workgroup ends ~0.8 us late); the row
// workgroups keep their four waves, the other four leave at once.
template <int DMAX, bool ADAM, int TT>
__global__ __launch_bounds__(512) void k_latent_bwd_blk(CompTable t, const int4* dirtab, const float* dhd,
                                                        const float* Wd0, int ldh, const float* h, const float* Wh,
                                                        float* dheads, float* dh, float* drpart, const float* g,
                                                        const float* hd, float* dWl, float beta, int B, int H, int D,
                                                        int NH, int Z, int n_blk, AdamArgs awl, const float* duals,
                                                        const float* dzp, float* dhF, const float* hF, float* dheadsF, const float* hdF,
                                                        StatsArgs sa) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // TT == 1: W_d0 [H][Z]
  __shared__ float red[4][4][16][17];  // [wave][interleaved tile][row][col]
  __shared__ __attribute__((aligned(16))) float dz_s[16][68];
  __shared__ __attribute__((aligned(16))) float dheads_s[16][kHeadsMax + 4];
  int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  MV_SPAN_BEGIN(4);
  if (sa.bce_part) {  // (uniform) the step's statistics job, dispatched first: launch 4's tiles end before it would there
    if (b == 0) {
      job_step_stats(&red[0][0][0][0], sa.bce_part, sa.kl, sa.bce_user, sa.stats, beta, B, D >> 4, sa.ncomp);
      MV_SPAN_END(4, 3);
      return;
    }
    b -= 1;
  }
  if (b >= n_blk) {
    b -= n_blk;
    if (hdF) {  // (uniform) tile index -> (H tile fastest, D tile): the 8 waves of a workgroup share g's column block mostly
      const int ntH = H >> 4, tw = b * 8 + wave, pt = tw / ntH, qt = tw - pt * ntH;
      if (pt * 16 < D) job_tn_halffrag<ADAM>(g, D, pt, D, hdF, qt, H, B >> 4, dWl, H, awl);
      MV_SPAN_END(4, 2);
      return;
    }
    const int ntHg = ((H >> 4) + kTileWaves5 - 1) / kTileWaves5;
    job_tn_wave<ADAM, true>(g, D, D, b / ntHg, hd, H, H, (b % ntHg) * kTileWaves5 + wave, B, dWl, H, awl);
    MV_SPAN_END(4, 2);
    return;
  }
  if (tid >= 256) return;  // (fragment-order form: 512-thread workgroups; the row part is written for four waves)
  const int MT = B >> 4;
  const int mt = b % MT, s = b / MT;  // s: the 64-column group of dh this workgroup produces
  MV_TDECL;
  MV_T(0);
  const int i = lane & 15, q = lane >> 4;
  constexpr int DS = DMAX + 2;
  constexpr int kPre = TT == 1 ? 1 : 2;  // 64-record rounds whose table entries and records are requested at the top
  typedef float fTT __attribute__((ext_vector_type(TT), aligned(4)));

  // ---- requests, in the order of use.  Dual records: wave w takes rows 4w .. 4w+3, lane = record of the row (64 per
  // round): consecutive lanes read consecutive records (coalesced), and neither the table entry nor the record address
  // depends on a loaded value, so everything below is in flight at once.
  const int nrec = NH + t.n;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  int4 te[kPre];
  f32x2 du[kPre][4][DS / 2];
#pragma unroll
  for (int rd = 0; rd < kPre; ++rd) {
    const int ri = rd * 64 + lane;
    const int rc = ri < nrec ? ri : 0;
    te[rd] = dirtab[rc];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const f32x2* rec = reinterpret_cast<const f32x2*>(duals + (((size_t)mt * 16 + wave * 4 + rr) * nrec + rc) * DS);
#pragma unroll
      for (int e = 0; e < DS / 2; ++e) du[rd][rr][e] = rec[e];
    }
  }
  // the ReLU mask of this workgroup's dh block: row-major h, or (dhF: dh leaves in fragment order, wave w = the 16-column
  // tile 4 s + w) the fragment-order copy launch 1 wrote
  const float4 hm = *reinterpret_cast<const float4*>(
      dhF ? hF + ((((size_t)((4 * s + wave) * 16 < H ? 4 * s + wave : 0) * (B >> 4) + mt) << 6) + lane) * 4
          : h + ((size_t)mt * 16 + (tid >> 4)) * H + ((s * 64 + 4 * (tid & 15) < H) ? s * 64 + 4 * (tid & 15) : 0));
  const int nchunks = H >> 4;
  const float* arow = dhd + (size_t)(mt * 16 + i) * H;
  const int KC = (NH + 15) >> 4;  // 16-wide k chunks of the dh contraction
  f32x4 acc[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (TT == 1) {
    // z_dim <= 16 (the BASELINE configs with few components): W_d0 is a few KB -- staged once through LDS with coalesced
    // 16-byte requests (H Z / 1024 per thread) instead of 4 strided 4-byte requests per chunk; the dhd block of this wave
    // (<= 8 chunks) is requested up front.
    float* wd_s = dyn;
    const int nw4 = (H * Z) >> 2;
    f32x4 wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // H Z <= 512 * 16: at most 8 per thread
      const int e4 = tid + 256 * u;
      wv[u] = *reinterpret_cast<const f32x4*>(Wd0 + 4 * (size_t)(e4 < nw4 ? e4 : 0));
    }
    float4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = wave + 4 * u;
      a[u] = *reinterpret_cast<const float4*>(arow + (c < nchunks ? c : 0) * 16 + 4 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int e = tid; e < 16 * (KC * 16 - NH); e += 256) {  // zero padding of the last chunk
      const int w = KC * 16 - NH;
      dheads_s[e / w][NH + e % w] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e4 = tid + 256 * u;
      if (e4 < nw4) *reinterpret_cast<f32x4*>(wd_s + 4 * e4) = wv[u];
    }
    lds_barrier();
    MV_T(1);
    const int zi = i < Z ? i : 0;  // tile columns past Z repeat column 0; nobody reads their outputs
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = wave + 4 * u;
      if (c < nchunks) {  // uniform
        const float* wk = wd_s + (size_t)(c * 16 + 4 * q) * Z + zi;
        acc[0] = mfma16(a[u].x, wk[0], acc[0]);
        acc[1] = mfma16(a[u].y, wk[Z], acc[1]);
        acc[0] = mfma16(a[u].z, wk[2 * Z], acc[0]);
        acc[1] = mfma16(a[u].w, wk[3 * Z], acc[1]);
      }
    }
    acc[0] += acc[1];
  } else if (dzp) {
    // (uniform) dz arrives as H / 16 partial products per row block from launch 4's tiles (k_dec1_bwd, LITE 2), each a
    // [16 rows][16 z columns] tile in the MFMA's output order: wave w adds the tiles nt = w, w + 4, ... of every z tile
    // (fixed order), the four waves meet in `red` like the contraction they replace -- no K = H product over dhd and W_d0,
    // whose operand requests alone took ~3 us to issue.
    const int ZT = (Z + 15) >> 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(dzp) + ((size_t)(mt * nchunks) * ZT << 6) + lane;
    for (int e = tid; e < 16 * (KC * 16 - NH); e += 256) {  // zero padding of the last chunk
      const int w = KC * 16 - NH;
      dheads_s[e / w][NH + e % w] = 0.f;
    }
    f32x4 pv[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int nt = wave + 4 * u;
#pragma unroll
      for (int zt = 0; zt < 4; ++zt)
        if (zt < TT) pv[u][zt] = src[(size_t)((nt < nchunks ? nt : 0) * ZT + (zt < ZT ? zt : 0)) << 6];
    }
    __builtin_amdgcn_sched_barrier(0);
    MV_T(1);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (wave + 4 * u < nchunks) {  // uniform
#pragma unroll
        for (int zt = 0; zt < 4; ++zt)
          if (zt < TT) acc[zt] += pv[u][zt];
      }
  } else {
  // a lane whose first column exists reads its TT columns even if the last ones lie past the row (they belong to the
  // next row / the padding after the matrix: finite or not, they only reach output columns >= Z, which nobody reads);
  // lanes entirely past Z re-read column 0
  const int ncol = (TT * i < Z) ? TT * i : 0;
  // Software pipeline over this wave's (at most 8) chunks: the CU's load path takes ~36 cycles per wave-level request and
  // a chunk is 5 requests x 4 waves, so the requests of the whole operand take ~3 us to ISSUE -- the MFMAs of chunk u run
  // while the requests of chunks u+3.. are still being issued (vmcnt lets a chunk be consumed as soon as it has landed).
  float4 a[8];
  fTT bq[8][4];
  auto request = [&](int u) {
    const int c = wave + 4 * u;
    const int cc = c < nchunks ? c : 0;
    a[u] = *reinterpret_cast<const float4*>(arow + cc * 16 + 4 * q);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
      bq[u][tp] = *reinterpret_cast<const fTT*>(Wd0 + (size_t)(cc * 16 + 4 * q + tp) * Z + ncol);
  };
  constexpr int kAhead = 3;
#pragma unroll
  for (int u = 0; u < kAhead; ++u) request(u);
  for (int e = tid; e < 16 * (KC * 16 - NH); e += 256) {  // zero padding of the last chunk
    const int w = KC * 16 - NH;
    dheads_s[e / w][NH + e % w] = 0.f;
  }
  MV_T(1);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u + kAhead < 8) request(u + kAhead);
    __builtin_amdgcn_sched_barrier(0);
    if (wave + 4 * u < nchunks) {  // uniform
      const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt] = mfma16(av[tp], bq[u][tp][tt], acc[tt]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  MV_T(2);
  // the W_heads requests of the last phase travel during the reduction and the record phase
  const int n0 = s * 64;
  const int col = (n0 + 4 * i < H) ? n0 + 4 * i : 0;
  f32x4 bw[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = wave + 4 * u;
    if (c < KC) {  // uniform: chunks this wave does not have are not requested
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        const int krow = c * 16 + 4 * q + tp;
        bw[u][tp] = *reinterpret_cast<const f32x4*>(Wh + (size_t)(krow < NH ? krow : 0) * H + col);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) red[wave][tt][q * 4 + r4][i] = acc[tt][r4];
  lds_barrier();
  {
    const int r = tid >> 4, cj = tid & 15;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)  // z column TT cj + tt (the contraction's interleaved tiles) | 16 tt + cj (partial tiles)
      dz_s[r][(TT > 1 && dzp) ? 16 * tt + cj : TT * cj + tt] =
          (red[0][tt][r][cj] + red[1][tt][r][cj]) + (red[2][tt][r][cj] + red[3][tt][r][cj]);
  }
  lds_barrier();
  MV_T(3);

  // ---- d(loss)/d(direction) = beta * d kl + <dz, d z>
  for (int rd0 = 0; rd0 * 64 < nrec; rd0 += kPre) {
    if (rd0 > 0) {  // more than 64 kPre records per row: further rounds on demand
#pragma unroll
      for (int rd = 0; rd < kPre; ++rd) {
        const int ri = (rd0 + rd) * 64 + lane;
        const int rc = ri < nrec ? ri : 0;
        te[rd] = dirtab[rc];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const f32x2* rec = reinterpret_cast<const f32x2*>(duals + (((size_t)mt * 16 + wave * 4 + rr) * nrec + rc) * DS);
#pragma unroll
          for (int e = 0; e < DS / 2; ++e) du[rd][rr][e] = rec[e];
        }
      }
    }
#pragma unroll
    for (int rd = 0; rd < kPre; ++rd) {
      const int ri = (rd0 + rd) * 64 + lane;
      if (ri < nrec && te[rd].x) {
        const int oc = te[rd].y, zc = te[rd].z, A = te[rd].w;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = wave * 4 + rr;
          float gv = beta * du[rd][rr][0][0];
#pragma unroll
          for (int e = 0; e < DMAX + 1; ++e) {  // branch-free: entries past the ambient dimension are selected away
            const float pv = dz_s[r][zc + e < 68 ? zc + e : 67] * du[rd][rr][(1 + e) >> 1][(1 + e) & 1];
            gv += e < A ? pv : 0.f;
          }
          gv = mt * 16 + r < sa.valid_rows ? gv : 0.f;  // (padding rows: no term)
          if (oc >= 0) dheads_s[r][oc] = gv;
          else if (s == 0) drpart[(size_t)(-oc - 1) * B + mt * 16 + r] = gv;
        }
      }
    }
  }
  lds_barrier();
  MV_T(4);
  if (s == 0) {
    if (dheadsF) {  // (uniform) fragment order, whole tiles (dheads_s is zero past NH): launch 6'' contracts fragments
      for (int e = tid; e < KC * 64; e += 256) {
        const int pt = e >> 6, l = e & 63, li = l & 15, lq = l >> 4;
        reinterpret_cast<f32x4*>(dheadsF)[((size_t)(pt * (B >> 4) + mt) << 6) + l] =
            f32x4{dheads_s[4 * lq][16 * pt + li], dheads_s[4 * lq + 1][16 * pt + li], dheads_s[4 * lq + 2][16 * pt + li],
                  dheads_s[4 * lq + 3][16 * pt + li]};
      }
    } else {
      for (int e = tid; e < 16 * NH; e += 256) {
        const int r = e / NH, n = e - r * NH;
        dheads[((size_t)mt * 16 + r) * ldh + n] = dheads_s[r][n];
      }
    }
  }

  // ---- dh columns [64 s, 64 s + 64) = (dheads W_heads) [h > 0]: K = NH in 16-wide chunks over the 4 waves
  {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = wave + 4 * u;
      if (c < KC) {  // uniform
        const float4 a4 = *reinterpret_cast<const float4*>(&dheads_s[i][c * 16 + 4 * q]);  // zero past NH
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) acc[tt] = mfma16(av[tp], bw[u][tp][tt], acc[tt]);
      }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) red[wave][tt][q * 4 + r4][i] = acc[tt][r4];
    lds_barrier();
    if (dhF) {  // (uniform) fragment order for launch 6 (k_enc_bwd3): lane (i, q) of wave w holds dh[16 mt + 4 q + t][n0 + 16 w + i]
      const int c = 16 * wave + i, cq = c >> 2, tt = c & 3;
      if (n0 + 16 * wave < H) {
        f32x4 v;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int r = 4 * q + t4;
          v[t4] = (red[0][tt][r][cq] + red[1][tt][r][cq]) + (red[2][tt][r][cq] + red[3][tt][r][cq]);
        }
        v[0] = hm.x > 0.f ? v[0] : 0.f;
        v[1] = hm.y > 0.f ? v[1] : 0.f;
        v[2] = hm.z > 0.f ? v[2] : 0.f;
        v[3] = hm.w > 0.f ? v[3] : 0.f;
        reinterpret_cast<f32x4*>(dhF)[((size_t)((4 * s + wave) * (B >> 4) + mt) << 6) + lane] = v;
      }
      MV_T(5);
      MV_TFLUSH(24, 6, 9);
      MV_SPAN_END(4, 1);
      return;
    }
    const int r = tid >> 4, cq = tid & 15;
    const int n = n0 + 4 * cq;
    if (n < H) {
      const size_t o = ((size_t)mt * 16 + r) * H + n;
      float4 v;
      v.x = (red[0][0][r][cq] + red[1][0][r][cq]) + (red[2][0][r][cq] + red[3][0][r][cq]);
      v.y = (red[0][1][r][cq] + red[1][1][r][cq]) + (red[2][1][r][cq] + red[3][1][r][cq]);
      v.z = (red[0][2][r][cq] + red[1][2][r][cq]) + (red[2][2][r][cq] + red[3][2][r][cq]);
      v.w = (red[0][3][r][cq] + red[1][3][r][cq]) + (red[2][3][r][cq] + red[3][3][r][cq]);
      v.x = hm.x > 0.f ? v.x : 0.f;
      v.y = hm.y > 0.f ? v.y : 0.f;
      v.z = hm.z > 0.f ? v.z : 0.f;
      v.w = hm.w > 0.f ? v.w : 0.f;
      *reinterpret_cast<float4*>(dh + o) = v;
    }
  }
  MV_T(5);
  MV_TFLUSH(24, 6, 9);
  MV_SPAN_END(4, 1);
}
