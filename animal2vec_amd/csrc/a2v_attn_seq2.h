// Student attention, round 6: whole-sequence forward and backward for bf16, HD = 64, S <= 160 (the kept frames of a clone:
// 143 .. 155 of 2000 as the sampler draws them) -- included by a2v_attn.hip (AttnArgs / AttnBwdArgs, tr_frag, dual_off, rows_max).
// Reference arithmetic: nn/modalities/modules.py:368-410 (AltAttention.forward), base.py:292-314 (ALiBi on the kept positions).
//
// What the round-5 kernels lost (profiles/r05_e_kernel_stats_large.csv: backward 895 us at MFMA-busy 0.12 against a 0.32 ms HBM
// floor, forward 346 us against 0.16): they ran two workgroups of four 220..248-register waves per CU, and the backward put six
// barriers and an LDS round trip of dS between its products, so every wave sat in exposed LDS / barrier latency with one partner
// per SIMD to cover it.  These kernels trade redundant matrix work for independence:
//
//   * a workgroup is FOUR waves, one per SIMD (one (n, h) sequence = ten 16-row groups: two or three "items" per wave), <= 128 registers and <= 50 KB of
//     LDS: three workgroups per CU, four waves per SIMD;
//   * backward = two ROLES, each barrier free and register resident:
//       role 1 (key group on the lanes, all query rows): S = Q K^T and dP = dO V^T with the key on the lane, so P and dS are
//         already the B operands of dV^T += dO^T P and dK^T += Q^T dS (no LDS round trip); needs the Q and dO images;
//       role 2 (query group on the lanes, all keys): S^T = K Q^T and dP^T = V dO^T recomputed with the query on the lane, so
//         dS^T is the B operand of dQ^T += K^T dS^T; needs the K and V images.
//     The score arithmetic runs twice (14 instead of 10 MFMA products per 16 x 16 score tile) and dS never crosses LDS; the
//     two roles use the SAME 40 KB of LDS one after the other (Q / dO images, then K / V images, fetched by LDS-DMA: no staging
//     registers), three barriers per workgroup in all;
//   * the dropout decision is made ONCE, by the forward, and kept as a bit mask (A2V "dropmask": 8 bytes per (row, lane
//     quarter), 5 KB per (n, h)); the backward reads bits instead of re-hashing every score in both roles (the hash was 19 of
//     ~35 vector issue slots per score with the key on the lane).  Without a mask pointer the backward hashes as before.
//
// Mask layout: dropmask[((n H + h) 160 + q) 4 + g] = {w0, w1} for the forward lane (q, g) that owns keys 16 j + 4 g + r:
//   j < 8: bit 31 - (4 j + r) of w0;  j = 8, 9: bit 7 - (4 (j - 8) + r) of w1;  1 = kept.
#pragma once

#define AS2_ROWS 160
#define AS2_NW 4
#define AS2_NT (64 * AS2_NW)
#define AS2_IMG (AS2_ROWS * 128)
#define AS2F_LDS (2 * AS2_IMG + 2 * AS2_ROWS * 4)
#define AS2B_LDS (2 * AS2_IMG + 5 * AS2_ROWS * 4 + 2 * 4 * AS2_ROWS * 4 + 192 * 4 + 64)

// one 1-KiB piece (8 rows x 128 B) of a dual-use image by LDS-DMA: lane l lands at row 8 piece + (l >> 3), slot l & 7, and fetches
// the chunk that belongs there (dual_off: slot = chunk ^ (row & 7)); rows past S re-read row S - 1 (finite data, masked by lse / kv_ok)
__device__ __forceinline__ void as2_dma_piece(const bf16* __restrict__ base, long long rowstride, int S, char* img, int piece, int lane) {
  const int r = piece * 8 + (lane >> 3);
  const int rc = r < S ? r : S - 1;
  const int c = (lane & 7) ^ (r & 7);
  __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)(base + (long long)rc * rowstride + c * 8), (af_lds_void_t*)(img + piece * 1024), 16, 0, 0);
}

// The workgroups of a launch are identical, start together and finish together: left alone, every CU's three resident workgroups
// fetch at the same time, then compute at the same time, then store at the same time -- load, math and store times ADD (measured:
// 110 + 170 + 69 us for the forward).  The first workgroup each CU slot receives (dispatch order: 256 CUs get one workgroup each, then
// a second, then a third) therefore starts `sleeps` x slot x ~4 us late; the offset between a CU's workgroups then persists, since
// each retiring workgroup is replaced at once by one of the same length.
__device__ __forceinline__ void as2_stagger(int sleeps) {
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (sleeps > 0 && lin >= 256 && lin < 768) {
    const int nsl = (lin >> 8) * sleeps;
    for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(127);
  }
}

__device__ __forceinline__ unsigned as2_pack2(float lo, float hi) {     // two fp32 -> packed bf16 pair (RNE), one instruction
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// ---------------------------------------------------------------------------------------------
// forward: K / V images by LDS-DMA, one barrier; a wave's item = 16 query rows against every key, whole score row in registers
template <bool DROP, bool MD, bool WM>
__global__ __launch_bounds__(AS2_NT, 3) void attn_seq2_fwd_kernel(AttnArgs a, unsigned* __restrict__ dropmask) {
  typedef bf16 T;
  constexpr int HD = 64, ND = 4, NKG = 10;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  char* const sK = smem;
  char* const sV = smem + AS2_IMG;
  float* const sPos = (float*)(smem + 2 * AS2_IMG);
  float* const sKb = sPos + AS2_ROWS;

  as2_stagger(a.xcd_order >> 8);
  const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
  unsigned long long t_1 = 0, t_2 = 0, t_3 = 0, t_4 = 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int fr = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const T* qkv = (const T*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = a.scale * LOG2E;
  const float c_rel = -a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f) / a.scale;       // ALiBi slope in accumulator units
#pragma unroll
  for (int u = 0; u < 40 / AS2_NW; ++u) {
    const int piece = widu + AS2_NW * u;                  // 0 .. 39: K image pieces 0..19, V image pieces 20..39
    if (piece < 20) as2_dma_piece(qkv + H * HD, rowstride, S, sK, piece, lane);
    else as2_dma_piece(qkv + 2 * H * HD, rowstride, S, sV, piece - 20, lane);
  }
  // per-key terms: position (keys past S: any finite value) and the score bias 0 / -1e30 that removes the keys past S -- it is the
  // S^T accumulator's INITIAL value (the MFMA's C operand), so no per-score select exists anywhere
  if (tid < AS2_ROWS) {
    sPos[tid] = tid < S ? (a.pos ? (float)a.pos[(long long)n * S + tid] : (float)tid) : 0.f;
    sKb[tid] = tid < S ? 0.f : -1.0e30f;
  }

  bf16x8 qf[2], qf_n[2];
  auto fetch_q = [&](int grp, bf16x8* dst) {
    const int q_ = grp * 16 + fr, qc_ = q_ < S ? q_ : S - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dst[ks] = *(const bf16x8*)(qkv + (long long)qc_ * rowstride + ks * 32 + g * 8);
  };
  fetch_q(wid, qf_n);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (a.xcd_order & 16) t_1 = __builtin_amdgcn_s_memtime();

  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const unsigned t16m1 = (a.drop.thresh >> 16) - 1u;
  const unsigned scale_bits = __float_as_uint(a.drop.scale);
#pragma unroll 1
  for (int it = 0; it < (AS2_ROWS / 16 + AS2_NW - 1) / AS2_NW; ++it) {
    const int grp = wid + AS2_NW * it;
    if (grp * 16 >= S) break;                              // (uniform; no barrier follows)
    if (a.xcd_order & 4) break;                            // (timing experiments, a2v_debug_set(40, flags): no item math)
    qf[0] = qf_n[0]; qf[1] = qf_n[1];
    if ((grp + AS2_NW) * 16 < S) fetch_q(grp + AS2_NW, qf_n);      // the next item's rows arrive under this item's math
    const int q = grp * 16 + fr;
    const int qc = q < S ? q : S - 1;
    const float pos_q = sPos[qc];

    // ---- S^T in accumulator units: s[j][r] for key 16 j + 4 g + r.  All ten key groups, always: rows past S hold copies of
    // row S - 1 (finite) and carry -1e30 from the accumulator's initial value
    f32x4 s[NKG];
#pragma unroll
    for (int j = 0; j < NKG; ++j) {
      bf16x8 kf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[ks] = *(const bf16x8*)(sK + dual_off(16 * j + fr, ks * 4 + g));
      const f32x4 pk = *(const f32x4*)(sPos + 16 * j + 4 * g);
      f32x4 acc = *(const f32x4*)(sKb + 16 * j + 4 * g);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[j][r] = __builtin_fmaf(__builtin_fabsf(pos_q - pk[r]), c_rel, acc[r]);
    }
    if ((a.xcd_order & 16) && it == 0) { asm volatile("" :: "v"(s[9][3])); t_2 = __builtin_amdgcn_s_memtime(); }
    // ---- softmax over the whole row, dropout (+ mask bits), P in place
    float mx = max3f(s[0][0], s[0][1], s[0][2]);
    mx = max2f(mx, s[0][3]);
#pragma unroll
    for (int j = 1; j < NKG; ++j) {
      mx = max3f(mx, s[j][0], s[j][1]);
      mx = max3f(mx, s[j][2], s[j][3]);
    }
    mx = rows_max(mx);
    const float nm = -mx * scale2;
    const unsigned rt = DROP ? drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qc)) : 0u;
    float ps = 0.f, pa = 0.f;
    unsigned w0 = 0u, w1 = 0u;
#pragma unroll
    for (int j = 0; j < NKG; ++j) {
      unsigned hp[2] = {0u, 0u};
      if (DROP) {                                          // keys 16 j + 4 g + {0,1} and + {2,3}: one hash per pair
        hp[0] = drop_hash_pair(rt, (unsigned)(16 * j + 4 * g));
        hp[1] = drop_hash_pair(rt, (unsigned)(16 * j + 4 * g + 2));
      }
      f32x4 kp = {0.f, 0.f, 0.f, 0.f};
      if constexpr (MD) kp = *(const f32x4*)(sPos + 16 * j + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], scale2, nm));
        ps += p;                                           // the normaliser is dropout-free
        if constexpr (MD) pa = __builtin_fmaf(__builtin_fabsf(pos_q - kp[r]), p, pa);
        if (DROP) {
          const unsigned h16 = (r & 1) ? (hp[r >> 1] >> 16) : (hp[r >> 1] & 0xFFFFu);
          const int d = (int)(t16m1 - h16);                // < 0  <=>  h16 >= thresh >> 16  <=>  kept
          p *= __uint_as_float((unsigned)(d >> 31) & scale_bits);
          if constexpr (WM) {
            if (j < 8) w0 = __builtin_amdgcn_alignbit(w0, (unsigned)d, 31);   // (w << 1) | sign(d)
            else w1 = __builtin_amdgcn_alignbit(w1, (unsigned)d, 31);
          }
        }
        s[j][r] = p;
      }
    }
    const float l_row = rows_sum(ps);
    const float a_row = MD ? rows_sum(pa) : 0.f;
    if ((a.xcd_order & 16) && it == 0) { asm volatile("" :: "v"(l_row)); t_3 = __builtin_amdgcn_s_memtime(); }
    // ---- O^T = V^T P^T over 32-key chunks
    f32x4 o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      {
        union { unsigned u[4]; bf16x8 v; } pf;
        pf.u[0] = as2_pack2(s[2 * st][0], s[2 * st][1]);
        pf.u[1] = as2_pack2(s[2 * st][2], s[2 * st][3]);
        pf.u[2] = as2_pack2(s[2 * st + 1][0], s[2 * st + 1][1]);
        pf.u[3] = as2_pack2(s[2 * st + 1][2], s[2 * st + 1][3]);
        const int r_lo = 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const int chl = 2 * i + (pp >> 1), sub = (pp & 1) << 3;
          const bf16x8 vf = tr_frag(sV + dual_off(r_lo, chl) + sub, sV + dual_off(r_hi, chl) + sub);
          o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf.v, o[i], 0, 0, 0);
        }
      }
    }
    if ((a.xcd_order & 16) && it == 0) { asm volatile("" :: "v"(o[3][3])); t_4 = __builtin_amdgcn_s_memtime(); }
    // ---- finish: O = O^T / l ; lse = ln2 * (m*scale2 + log2 l)
    const float inv = 1.f / l_row;
    if (q < S && !(a.xcd_order & 1)) {
      T* orow = (T*)a.out + ((long long)n * S + q) * ((long long)H * HD) + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        f32x4 v = o[i] * inv;
        Vec4<T>::store(orow + 16 * i + 4 * g, v);
      }
      if (g == 0 && a.lse) a.lse[((long long)n * H + h) * S + q] = (mx * scale2 + __builtin_amdgcn_logf(l_row)) * 0.6931471805599453f;
      if (MD && g == 0) a.mdist[((long long)n * H + h) * S + q] = a_row * inv;
      if (WM && !(a.xcd_order & 2)) {
        uint2 w; w.x = w0; w.y = w1;
        *(uint2*)(dropmask + ((((long long)n * H + h) * AS2_ROWS + q) * 4 + g) * 2) = w;
      }
    }
  }
  if ((a.xcd_order & 16) && a.mdist && lane == 0) {       // diagnostic stamps (cycles): prologue, S^T, softmax, PV of item 0, whole wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_5 = __builtin_amdgcn_s_memtime();
    float* d = a.mdist + ((long long)n * H + h) * S + wid * 8;
    d[0] = (float)(t_1 - t_0); d[1] = (float)(t_2 - t_1); d[2] = (float)(t_3 - t_2); d[3] = (float)(t_4 - t_3); d[4] = (float)(t_5 - t_0);
  }
}

// ---------------------------------------------------------------------------------------------
// backward.  DM: 0 no dropout, 1 dropout by hash, 2 dropout from the forward's bit mask
template <int DM>
__global__ __launch_bounds__(AS2_NT, 3) void attn_seq2_bwd_kernel(AttnBwdArgs a, const bf16* __restrict__ o_fwd, const unsigned* __restrict__ dropmask, int stagger) {
  constexpr int HD = 64, ND = 4;
  as2_stagger(stagger);
  extern __shared__ __attribute__((aligned(128))) char smem[];
  char* const imgA = smem;                                 // phase 1: Q, phase 2: K   ([160][64] dual-use images)
  char* const imgB = smem + AS2_IMG;                       // phase 1: dO, phase 2: V
  float* const sLse = (float*)(smem + 2 * AS2_IMG);        // lse * log2(e); +inf on padded rows
  float* const sDelta = sLse + AS2_ROWS;
  float* const sPos = sDelta + AS2_ROWS;
  float* const sMd = sPos + AS2_ROWS;
  float* const sKb = sMd + AS2_ROWS;                       // score bias of a key: 0, or -1e30 past S (role 2's accumulator start)
  unsigned* const sMask = (unsigned*)(sKb + AS2_ROWS);     // DM 2: [word 2][lane quarter 4][row 160]; DM 1: row terms [160]
  float* const sBias = (float*)(sMask + 2 * 4 * AS2_ROWS); // [3][64]
  float* const sRed = sBias + 192;                         // [AS2_NW] d alibi_scale partial sums

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int fr = lane & 15, g = lane >> 4;
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const long long orow = (long long)H * HD;
  const long long nh = (long long)n * H + h;
  const bf16* qkv = (const bf16*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const bf16* dO = (const bf16*)a.d_out + (long long)n * S * orow + (long long)h * HD;
  const bf16* Of = o_fwd + (long long)n * S * orow + (long long)h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float slope = a.slopes[h];
  const float asc = a.alibi_scale[h];
  const float coef2 = -slope * fmaxf(asc, 0.f) * LOG2E;
  const float scale2 = a.scale * LOG2E;
  const int nsub = (S + 15) >> 4;                          // 16-row groups with rows (<= 10)
  const int npair = (nsub + 1) >> 1;
  const unsigned scale_bits = __float_as_uint(a.drop.scale);

  // ---------------- prologue: Q / dO images by LDS-DMA; per-row constants; mask words; first key group's fragments
#pragma unroll
  for (int u = 0; u < 40 / AS2_NW; ++u) {
    const int piece = widu + AS2_NW * u;
    if (piece < 20) as2_dma_piece(qkv, rowstride, S, imgA, piece, lane);
    else as2_dma_piece(dO, orow, S, imgB, piece - 20, lane);
  }
  bf16x8 kf_n[2], vf_n[2];
  auto fetch_kv = [&](int grp, bf16x8* kd, bf16x8* vd) {
    const int kvc_ = min(grp * 16 + fr, S - 1);
    const bf16* krow_ = qkv + (long long)kvc_ * rowstride + (long long)H * HD + g * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kd[ks] = *(const bf16x8*)(krow_ + ks * 32);
      vd[ks] = *(const bf16x8*)(krow_ + (long long)H * HD + ks * 32);
    }
  };
  fetch_kv(wid, kf_n, vf_n);
  if (tid < AS2_ROWS) {
    const bool rok = tid < S;
    float dl = 0.f, lse_r = INFINITY, pos_r = 0.f, md_r = 0.f;         // exp2(x - inf) = 0: padded queries vanish
    if (rok) {
      uint4 oz[8], dz[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        oz[u] = *(const uint4*)(Of + (long long)tid * orow + u * 8);
        dz[u] = *(const uint4*)(dO + (long long)tid * orow + u * 8);
      }
      lse_r = a.lse[nh * S + tid] * LOG2E;
      pos_r = a.pos ? (float)a.pos[(long long)n * S + tid] : (float)tid;
      if (a.mdist) md_r = a.mdist[nh * S + tid];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bf16x8 ov = *(const bf16x8*)&oz[u];
        const bf16x8 dv_ = *(const bf16x8*)&dz[u];
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += (float)ov[i] * (float)dv_[i];
      }
    }
    sDelta[tid] = dl;
    sLse[tid] = lse_r;
    sPos[tid] = pos_r;
    sMd[tid] = md_r;
    sKb[tid] = rok ? 0.f : -1.0e30f;
    if constexpr (DM == 1) sMask[tid] = drop_row_term(a.drop.seed, (unsigned)(nh * S + min(tid, S - 1)));
  }
  if constexpr (DM == 2) {
#pragma unroll
    for (int u = 0; u < (4 * AS2_ROWS + AS2_NT - 1) / AS2_NT; ++u) {
      const int idx = tid + AS2_NT * u, q_ = idx >> 2, g_ = idx & 3;
      if (idx < 4 * AS2_ROWS) {
        uint2 w = make_uint2(0u, 0u);
        if (q_ < S) w = *(const uint2*)(dropmask + ((nh * AS2_ROWS + q_) * 4 + g_) * 2);
        sMask[g_ * AS2_ROWS + q_] = w.x;
        sMask[(4 + g_) * AS2_ROWS + q_] = w.y;
      }
    }
  }
  if (tid < 192) sBias[tid] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float dal = 0.f;
  // ================= phase 1 -- role 1: this wave's 16 keys on the lanes against ALL query rows: dK, dV
#pragma unroll 1
  for (int it = 0; it < (AS2_ROWS / 16 + AS2_NW - 1) / AS2_NW; ++it) {
    const int grp = wid + AS2_NW * it;
    if (grp * 16 >= S) break;                              // uniform
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { kf[ks] = kf_n[ks]; vf[ks] = vf_n[ks]; }
    if ((grp + AS2_NW) * 16 < S) fetch_kv(grp + AS2_NW, kf_n, vf_n);
    const int kv = grp * 16 + fr;
    const bool kv_ok = kv < S;
    const int kvc = kv_ok ? kv : S - 1;
    const float pos_kv = sPos[kvc];
    // mask words of this lane's key: word grp >> 3, lane quarter fr >> 2, bit 31 - (4 (grp & 7) + (fr & 3)) (7 - ... in word 1)
    const unsigned* const mrow = sMask + ((grp >> 3) * 4 + (fr >> 2)) * AS2_ROWS;
    const unsigned msh = (grp < 8 ? 31u : 7u) - (unsigned)(4 * (grp & 7) + (fr & 3));

    f32x4 dk[ND], dv[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    auto rows16 = [&](int rb, unsigned& p01, unsigned& p23, unsigned& d01, unsigned& d23) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 qa = *(const bf16x8*)(imgA + dual_off(rb + fr, ks * 4 + g));
        const bf16x8 da = *(const bf16x8*)(imgB + dual_off(rb + fr, ks * 4 + g));
        sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], pacc, 0, 0, 0);
      }
      const f32x4 lse4 = *(const f32x4*)(sLse + rb + 4 * g);
      const f32x4 del4 = *(const f32x4*)(sDelta + rb + 4 * g);
      const f32x4 pq4 = *(const f32x4*)(sPos + rb + 4 * g);
      uint4 mw = make_uint4(0u, 0u, 0u, 0u);
      if constexpr (DM == 2) mw = *(const uint4*)(mrow + rb + 4 * g);
      if constexpr (DM == 1) mw = *(const uint4*)(sMask + rb + 4 * g);
      const unsigned mws[4] = {mw.x, mw.y, mw.z, mw.w};
      float pv[4], dsv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dist = __builtin_fabsf(pq4[r] - pos_kv);
        const float e2 = __builtin_fmaf(sacc[r], scale2, __builtin_fmaf(dist, coef2, -lse4[r]));
        const float p = __builtin_amdgcn_exp2f(e2);
        float mlt = 1.f;                                   // dropout as ONE multiplier (0 or 1/(1-p)), the forward's mask
        if constexpr (DM == 2) mlt = __uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)mws[r], msh, 1u) & scale_bits);
        if constexpr (DM == 1) mlt = drop_keep2(a.drop, mws[r], (unsigned)kvc) ? a.drop.scale : 0.f;
        dsv[r] = p * __builtin_fmaf(pacc[r], mlt, -del4[r]);
        pv[r] = p * mlt;
      }
      p01 = as2_pack2(pv[0], pv[1]); p23 = as2_pack2(pv[2], pv[3]);
      d01 = as2_pack2(dsv[0], dsv[1]); d23 = as2_pack2(dsv[2], dsv[3]);
    };
#pragma unroll 1
    for (int pr = 0; pr < npair; ++pr) {
      union { unsigned u[4]; bf16x8 v; } pf, df;
      rows16(32 * pr, pf.u[0], pf.u[1], df.u[0], df.u[1]);
      rows16(32 * pr + 16, pf.u[2], pf.u[3], df.u[2], df.u[3]);
      // dV^T += dO^T P ; dK^T += Q^T dS      (k = q, permuted order shared by both operands)
      const int r_lo = 32 * pr + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int chl = 2 * i + (pp >> 1), sub = (pp & 1) << 3;
        const bf16x8 dof = tr_frag(imgB + dual_off(r_lo, chl) + sub, imgB + dual_off(r_hi, chl) + sub);
        const bf16x8 qf = tr_frag(imgA + dual_off(r_lo, chl) + sub, imgA + dual_off(r_hi, chl) + sub);
        dv[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf.v, dv[i], 0, 0, 0);
        dk[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, df.v, dk[i], 0, 0, 0);
      }
    }
    // dK, dV of this key group: D rows d = 16 i + 4 g + r, column key = fr
    if (kv_ok) {
      bf16* dkv = (bf16*)a.dqkv + ((long long)n * S + kv) * rowstride + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        f32x4 kk = dk[i] * a.scale;
        Vec4<bf16>::store(dkv + (long long)H * HD + 16 * i + 4 * g, kk);
        Vec4<bf16>::store(dkv + 2LL * H * HD + 16 * i + 4 * g, dv[i]);
      }
    }
    if (a.bias_sum) {          // column sums over this group's keys of what was stored (bf16-rounded), 16 key lanes -> 1
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sk = kv_ok ? (float)(bf16)(dk[i][r] * a.scale) : 0.f;
          float sv_ = kv_ok ? (float)(bf16)dv[i][r] : 0.f;
          sk = row16_sum(sk);
          sv_ = row16_sum(sv_);
          if (fr == 0) {
            atomicAdd(&sBias[64 + 16 * i + 4 * g + r], sk);
            atomicAdd(&sBias[128 + 16 * i + 4 * g + r], sv_);
          }
        }
    }
  }

  // ---------------- image switch: this wave's first query group (role 2) from global, then K / V over the Q / dO images
  bf16x8 qf_n[2], dof_n[2];
  auto fetch_qd = [&](int grp, bf16x8* qd, bf16x8* dd) {
    const int qc_ = min(grp * 16 + fr, S - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qd[ks] = *(const bf16x8*)(qkv + (long long)qc_ * rowstride + ks * 32 + g * 8);
      dd[ks] = *(const bf16x8*)(dO + (long long)qc_ * orow + ks * 32 + g * 8);
    }
  };
  fetch_qd(min(wid, nsub - 1), qf_n, dof_n);
  __syncthreads();                                         // every wave has finished reading the Q / dO images
#pragma unroll
  for (int u = 0; u < 40 / AS2_NW; ++u) {
    const int piece = widu + AS2_NW * u;
    if (piece < 20) as2_dma_piece(qkv + H * HD, rowstride, S, imgA, piece, lane);
    else as2_dma_piece(qkv + 2 * H * HD, rowstride, S, imgB, piece - 20, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ================= phase 2 -- role 2: this wave's 16 query rows on the lanes against ALL keys: dQ, d alibi_scale
  f32x4 bq[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) bq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned t16m1 = (a.drop.thresh >> 16) - 1u;
#pragma unroll 1
  for (int it = 0; it < (AS2_ROWS / 16 + AS2_NW - 1) / AS2_NW; ++it) {
    const int grp = wid + AS2_NW * it;
    if (grp * 16 >= S) break;                              // uniform
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { qf[ks] = qf_n[ks]; dof[ks] = dof_n[ks]; }
    if ((grp + AS2_NW) * 16 < S) fetch_qd(grp + AS2_NW, qf_n, dof_n);
    const int q = grp * 16 + fr;
    const float lse_q = sLse[q], del_q = sDelta[q], pos_q = sPos[q], md_q = sMd[q];     // (rows S .. 159: lse = +inf -> p = 0)
    unsigned w0 = 0u, w1 = 0u;
    if constexpr (DM == 2) { w0 = sMask[g * AS2_ROWS + q]; w1 = sMask[(4 + g) * AS2_ROWS + q]; }
    if constexpr (DM == 1) w0 = sMask[q];                  // row term

    f32x4 dq[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned wbits = w0;                                   // DM 2: the next score's decision is the TOP bit (the forward pushed them in this order)
#pragma unroll 1
    for (int pr = 0; pr < 5; ++pr) {                       // all ten key groups, always: keys past S start from -1e30 (p = 0, dS = 0)
      union { unsigned u[4]; bf16x8 v; } df;
      if (DM == 2 && pr == 4) wbits = w1 << 24;            // key groups 8 and 9: the eight bits of the second word
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * pr + jj;
        f32x4 sacc = *(const f32x4*)(sKb + 16 * j + 4 * g), pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 ka = *(const bf16x8*)(imgA + dual_off(16 * j + fr, ks * 4 + g));
          const bf16x8 va = *(const bf16x8*)(imgB + dual_off(16 * j + fr, ks * 4 + g));
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ks], sacc, 0, 0, 0);
          pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[ks], pacc, 0, 0, 0);
        }
        const f32x4 pk = *(const f32x4*)(sPos + 16 * j + 4 * g);
        unsigned hp[2] = {0u, 0u};
        if constexpr (DM == 1) {
          hp[0] = drop_hash_pair(w0, (unsigned)(16 * j + 4 * g));
          hp[1] = drop_hash_pair(w0, (unsigned)(16 * j + 4 * g + 2));
        }
        float dsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dist = __builtin_fabsf(pos_q - pk[r]);
          const float e2 = __builtin_fmaf(sacc[r], scale2, __builtin_fmaf(dist, coef2, -lse_q));
          const float p = __builtin_amdgcn_exp2f(e2);
          float mlt = 1.f;
          if constexpr (DM == 2) {
            mlt = __uint_as_float((unsigned)((int)wbits >> 31) & scale_bits);
            wbits += wbits;
          }
          if constexpr (DM == 1) {
            const unsigned h16 = (r & 1) ? (hp[r >> 1] >> 16) : (hp[r >> 1] & 0xFFFFu);
            mlt = __uint_as_float((unsigned)((int)(t16m1 - h16) >> 31) & scale_bits);
          }
          dsv[r] = p * __builtin_fmaf(pacc[r], mlt, -del_q);
          dal = __builtin_fmaf(-dsv[r], dist - md_q, dal);
        }
        df.u[2 * jj] = as2_pack2(dsv[0], dsv[1]);
        df.u[2 * jj + 1] = as2_pack2(dsv[2], dsv[3]);
      }
      // dQ^T += K^T dS^T   (A = K^T by transposed reads of the K image, k = key in the order of the packed dS^T)
      const int r_lo = 32 * pr + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int chl = 2 * i + (pp >> 1), sub = (pp & 1) << 3;
        const bf16x8 kt = tr_frag(imgA + dual_off(r_lo, chl) + sub, imgA + dual_off(r_hi, chl) + sub);
        dq[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, df.v, dq[i], 0, 0, 0);
      }
    }
    // dQ of this query group: D rows d = 16 i + 4 g + r, column q = fr; its column sums (of what is stored) ride in registers
    // until the wave's last item
    const float qv = q < S ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const f32x4 v = dq[i] * a.scale;
      union { unsigned u[2]; bf16x4 b; } pk2;
      pk2.u[0] = as2_pack2(v[0], v[1]);
      pk2.u[1] = as2_pack2(v[2], v[3]);
      if (q < S) *(bf16x4*)((bf16*)a.dqkv + ((long long)n * S + q) * rowstride + (long long)h * HD + 16 * i + 4 * g) = pk2.b;
      bq[i][0] = __builtin_fmaf(__uint_as_float(pk2.u[0] << 16), qv, bq[i][0]);
      bq[i][1] = __builtin_fmaf(__uint_as_float(pk2.u[0] & 0xFFFF0000u), qv, bq[i][1]);
      bq[i][2] = __builtin_fmaf(__uint_as_float(pk2.u[1] << 16), qv, bq[i][2]);
      bq[i][3] = __builtin_fmaf(__uint_as_float(pk2.u[1] & 0xFFFF0000u), qv, bq[i][3]);
    }
  }
  if (a.bias_sum) {
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sq = row16_sum(bq[i][r]);
        if (fr == 0) atomicAdd(&sBias[16 * i + 4 * g + r], sq);
      }
  }
  // ---------------- epilogue: bias sums and the d alibi_scale partial of this (n, h)
  dal = wave_sum(dal);
  if (lane == 0) sRed[wid] = dal;
  __syncthreads();
  if (a.bias_sum && tid < 192) atomicAdd(a.bias_sum + (long long)(tid >> 6) * H * HD + (long long)h * HD + (tid & 63), sBias[tid]);
  if (tid == 0 && a.dalibi) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < AS2_NW; ++w) sum += sRed[w];
    const float v = asc >= 0.f ? sum * slope : 0.f;
    if (a.dal_part) a.dal_part[((long long)h * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x] = v;
    else atomicAdd(a.dalibi + h, v);
  }
}
