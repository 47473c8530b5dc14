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
// Mask layout: per (n, h) 1280 words [word 2][lane quarter g 4][row q 160] = w0 / w1 of the forward lane (q, g) that owns keys 16 j + 4 g + r:
//   j < 8: bit 31 - (4 j + r) of w0;  j = 8, 9: bit 7 - (4 (j - 8) + r) of w1;  1 = kept.
#pragma once

#define AS2_ROWS 160
#define AS2_NW 4
#define AS2_NT (64 * AS2_NW)
#define AS2_IMG (AS2_ROWS * 128)
#define AS2F_NW 10                                  // forward: ten waves, one 16-row query group each
#define AS2F_NT (64 * AS2F_NW)
#define AS2F_BUF (2 * AS2_IMG + 2 * AS2_ROWS * 4)   // K image, V image, positions, key bias
#define AS2F_LDS (2 * AS2F_BUF)

// a per-sequence base pointer, kept in scalar registers and opaque: hipcc otherwise reassociates base + sequence offset + lane
// offset into (base + lane offset) + sequence offset and carries one 64-bit VGPR pair per global array through the whole kernel
template <typename P> __device__ __forceinline__ P as2_sbase(P p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  return (P)v;
}

// one 1-KiB piece (8 rows x 128 B) of a dual-use image by LDS-DMA: lane l lands at row 8 piece + (l >> 3), slot l & 7, and fetches
// the chunk that belongs there (dual_off: slot = chunk ^ (row & 7)); rows past S re-read row S - 1 (finite data, masked by lse / kv_ok)
__device__ __forceinline__ void as2_dma_piece(const bf16* __restrict__ base, long long rowstride, int S, char* img, int piece, int lane) {
  const int r = piece * 8 + (lane >> 3);
  const int rc = r < S ? r : S - 1;
  const int c = (lane & 7) ^ (r & 7);
  // (uniform 64-bit base + 32-bit lane offset: one address register per piece, no 64-bit vector arithmetic)
  const unsigned off = (unsigned)(rc * (int)rowstride + c * 8) * 2u;
  __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)((const char*)base + off), (af_lds_void_t*)(img + piece * 1024), 16, 0, 0);
}

__device__ __forceinline__ unsigned as2_pack2(float lo, float hi) {     // two fp32 -> packed bf16 pair (RNE), one instruction
  unsigned r;
  asm(A2V_ASM_CVT_PK " %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// O^T += V^T P^T of one 16-row query group against all 160 keys as ONE asm block (operands: %0-%3 the four output blocks, %4-%8 the
// five 32-key P^T fragments, %9 the lane's byte address in the V image): the V^T fragments come by ds_read_b64_tr_b16 two operands
// ahead of their MFMAs behind counted lgkmcnt waits, in fixed registers v152-v154 (addresses) and v156-v167 (a ring of three) that
// the compiler only sees as clobbers.  Why not the builtin: with the next sequence's LDS-DMA in flight hipcc puts s_waitcnt
// vmcnt(0) in front of it ("may alias").  Why not separate asm statements: hipcc may COPY an asm output before the wait that
// makes it valid (one build in four did: NaN outputs) -- inside one statement nothing can be inserted.
#define AS2_PV_ASM \
    "v_xor_b32 v152, 32, %9\n" \
    "v_xor_b32 v153, 64, %9\n" \
    "v_xor_b32 v154, 0x60, %9\n" \
    "ds_read_b64_tr_b16 v[156:157], %9 offset:0\n" \
    "ds_read_b64_tr_b16 v[158:159], %9 offset:2048\n" \
    "ds_read_b64_tr_b16 v[160:161], v152 offset:0\n" \
    "ds_read_b64_tr_b16 v[162:163], v152 offset:2048\n" \
    "ds_read_b64_tr_b16 v[164:165], v153 offset:0\n" \
    "ds_read_b64_tr_b16 v[166:167], v153 offset:2048\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %0, v[156:159], %4, %0\n" \
    "ds_read_b64_tr_b16 v[156:157], v154 offset:0\n" \
    "ds_read_b64_tr_b16 v[158:159], v154 offset:2048\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %1, v[160:163], %4, %1\n" \
    "ds_read_b64_tr_b16 v[160:161], %9 offset:4096\n" \
    "ds_read_b64_tr_b16 v[162:163], %9 offset:6144\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %2, v[164:167], %4, %2\n" \
    "ds_read_b64_tr_b16 v[164:165], v152 offset:4096\n" \
    "ds_read_b64_tr_b16 v[166:167], v152 offset:6144\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %3, v[156:159], %4, %3\n" \
    "ds_read_b64_tr_b16 v[156:157], v153 offset:4096\n" \
    "ds_read_b64_tr_b16 v[158:159], v153 offset:6144\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %0, v[160:163], %5, %0\n" \
    "ds_read_b64_tr_b16 v[160:161], v154 offset:4096\n" \
    "ds_read_b64_tr_b16 v[162:163], v154 offset:6144\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %1, v[164:167], %5, %1\n" \
    "ds_read_b64_tr_b16 v[164:165], %9 offset:8192\n" \
    "ds_read_b64_tr_b16 v[166:167], %9 offset:10240\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %2, v[156:159], %5, %2\n" \
    "ds_read_b64_tr_b16 v[156:157], v152 offset:8192\n" \
    "ds_read_b64_tr_b16 v[158:159], v152 offset:10240\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %3, v[160:163], %5, %3\n" \
    "ds_read_b64_tr_b16 v[160:161], v153 offset:8192\n" \
    "ds_read_b64_tr_b16 v[162:163], v153 offset:10240\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %0, v[164:167], %6, %0\n" \
    "ds_read_b64_tr_b16 v[164:165], v154 offset:8192\n" \
    "ds_read_b64_tr_b16 v[166:167], v154 offset:10240\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %1, v[156:159], %6, %1\n" \
    "ds_read_b64_tr_b16 v[156:157], %9 offset:12288\n" \
    "ds_read_b64_tr_b16 v[158:159], %9 offset:14336\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %2, v[160:163], %6, %2\n" \
    "ds_read_b64_tr_b16 v[160:161], v152 offset:12288\n" \
    "ds_read_b64_tr_b16 v[162:163], v152 offset:14336\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %3, v[164:167], %6, %3\n" \
    "ds_read_b64_tr_b16 v[164:165], v153 offset:12288\n" \
    "ds_read_b64_tr_b16 v[166:167], v153 offset:14336\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %0, v[156:159], %7, %0\n" \
    "ds_read_b64_tr_b16 v[156:157], v154 offset:12288\n" \
    "ds_read_b64_tr_b16 v[158:159], v154 offset:14336\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %1, v[160:163], %7, %1\n" \
    "ds_read_b64_tr_b16 v[160:161], %9 offset:16384\n" \
    "ds_read_b64_tr_b16 v[162:163], %9 offset:18432\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %2, v[164:167], %7, %2\n" \
    "ds_read_b64_tr_b16 v[164:165], v152 offset:16384\n" \
    "ds_read_b64_tr_b16 v[166:167], v152 offset:18432\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %3, v[156:159], %7, %3\n" \
    "ds_read_b64_tr_b16 v[156:157], v153 offset:16384\n" \
    "ds_read_b64_tr_b16 v[158:159], v153 offset:18432\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %0, v[160:163], %8, %0\n" \
    "ds_read_b64_tr_b16 v[160:161], v154 offset:16384\n" \
    "ds_read_b64_tr_b16 v[162:163], v154 offset:18432\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %1, v[164:167], %8, %1\n" \
    "s_waitcnt lgkmcnt(2)\n" \
    A2V_ASM_MFMA_16x16x32 " %2, v[156:159], %8, %2\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    A2V_ASM_MFMA_16x16x32 " %3, v[160:163], %8, %3\n" \
    "s_nop 7\n" \
    "s_nop 7\n"

// ---------------------------------------------------------------------------------------------
// forward, PERSISTENT and software pipelined.  What the phase stamps of a one-shot kernel showed (profiles/r06_attn_stamp.txt): a
// workgroup spends 7.1 k cycles waiting for its K / V images and then 7 k cycles per 16-row item at the VALU pipe's pace -- and
// since all workgroups of a launch are alike, fetch time and math time ADD (110 + 170 + 69 us store tail).  Here one workgroup per CU
// of TEN waves (one item each: the sequence's ten 16-row query groups) walks the (n, h) sequences b, b + G, b + 2G ... ; the next
// sequence's images arrive by LDS-DMA in the OTHER half of a double buffer while this one is computed, its Q rows and positions are
// fetched to registers meanwhile: one barrier per sequence, nothing fetched is waited for before a sequence's math is done.
template <bool DROP, bool MD, bool WM>
__global__ __launch_bounds__(AS2F_NT, 3) void attn_seq2_fwd_kernel(AttnArgs a, unsigned* __restrict__ dropmask) {
  typedef bf16 T;
  constexpr int HD = 64, ND = 4, NKG = 10;
  extern __shared__ __attribute__((aligned(128))) char smem[];   // two buffers of [K image | V image | positions | key bias]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int fr = lane & 15, g = lane >> 4;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const int npairs = a.N * H, G = gridDim.x;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = a.scale * LOG2E;
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const unsigned t16m1 = (a.drop.thresh >> 16) - 1u;
  const unsigned scale_bits = __float_as_uint(a.drop.scale);
  const int q = wid * 16 + fr;                             // this wave's item: query group wid, always
  const int qc = q < S ? q : S - 1;
  const bool active = wid * 16 < S;                        // (uniform) waves past the sequence only move data and meet the barriers

  // everything the NEXT sequence needs: images by DMA into buffer `buf`, positions and Q rows into registers
  // (nothing fetched here may be USED before the sequence's math is done: a use makes hipcc wait vmcnt(0) -- the DMA just issued)
  bf16x8 qf_n[2];
  int pos_n = 0;
  float slope_n = 0.f, asc_n = 0.f;
  auto fetch = [&](int p, int buf) {
    const int n_ = p / H, h_ = p - n_ * H;
    const T* qkv_ = as2_sbase((const T*)a.qkv + (long long)n_ * S * rowstride + (long long)h_ * HD);
    char* const base = smem + buf * AS2F_BUF;
#pragma unroll
    for (int u = 0; u < 40 / AS2F_NW; ++u) {
      const int piece = widu + AS2F_NW * u;                // 0 .. 39: K image pieces 0..19, V image pieces 20..39
      const int lp = piece < 20 ? piece : piece - 20;
      as2_dma_piece(qkv_ + (piece < 20 ? 1 : 2) * H * HD, rowstride, S, base + (piece - lp) * 1024, lp, lane);
    }
    if (tid < S) pos_n = a.pos ? a.pos[(long long)n_ * S + tid] : tid;
    slope_n = a.slopes[h_];
    asc_n = a.alibi_scale[h_];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf_n[ks] = *(const bf16x8*)(qkv_ + (long long)qc * rowstride + ks * 32 + g * 8);
  };
  auto publish = [&](int buf) {                            // per-key terms of the fetched sequence: position, and the score bias
    float* const sPos_ = (float*)(smem + buf * AS2F_BUF + 2 * AS2_IMG);      // 0 / -1e30 that removes keys past S (it is the S^T
    if (tid < AS2_ROWS) { sPos_[tid] = tid < S ? (float)pos_n : 0.f; sPos_[AS2_ROWS + tid] = tid < S ? 0.f : -1.0e30f; }   // accumulator's INITIAL value)
  };
  int p = blockIdx.x;
  if (p >= npairs) return;
  fetch(p, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  publish(0);
  __syncthreads();

#pragma unroll 1
  for (int t = 0; p < npairs; ++t, p += G) {
    const int cur = t & 1;
    const char* const sK = smem + cur * AS2F_BUF;
    const char* const sV = sK + AS2_IMG;
    const float* const sPos = (const float*)(sK + 2 * AS2_IMG);
    const float* const sKb = sPos + AS2_ROWS;
    const int n = p / H, h = p - n * H;
    bf16x8 qf[2];
    qf[0] = qf_n[0]; qf[1] = qf_n[1];
    const float c_rel = -slope_n * fmaxf(asc_n, 0.f) / a.scale;       // ALiBi slope in accumulator units
    const bool more = p + G < npairs;                      // (uniform)
    if (more) fetch(p + G, cur ^ 1);
    if (active) {
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
        for (int ks = 0; ks < 2; ++ks) acc = A2V_MFMA_16x16x32(kf[ks], qf[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[j][r] = __builtin_fmaf(__builtin_fabsf(pos_q - pk[r]), c_rel, acc[r]);
      }
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
        if (DROP) {                                        // keys 16 j + 4 g + {0,1} and + {2,3}: one hash per pair
          hp[0] = drop_hash_pair(rt, (unsigned)(16 * j + 4 * g));
          hp[1] = drop_hash_pair(rt, (unsigned)(16 * j + 4 * g + 2));
        }
        f32x4 kp = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MD) kp = *(const f32x4*)(sPos + 16 * j + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr_ = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], scale2, nm));
          ps += pr_;                                       // the normaliser is dropout-free
          if constexpr (MD) pa = __builtin_fmaf(__builtin_fabsf(pos_q - kp[r]), pr_, pa);
          if (DROP) {
            const unsigned h16 = (r & 1) ? (hp[r >> 1] >> 16) : (hp[r >> 1] & 0xFFFFu);
            const int d = (int)(t16m1 - h16);              // < 0  <=>  h16 >= thresh >> 16  <=>  kept
            pr_ *= __uint_as_float((unsigned)(d >> 31) & scale_bits);
            if constexpr (WM) {
              if (j < 8) w0 = __builtin_amdgcn_alignbit(w0, (unsigned)d, 31);   // (w << 1) | sign(d)
              else w1 = __builtin_amdgcn_alignbit(w1, (unsigned)d, 31);
            }
          }
          s[j][r] = pr_;
        }
      }
      const float l_row = rows_sum(ps);
      const float a_row = MD ? rows_sum(pa) : 0.f;
      // ---- O^T = V^T P^T over 32-key chunks.  The V^T fragments come by RAW-asm transposed reads, two operands ahead of their
      // MFMAs behind counted lgkmcnt waits: through the builtin hipcc puts s_waitcnt vmcnt(0) in front of the first of them -- the
      // next sequence's LDS-DMA, issued a moment ago, "may alias" -- and the pipeline is gone (the wait asm carries the registers
      // as operands, so no MFMA can be scheduled above it)
      f32x4 o[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      {
        const unsigned vb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(const_cast<char*>(sV)) +
                            (unsigned)((4 * g + qq) * 128 + ((((pp >> 1) ^ ((4 * g + qq) & 7)) & 7) << 4) + ((pp & 1) << 3));
        union { unsigned u[4]; bf16x8 v; } pf[5];
#pragma unroll
        for (int st = 0; st < 5; ++st) {
          pf[st].u[0] = as2_pack2(s[2 * st][0], s[2 * st][1]);
          pf[st].u[1] = as2_pack2(s[2 * st][2], s[2 * st][3]);
          pf[st].u[2] = as2_pack2(s[2 * st + 1][0], s[2 * st + 1][1]);
          pf[st].u[3] = as2_pack2(s[2 * st + 1][2], s[2 * st + 1][3]);
        }
        asm volatile(AS2_PV_ASM
                     : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3])
                     : "v"(pf[0].v), "v"(pf[1].v), "v"(pf[2].v), "v"(pf[3].v), "v"(pf[4].v), "v"(vb)
                     : "memory", "v152", "v153", "v154", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165",
                       "v166", "v167");
      }
      // ---- finish: O = O^T / l ; lse = ln2 * (m*scale2 + log2 l).  The next sequence's fetches are retired BEFORE this
      // sequence's stores go out (the vector-memory counter is in order: waiting behind the stores would expose their write latency
      // every sequence -- 17.8 k cycles per sequence instead of 8 k, measured)
      const float inv = 1.f / l_row;
#pragma unroll
      for (int i = 0; i < ND; ++i) o[i] = o[i] * inv;
      const float lse_v = (mx * scale2 + __builtin_amdgcn_logf(l_row)) * 0.6931471805599453f;
      const float md_v = a_row * inv;
      if (more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(cur ^ 1);
      }
      if (q < S) {
        T* orow = (T*)a.out + ((long long)n * S + q) * ((long long)H * HD) + (long long)h * HD;
#pragma unroll
        for (int i = 0; i < ND; ++i) Vec4<T>::store(orow + 16 * i + 4 * g, o[i]);
        if (g == 0 && a.lse) a.lse[((long long)n * H + h) * S + q] = lse_v;
        if (MD && g == 0) a.mdist[((long long)n * H + h) * S + q] = md_v;
        if constexpr (WM) {
          unsigned* const mb = dropmask + ((long long)n * H + h) * (AS2_ROWS * 8);
          mb[g * AS2_ROWS + q] = w0;
          mb[(4 + g) * AS2_ROWS + q] = w1;
        }
      }
    } else if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces, positions and Q rows of the next sequence
      publish(cur ^ 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave is done with this buffer; the other one is complete
    asm volatile("" ::: "memory");
  }
}

// ---------------------------------------------------------------------------------------------
// backward, PERSISTENT and software pipelined like the forward.  DM: 0 no dropout, 1 dropout by hash, 2 dropout from the forward's
// bit mask.  One workgroup per CU of TEN waves walks the (n, h) sequences b, b + G, ...; wave w owns key group w in role 1 and query
// group w in role 2.  LDS: buffer A (Q, dO images: role 1's operands), buffer B (K, V images: role 2's), two sets of per-row
// constants.  While role 1 of a sequence runs on A its K / V images land in B; while role 2 runs on B the NEXT sequence's Q / dO
// images land in A and its key-group fragments, O quarter rows (delta) and row constants arrive in registers:
//     role 1 (p) | barrier | DMA A(p+1) ... role 2 (p) | barrier | DMA B(p+1), delta and constants of p+1 | barrier | role 1 (p+1)
// The transposed-read products are single asm statements (see AS2_PV_ASM: no compiler wait, no copies of in-flight registers).
#define AS2_R1_ASM \
    "v_xor_b32 v153, 32, %10\n" \
    "v_xor_b32 v154, 64, %10\n" \
    "v_xor_b32 v155, 0x60, %10\n" \
    "ds_read_b64_tr_b16 v[156:157], %10 offset:20480\n" \
    "ds_read_b64_tr_b16 v[158:159], %10 offset:22528\n" \
    "ds_read_b64_tr_b16 v[160:161], %10 offset:0\n" \
    "ds_read_b64_tr_b16 v[162:163], %10 offset:2048\n" \
    "ds_read_b64_tr_b16 v[164:165], v153 offset:20480\n" \
    "ds_read_b64_tr_b16 v[166:167], v153 offset:22528\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %0, v[156:159], %8, %0\n" \
    "ds_read_b64_tr_b16 v[156:157], v153 offset:0\n" \
    "ds_read_b64_tr_b16 v[158:159], v153 offset:2048\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %4, v[160:163], %9, %4\n" \
    "ds_read_b64_tr_b16 v[160:161], v154 offset:20480\n" \
    "ds_read_b64_tr_b16 v[162:163], v154 offset:22528\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %1, v[164:167], %8, %1\n" \
    "ds_read_b64_tr_b16 v[164:165], v154 offset:0\n" \
    "ds_read_b64_tr_b16 v[166:167], v154 offset:2048\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %5, v[156:159], %9, %5\n" \
    "ds_read_b64_tr_b16 v[156:157], v155 offset:20480\n" \
    "ds_read_b64_tr_b16 v[158:159], v155 offset:22528\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %2, v[160:163], %8, %2\n" \
    "ds_read_b64_tr_b16 v[160:161], v155 offset:0\n" \
    "ds_read_b64_tr_b16 v[162:163], v155 offset:2048\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %6, v[164:167], %9, %6\n" \
    "s_waitcnt lgkmcnt(2)\n" \
    A2V_ASM_MFMA_16x16x32 " %3, v[156:159], %8, %3\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    A2V_ASM_MFMA_16x16x32 " %7, v[160:163], %9, %7\n" \
    "s_nop 7\n" \
    "s_nop 7\n"

#define AS2_R2_ASM \
    "v_xor_b32 v153, 32, %5\n" \
    "v_xor_b32 v154, 64, %5\n" \
    "v_xor_b32 v155, 0x60, %5\n" \
    "ds_read_b64_tr_b16 v[156:157], %5 offset:0\n" \
    "ds_read_b64_tr_b16 v[158:159], %5 offset:2048\n" \
    "ds_read_b64_tr_b16 v[160:161], v153 offset:0\n" \
    "ds_read_b64_tr_b16 v[162:163], v153 offset:2048\n" \
    "ds_read_b64_tr_b16 v[164:165], v154 offset:0\n" \
    "ds_read_b64_tr_b16 v[166:167], v154 offset:2048\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %0, v[156:159], %4, %0\n" \
    "ds_read_b64_tr_b16 v[156:157], v155 offset:0\n" \
    "ds_read_b64_tr_b16 v[158:159], v155 offset:2048\n" \
    "s_waitcnt lgkmcnt(4)\n" \
    A2V_ASM_MFMA_16x16x32 " %1, v[160:163], %4, %1\n" \
    "s_waitcnt lgkmcnt(2)\n" \
    A2V_ASM_MFMA_16x16x32 " %2, v[164:167], %4, %2\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    A2V_ASM_MFMA_16x16x32 " %3, v[156:159], %4, %3\n" \
    "s_nop 7\n" \
    "s_nop 7\n"

// Column sums of a stored [64 d x 16 row] bf16 tile (what a wave just packed for its dK / dV / dQ stores: lane (row fr, quarter g)
// holds d = 16 i + 4 g .. + 3 of block i as one 64-bit pair): the qkv Linear's bias gradient.  By DPP the reduction over the 16 row
// lanes cost 32 x ~8 vector instructions per key group -- a fifth of the backward's vector work.  Here the tile goes to a 2-KB LDS
// scratch of the wave as [row][d] blocks, comes back TRANSPOSED (rows in the MFMA's k dimension) and is multiplied with a ones
// operand (k < 16; zeros above): four MFMAs, no vector arithmetic, exact fp32 sums of the stored values.  The sums of block i leave
// as one ds_write_b128 per lane into the wave's own slots (all 16 row lanes write the same four values to the same address).
// Operands: %0-%3 the four packed blocks, %4 scratch write address (+ 32 fr + 8 g), %5 transposed-read address (+ 32 (8 (g & 1) +
// (fr >> 2)) + 8 (fr & 3)), %6 ones operand, %7 destination (+ 16 g).
#define AS2_COLSUM_ASM \
    "ds_write_b64 %4, %0 offset:0\n" \
    "ds_write_b64 %4, %1 offset:512\n" \
    "ds_write_b64 %4, %2 offset:1024\n" \
    "ds_write_b64 %4, %3 offset:1536\n" \
    "ds_read_b64_tr_b16 v[152:153], %5 offset:0\n" \
    "ds_read_b64_tr_b16 v[154:155], %5 offset:128\n" \
    "ds_read_b64_tr_b16 v[156:157], %5 offset:512\n" \
    "ds_read_b64_tr_b16 v[158:159], %5 offset:640\n" \
    "s_waitcnt lgkmcnt(2)\n" \
    A2V_ASM_MFMA_16x16x32 " v[160:163], v[152:155], %6, 0\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    A2V_ASM_MFMA_16x16x32 " v[164:167], v[156:159], %6, 0\n" \
    "ds_read_b64_tr_b16 v[152:153], %5 offset:1024\n" \
    "ds_read_b64_tr_b16 v[154:155], %5 offset:1152\n" \
    "ds_read_b64_tr_b16 v[156:157], %5 offset:1536\n" \
    "ds_read_b64_tr_b16 v[158:159], %5 offset:1664\n" \
    "s_nop 7\n" \
    "s_nop 7\n" \
    "s_nop 3\n" \
    "ds_write_b128 %7, v[160:163] offset:0\n" \
    "ds_write_b128 %7, v[164:167] offset:64\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    A2V_ASM_MFMA_16x16x32 " v[160:163], v[152:155], %6, 0\n" \
    A2V_ASM_MFMA_16x16x32 " v[164:167], v[156:159], %6, 0\n" \
    "s_nop 7\n" \
    "s_nop 7\n" \
    "s_nop 3\n" \
    "ds_write_b128 %7, v[160:163] offset:128\n" \
    "ds_write_b128 %7, v[164:167] offset:192\n"

#define AS2B_NW 10
#define AS2B_NT (64 * AS2B_NW)
#define AS2B_SET (5 * AS2_ROWS * 4 + 2 * 4 * AS2_ROWS * 4)      // lse2, delta, pos, md, key bias; mask words / row terms
#define AS2B_RAW (3 * AS2_ROWS * 4)                                 // lse, mean distance, position of the NEXT sequence as fetched
#define AS2B_LDS (5 * AS2_IMG + 2 * AS2B_SET + AS2B_RAW + AS2B_NW * 192 * 4 + 64 + AS2B_NW * 2048)
#define AS2_CLOB "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"

template <int DM>
__global__ __launch_bounds__(AS2B_NT, 3) void attn_seq2_bwd_kernel(AttnBwdArgs a, const bf16* __restrict__ o_fwd, const unsigned* __restrict__ dropmask) {
  constexpr int HD = 64, ND = 4;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  char* const imgQ = smem;                                 // buffer A: Q, dO  ([160][64] dual-use images)
  char* const imgDO = smem + AS2_IMG;
  char* const imgO = smem + 2 * AS2_IMG;                   // the NEXT sequence's forward output rows (delta); fetched with buffer A
  char* const imgK = smem + 3 * AS2_IMG;                   // buffer B: K, V
  char* const imgV = smem + 4 * AS2_IMG;
  char* const sets = smem + 5 * AS2_IMG;
  char* const sRaw = sets + 2 * AS2B_SET;
  float* const sBias = (float*)(sRaw + AS2B_RAW);      // [wave][3][64]: every wave's own column sums (plain stores)
  float* const sRed = sBias + AS2B_NW * 192;
  char* const sTile = (char*)(sRed + 16);                  // [wave][2 KB]: transposition scratch of the column sums                         // [AS2B_NW] d alibi_scale partial sums

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int fr = lane & 15, g = lane >> 4;
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const long long orow = (long long)H * HD;
  const int npairs = a.N * H, G = gridDim.x;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = a.scale * LOG2E;
  const int npair = (((S + 15) >> 4) + 1) >> 1;            // pairs of 16-row query groups with rows
  const unsigned scale_bits = __float_as_uint(a.drop.scale);
  const unsigned t16m1 = (a.drop.thresh >> 16) - 1u;
  const bool active = wid * 16 < S;                        // (uniform) this wave's key group / query group has rows
  const int row = wid * 16 + fr;                           // this lane's key (role 1) and query (role 2)
  const int rowc = row < S ? row : S - 1;
  const int drow = tid >> 2, dq4 = tid & 3;                // delta step: row and quarter of the row this thread sums

  // ---- what the NEXT sequence needs in registers (nothing of it may be used before the current role's math is done)
  bf16x8 kf_n[2], vf_n[2];                                 // role 1: K / V rows of this wave's key group
  float slope_n = 0.f, asc_n = 0.f;
  // Q / dO images into buffer A; with them everything else of that sequence that is not a register: its forward output rows, its
  // dropout mask (straight into constants set `c`: the forward wrote it in the [word][lane quarter][row] order role 1 reads) and
  // its lse / mean distance / position rows as fetched (4-byte LDS-DMA, one float per lane)
  auto dma_a = [&](int p, int c) {
    const int n_ = p / H, h_ = p - n_ * H;
    const long long nh_ = (long long)n_ * H + h_;
    const bf16* qkv_ = as2_sbase((const bf16*)a.qkv + (long long)n_ * S * rowstride + (long long)h_ * HD);
    const bf16* dO_ = as2_sbase((const bf16*)a.d_out + (long long)n_ * S * orow + (long long)h_ * HD);
    const bf16* of_ = as2_sbase(o_fwd + (long long)n_ * S * orow + (long long)h_ * HD);
#pragma unroll
    for (int u = 0; u < 60 / AS2B_NW; ++u) {
      const int piece = widu + AS2B_NW * u;                // 0 .. 59: Q, dO, O images of 20 pieces each, contiguous in LDS
      const bf16* const src = piece < 20 ? qkv_ : (piece < 40 ? dO_ : of_);          // (scalar selects: no branch, no waterfall)
      const long long rs = piece < 20 ? rowstride : orow;
      const int lp = piece < 20 ? piece : (piece < 40 ? piece - 20 : piece - 40);
      as2_dma_piece(src, rs, S, imgQ + (piece - lp) * 1024, lp, lane);
    }
    if constexpr (DM == 2) {
      if (widu < 5)
        __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)((const char*)as2_sbase(dropmask + nh_ * (AS2_ROWS * 8) + widu * 256) + (unsigned)(lane * 16)),
                                         (af_lds_void_t*)(sets + c * AS2B_SET + 5 * AS2_ROWS * 4 + widu * 1024), 16, 0, 0);
    }
    if (tid < AS2_ROWS) {                                  // (waves 0 .. 2; lane l of wave w lands at row 64 w + l)
      const unsigned ot = (unsigned)min(tid, S - 1) * 4u;
      __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)((const char*)as2_sbase(a.lse + nh_ * S) + ot), (af_lds_void_t*)(sRaw + widu * 256), 4, 0, 0);
      if (a.mdist) __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)((const char*)as2_sbase(a.mdist + nh_ * S) + ot), (af_lds_void_t*)(sRaw + AS2_ROWS * 4 + widu * 256), 4, 0, 0);
      if (a.pos) __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)((const char*)as2_sbase(a.pos + (long long)n_ * S) + ot), (af_lds_void_t*)(sRaw + 2 * AS2_ROWS * 4 + widu * 256), 4, 0, 0);
    }
  };
  auto dma_b = [&](int p) {
    const int n_ = p / H, h_ = p - n_ * H;
    const bf16* qkv_ = as2_sbase((const bf16*)a.qkv + (long long)n_ * S * rowstride + (long long)h_ * HD);
#pragma unroll
    for (int u = 0; u < 40 / AS2B_NW; ++u) {
      const int piece = widu + AS2B_NW * u;
      const int lp = piece < 20 ? piece : piece - 20;
      as2_dma_piece(qkv_ + (piece < 20 ? 1 : 2) * H * HD, rowstride, S, imgK + (piece - lp) * 1024, lp, lane);
    }
  };
  // per-lane BYTE offsets (32 bit, loop invariant) against per-sequence uniform bases: saddr + voffset loads / stores, no 64-bit VALU
  const unsigned offKV = (unsigned)(rowc * (int)rowstride + H * HD + g * 8) * 2u;      // K row of this lane's key (V: + H HD 2 bytes)
  const unsigned offRow = (unsigned)(rowc * (int)rowstride + 4 * g) * 2u;              // q slot of this lane's row, columns 4 g ..
  auto fetch_regs = [&](int p) {
    const int n_ = p / H, h_ = p - n_ * H;
    const char* const qb = as2_sbase((const char*)((const bf16*)a.qkv + (long long)n_ * S * rowstride + (long long)h_ * HD));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf_n[ks] = *(const bf16x8*)(qb + offKV + ks * 64);
      vf_n[ks] = *(const bf16x8*)(qb + offKV + H * HD * 2 + ks * 64);
    }
    slope_n = a.slopes[h_];
    asc_n = a.alibi_scale[h_];
  };
  // delta of the fetched sequence from its dO image (already in buffer A) and the O quarter rows; its constants into set `c`
  auto delta_step = [&](int p, int c) {
    float* const sLse_ = (float*)(sets + c * AS2B_SET);
    unsigned* const sMask_ = (unsigned*)(sLse_ + 5 * AS2_ROWS);
    float dl = 0.f;
    if (drow < S) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 ov = *(const bf16x8*)(imgO + dual_off(drow, 2 * dq4 + u));
        const bf16x8 dv_ = *(const bf16x8*)(imgDO + dual_off(drow, 2 * dq4 + u));
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += (float)ov[i] * (float)dv_[i];
      }
    }
    dl += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dl), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    dl += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dl), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    if (dq4 == 0) sLse_[AS2_ROWS + drow] = dl;             // (rows past S: 0)
    if (tid < AS2_ROWS) {
      const bool rok = tid < S;
      const float* const raw = (const float*)sRaw;
      sLse_[tid] = rok ? raw[tid] * LOG2E : INFINITY;      // exp2(x - inf) = 0: padded queries vanish
      sLse_[2 * AS2_ROWS + tid] = rok ? (a.pos ? (float)((const int*)raw)[2 * AS2_ROWS + tid] : (float)tid) : 0.f;
      sLse_[3 * AS2_ROWS + tid] = (rok && a.mdist) ? raw[AS2_ROWS + tid] : 0.f;
      sLse_[4 * AS2_ROWS + tid] = rok ? 0.f : -1.0e30f;    // score bias of a key past S (role 2's accumulator start)
      if constexpr (DM == 1) {
        const int n_ = p / H, h_ = p - n_ * H;
        sMask_[tid] = drop_row_term(a.drop.seed, (unsigned)(((long long)n_ * H + h_) * S + min(tid, S - 1)));
      }
    }
  };

  // column sums (AS2_COLSUM_ASM): this wave's scratch and destination slots
  const unsigned cs_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(sTile + wid * 2048);
  const unsigned cs_w = cs_base + (unsigned)(fr * 32 + g * 8);
  const unsigned cs_r = cs_base + (unsigned)((8 * (g & 1) + (fr >> 2)) * 32 + (fr & 3) * 8);
  const unsigned cs_o = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(sBias + wid * 192) + (unsigned)(g * 16);
  union { unsigned u[4]; bf16x8 v; } cs_ones_u;
#pragma unroll
  for (int i = 0; i < 4; ++i) cs_ones_u.u[i] = g < 2 ? (sizeof(bf16) == 2 && (float)(bf16)1.0f == 1.0f && __builtin_bit_cast(unsigned short, (bf16)1.0f) == 0x3F80 ? 0x3F803F80u : 0x3C003C00u) : 0u;
  const bf16x8 cs_ones = cs_ones_u.v;
  auto as2_colsum = [&](unsigned (*t)[2], unsigned wa, unsigned ra, const bf16x8& ones, unsigned oa) {
    const unsigned long long b0 = ((unsigned long long)t[0][1] << 32) | t[0][0], b1 = ((unsigned long long)t[1][1] << 32) | t[1][0];
    const unsigned long long b2 = ((unsigned long long)t[2][1] << 32) | t[2][0], b3 = ((unsigned long long)t[3][1] << 32) | t[3][0];
    asm volatile(AS2_COLSUM_ASM :: "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(wa), "v"(ra), "v"(ones), "v"(oa) : "memory", AS2_CLOB);
  };

  int p = blockIdx.x;
  if (p >= npairs) return;
  dma_a(p, 0);
  dma_b(p);
  fetch_regs(p);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  delta_step(p, 0);
  __syncthreads();

#pragma unroll 1
  for (int t = 0; p < npairs; ++t, p += G) {
    const int c = t & 1;
    const float* const sLse = (const float*)(sets + c * AS2B_SET);
    const float* const sDelta = sLse + AS2_ROWS;
    const float* const sPos = sDelta + AS2_ROWS;
    const float* const sMd = sPos + AS2_ROWS;
    const float* const sKb = sMd + AS2_ROWS;
    const unsigned* const sMask = (const unsigned*)(sKb + AS2_ROWS);
    const int n = p / H, h = p - n * H;
    const bool more = p + G < npairs;                      // (uniform)
    const float slope = slope_n, asc = asc_n;
    const float coef2 = -slope * fmaxf(asc, 0.f) * LOG2E;
    float dal = 0.f;
    bf16x8 qf[2], dof[2];

    // ================= role 1: this wave's 16 keys on the lanes against ALL query rows: dK, dV
    if (active) {
      bf16x8 kf[2], vf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kf[ks] = kf_n[ks]; vf[ks] = vf_n[ks]; }
      const float pos_kv = sPos[rowc];
      // mask words of this lane's key: word wid >> 3, lane quarter fr >> 2, bit 31 - (4 (wid & 7) + (fr & 3)) (7 - ... in word 1)
      const unsigned* const mrow = sMask + ((wid >> 3) * 4 + (fr >> 2)) * AS2_ROWS;
      const unsigned msh = (wid < 8 ? 31u : 7u) - (unsigned)(4 * (wid & 7) + (fr & 3));
      f32x4 dk[ND], dv[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      auto rows16 = [&](int rb, unsigned& p01, unsigned& p23, unsigned& d01, unsigned& d23) {
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 qa = *(const bf16x8*)(imgQ + dual_off(rb + fr, ks * 4 + g));
          const bf16x8 da = *(const bf16x8*)(imgDO + dual_off(rb + fr, ks * 4 + g));
          sacc = A2V_MFMA_16x16x32(qa, kf[ks], sacc, 0, 0, 0);
          pacc = A2V_MFMA_16x16x32(da, vf[ks], pacc, 0, 0, 0);
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
          const float pe = __builtin_amdgcn_exp2f(e2);
          float mlt = 1.f;                                 // dropout as ONE multiplier (0 or 1/(1-p)), the forward's mask
          if constexpr (DM == 2) mlt = __uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)mws[r], msh, 1u) & scale_bits);
          if constexpr (DM == 1) mlt = drop_keep2(a.drop, mws[r], (unsigned)rowc) ? a.drop.scale : 0.f;
          dsv[r] = pe * __builtin_fmaf(pacc[r], mlt, -del4[r]);
          pv[r] = pe * mlt;
        }
        p01 = as2_pack2(pv[0], pv[1]); p23 = as2_pack2(pv[2], pv[3]);
        d01 = as2_pack2(dsv[0], dsv[1]); d23 = as2_pack2(dsv[2], dsv[3]);
      };
      // lane part of the transposed-read addresses (Q image; the dO image is 20480 bytes further): row 4 g + qq of a 32-row
      // step, 16-byte chunk (pp >> 1) ^ (row & 7), + 8 (pp & 1) bytes; output block i flips address bits 5-6
      const unsigned tb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)imgQ +
                          (unsigned)((4 * g + qq) * 128 + ((((pp >> 1) ^ ((4 * g + qq) & 7)) & 7) << 4) + ((pp & 1) << 3));
#pragma unroll 1
      for (int pr = 0; pr < npair; ++pr) {
        union { unsigned u[4]; bf16x8 v; } pf, df;
        rows16(32 * pr, pf.u[0], pf.u[1], df.u[0], df.u[1]);
        rows16(32 * pr + 16, pf.u[2], pf.u[3], df.u[2], df.u[3]);
        // dV^T += dO^T P ; dK^T += Q^T dS      (k = q, permuted order shared by both operands)
        asm volatile(AS2_R1_ASM
                     : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(dv[3]), "+v"(dk[0]), "+v"(dk[1]), "+v"(dk[2]), "+v"(dk[3])
                     : "v"(pf.v), "v"(df.v), "v"(tb + (unsigned)(pr * 4096))
                     : "memory", AS2_CLOB);
      }
      // this wave's query group for role 2, while the Q / dO images are still here
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = *(const bf16x8*)(imgQ + dual_off(wid * 16 + fr, ks * 4 + g));
        dof[ks] = *(const bf16x8*)(imgDO + dual_off(wid * 16 + fr, ks * 4 + g));
      }
      // dK, dV of this key group: D rows d = 16 i + 4 g + r, column key = fr.  (The K / V images of role 2 -- DMA issued a whole
      // role ago -- are retired first: behind the stores the wait would expose their write latency.)
      unsigned kp[ND][2], vp[ND][2];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const f32x4 kk = dk[i] * a.scale;
        kp[i][0] = as2_pack2(kk[0], kk[1]); kp[i][1] = as2_pack2(kk[2], kk[3]);
        vp[i][0] = as2_pack2(dv[i][0], dv[i][1]); vp[i][1] = as2_pack2(dv[i][2], dv[i][3]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (row < S) {
        char* const db = as2_sbase((char*)((bf16*)a.dqkv + (long long)n * S * rowstride + (long long)h * HD));
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          *(uint2*)(db + offRow + H * HD * 2 + i * 32) = make_uint2(kp[i][0], kp[i][1]);
          *(uint2*)(db + offRow + H * HD * 4 + i * 32) = make_uint2(vp[i][0], vp[i][1]);
        }
      }
      if (a.bias_sum) {        // column sums over this group's keys of what was stored (AS2_COLSUM_ASM)
        if (wid * 16 + 16 > S) {                           // (uniform: the group with rows past S)
#pragma unroll
          for (int i = 0; i < ND; ++i) {
            if (row >= S) { kp[i][0] = 0u; kp[i][1] = 0u; vp[i][0] = 0u; vp[i][1] = 0u; }
          }
        }
        as2_colsum(kp, cs_w, cs_r, cs_ones, cs_o + 256);
        as2_colsum(vp, cs_w, cs_r, cs_ones, cs_o + 512);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // #1: every wave is done with buffer A; buffer B (K, V of p) is complete
    asm volatile("" ::: "memory");
    if (more) { dma_a(p + G, c ^ 1); fetch_regs(p + G); }

    // ================= role 2: this wave's 16 query rows on the lanes against ALL keys: dQ, d alibi_scale
    unsigned qp[ND][2];
    if (active) {
      const float lse_q = sLse[row], del_q = sDelta[row], pos_q = sPos[row], md_q = sMd[row];   // (rows S .. 159: lse = +inf -> p = 0)
      unsigned w0 = 0u, w1 = 0u;
      if constexpr (DM == 2) { w0 = sMask[g * AS2_ROWS + row]; w1 = sMask[(4 + g) * AS2_ROWS + row]; }
      if constexpr (DM == 1) w0 = sMask[row];              // row term
      f32x4 dq[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned tbk = (unsigned)(size_t)(__attribute__((address_space(3))) char*)imgK +
                           (unsigned)((4 * g + qq) * 128 + ((((pp >> 1) ^ ((4 * g + qq) & 7)) & 7) << 4) + ((pp & 1) << 3));
      unsigned wbits = w0;                                 // DM 2: the next score's decision is the TOP bit (the forward pushed them in this order)
#pragma unroll 1
      for (int pr = 0; pr < 5; ++pr) {                     // all ten key groups, always: keys past S start from -1e30 (p = 0, dS = 0)
        union { unsigned u[4]; bf16x8 v; } df;
        if (DM == 2 && pr == 4) wbits = w1 << 24;          // key groups 8 and 9: the eight bits of the second word
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * pr + jj;
          f32x4 sacc = *(const f32x4*)(sKb + 16 * j + 4 * g), pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 ka = *(const bf16x8*)(imgK + dual_off(16 * j + fr, ks * 4 + g));
            const bf16x8 va = *(const bf16x8*)(imgV + dual_off(16 * j + fr, ks * 4 + g));
            sacc = A2V_MFMA_16x16x32(ka, qf[ks], sacc, 0, 0, 0);
            pacc = A2V_MFMA_16x16x32(va, dof[ks], pacc, 0, 0, 0);
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
            const float pe = __builtin_amdgcn_exp2f(e2);
            float mlt = 1.f;
            if constexpr (DM == 2) {
              mlt = __uint_as_float((unsigned)((int)wbits >> 31) & scale_bits);
              wbits += wbits;
            }
            if constexpr (DM == 1) {
              const unsigned h16 = (r & 1) ? (hp[r >> 1] >> 16) : (hp[r >> 1] & 0xFFFFu);
              mlt = __uint_as_float((unsigned)((int)(t16m1 - h16) >> 31) & scale_bits);
            }
            dsv[r] = pe * __builtin_fmaf(pacc[r], mlt, -del_q);
            dal = __builtin_fmaf(-dsv[r], dist - md_q, dal);
          }
          df.u[2 * jj] = as2_pack2(dsv[0], dsv[1]);
          df.u[2 * jj + 1] = as2_pack2(dsv[2], dsv[3]);
        }
        // dQ^T += K^T dS^T   (A = K^T by transposed reads of the K image, k = key in the order of the packed dS^T)
        asm volatile(AS2_R2_ASM
                     : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3])
                     : "v"(df.v), "v"(tbk + (unsigned)(pr * 4096))
                     : "memory", AS2_CLOB);
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const f32x4 v = dq[i] * a.scale;
        qp[i][0] = as2_pack2(v[0], v[1]);
        qp[i][1] = as2_pack2(v[2], v[3]);
      }
    }
    // the next sequence's Q / dO images and registers are retired before this sequence's dQ goes out
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (active) {
      // dQ of this query group: D rows d = 16 i + 4 g + r, column q = fr
      if (row < S) {
        char* const db = as2_sbase((char*)((bf16*)a.dqkv + (long long)n * S * rowstride + (long long)h * HD));
#pragma unroll
        for (int i = 0; i < ND; ++i) *(uint2*)(db + offRow + i * 32) = make_uint2(qp[i][0], qp[i][1]);
      }
      if (a.bias_sum) {
        if (wid * 16 + 16 > S) {
#pragma unroll
          for (int i = 0; i < ND; ++i) {
            if (row >= S) { qp[i][0] = 0u; qp[i][1] = 0u; }
          }
        }
        as2_colsum(qp, cs_w, cs_r, cs_ones, cs_o);
      }
    }
    dal = wave_sum(dal);
    if (lane == 0) sRed[wid] = active ? dal : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // #2: every wave is done with buffer B; buffer A (Q, dO of p + G) is complete
    asm volatile("" ::: "memory");
    if (more) dma_b(p + G);
    // bias sums and the d alibi_scale partial of this (n, h)
    if (a.bias_sum && tid < 192) {
      const int nact = min((S + 15) >> 4, AS2B_NW);
      float sum = 0.f;
      for (int w = 0; w < nact; ++w) sum += sBias[w * 192 + tid];
      atomicAdd(a.bias_sum + (long long)(tid >> 6) * H * HD + (long long)h * HD + (tid & 63), sum);
    }
    if (tid == 0 && a.dalibi) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < AS2B_NW; ++w) sum += sRed[w];
      const float v = asc >= 0.f ? sum * slope : 0.f;
      if (a.dal_part) a.dal_part[(long long)h * a.N + n] = v;
      else atomicAdd(a.dalibi + h, v);
    }
    if (more) delta_step(p + G, c ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // #3: the next sequence's constants are published
    asm volatile("" ::: "memory");
  }
}
