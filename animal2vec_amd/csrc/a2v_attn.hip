// Fused multi-head attention with analytic ALiBi for gfx950 (reference modules.py:368-410 +
// base.py:292-314,553-698).  The (N,H,S,S) bias tensor of the reference is never built: the bias is
//     -slope_h * max(alibi_scale_h, 0) * |pos[n,i] - pos[n,j]|
// computed in-register from the original frame indices of the (possibly masked-out) tokens.
//
// Forward  : flash style, 64 query rows per workgroup (4 waves x 16), 64-key tiles through LDS.
//            Products are issued "swapped" (S^T = K Q^T, O^T = V^T P^T) so that a query row lives on
//            one lane column: softmax statistics are lane-local (+2 shuffles) and P never leaves
//            registers (its accumulator layout IS the next MFMA's B operand, k order permuted).
// Backward : one workgroup per 64-key block, loops over query tiles ("key on the lane"): dK^T, dV^T
//            accumulate in registers, only dS crosses LDS once for dQ, dQ is summed across key
//            blocks with fp32 atomics; d(alibi_scale) is reduced per head.
// qkv layout (N,S,3,H,HD) = the qkv Linear's output; O / dO layout (N,S,H*HD).
// T = bf16 -> v_mfma_f32_16x16x32_bf16 ; T = float -> v_mfma_f32_16x16x4_f32 (parity mode).
#include "a2v_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void af_lds_void_t;
typedef __attribute__((address_space(1))) const void af_gbl_cvoid_t;

template <typename T> struct AT {   // LDS tile addressing for [64][HD] tiles
  // row-read image (ds_read_b128 of 8 contiguous d): XOR swizzle of 16-B chunks
  // tr-read image (4 rows x 16 cols blocks): XOR swizzle of 32-B blocks
};

template <int HD> __device__ __forceinline__ int rowimg_off(int r, int chunk) {   // bf16, 16-B chunk index
  constexpr int NCH = HD / 8;
  return r * (HD * 2) + (((chunk ^ ((r >> 1) & (NCH - 1))) & (NCH - 1)) << 4);
}
template <int HD> __device__ __forceinline__ int trimg_off(int r, int col) {     // bf16 element (r,col)
  constexpr int NB = HD / 16;
  return r * (HD * 2) + ((((col >> 4) ^ ((r >> 1) & (NB - 1))) & (NB - 1)) << 5) + ((col & 15) << 1);
}

__device__ __forceinline__ bf16x8 tr_frag(const char* tile_lo, const char* tile_hi) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)tile_lo);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)tile_hi);
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

__device__ __forceinline__ int dual_off(int r, int chunk) { return r * 128 + ((chunk ^ (r & 7)) << 4); }   // [rows][64] bf16

struct AttnArgs {
  const void* qkv; void* out; float* lse;
  const int* pos;                 // (N,S) original frame index or nullptr (= arange)
  const float* slopes;            // (H)
  const float* alibi_scale;       // (H) learned, clamped at 0 here
  int N, S, H;
  float scale;
  Drop drop;                      // attention dropout on the softmax probabilities (modules.py:405)
  int xcd_order;                  // 1: 1-D launch, query blocks of one (n, h) share an XCD (attn_fwd_bf16_kernel)
  const float* qk_bound;          // (SPLIT, N, H, 2): max |q_i|^2 and max |k_j|^2 per (n, h) and row range, or nullptr: ALiBi band limit
  float* mdist;                   // (N,H,S) or nullptr: c_i = sum_j P_ij |pos_i - pos_j|, the row's mean attended distance (dropout-free P);
                                  // the backward centres |pos_i - pos_j| on it when it forms d alibi_scale (see AttnBwdArgs::mdist)
  float band_nats;                // ALiBi band limit: keys whose softmax weight is provably below exp(-band_nats) of the row sum are not visited
  unsigned* dropmask;             // (N,H,160,4,2) u32 or nullptr: the whole-sequence forward (S <= 160, bf16, HD 64) writes its dropout decisions here (a2v_attn_seq2.h)
  const float* qk_norm;           // host side only: (N, 3 H HD / 32) squared-norm maxima of the 32-column segments of q | k | v as the qkv product's
                                  // F_QKNORM epilogue left them, or nullptr (then attn_qk_bound_kernel reads q and k again)
};

struct AttnBwdArgs {
  const void* qkv; const void* d_out; const float* lse; const float* delta;
  void* dqkv;               // (N,S,3,H,HD): k and v slots written here
  float* dq32;              // (N,S,H,HD) fp32, zeroed by the caller, accumulated with atomics
  float* dalibi;            // (H) fp32 accumulated: d loss / d alibi_scale_h
  const int* pos; const float* slopes; const float* alibi_scale;
  int N, S, H;
  float scale;
  Drop drop;
  float* dal_part;          // per-workgroup partial sums of d alibi_scale go here (the stream's workspace; folded by dalibi_fold_kernel); NULL: atomics
  float* bias_sum;          // (3, H, HD) fp32 or NULL: column sums of dqkv (the qkv Linear's bias gradient), whole-sequence kernel only
  // (N,H,S) or NULL: c_i = sum_j P_ij |pos_i - pos_j| from the forward.  d alibi_scale_h = -slope_h sum_ij dS_ij |pos_i - pos_j|, and
  // sum_j dS_ij = 0 exactly, so any per-row constant may be subtracted from the distance.  What is computed is dS_ij = P_ij (dP_ij -
  // delta_i) with delta_i = dO_i . O_i from the ROUNDED forward output: it misses sum_j P_ij dP_ij by some e_i (bf16: 2^-9 relative),
  // which puts -e_i P_ij into every dS_ij and -e_i c_i into the row's distance sum -- e_i times the mean attended distance, a
  // systematic term as large as the (heavily cancelling) true sum at long sequences.  With the distance centred on c_i that term
  // is -e_i sum_j P_ij (|.| - c_i) = 0: the gradient is as accurate as the dS values themselves.
  const float* mdist;
  const unsigned* dropmask;  // the forward's dropout decisions (AttnArgs::dropmask) or nullptr: re-hash
};

#define FP32_PITCH(HD) ((HD) + 4)
#define A2V_QKB_SPLIT 16           // row ranges of attn_qk_bound_kernel (ALiBi band limit of the long-sequence forward)

// =============================================================================================
//                                            forward
// =============================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  constexpr bool BF = std::is_same<T, bf16>::value;
  constexpr int ND = HD / 16;                 // 16-wide d blocks
  constexpr int TILE_BYTES = BF ? 64 * HD * 2 : 64 * FP32_PITCH(HD) * 4;
  __shared__ __attribute__((aligned(16))) char sK[TILE_BYTES];
  __shared__ __attribute__((aligned(16))) char sV[TILE_BYTES];
  __shared__ float sPos[64];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 64, h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;                 // elements between tokens in qkv
  const T* qkv = (const T*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const float coef = -a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f);

  const int q = q0 + wid * 16 + fr;                          // this lane's query row
  const int qc = q < S ? q : S - 1;
  const float pos_q = a.pos ? (float)a.pos[(long long)n * S + qc] : (float)qc;

  // Q fragments (B operand of S^T = K Q^T): lane (q=fr, g) holds Q[q][8g+jj (+32ks)]
  bf16x8 qf_b[BF ? HD / 32 : 1];
  float qf_f[BF ? 1 : HD / 4];
  if constexpr (BF) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) qf_b[ks] = *(const bf16x8*)(qkv + (long long)qc * rowstride + ks * 32 + g * 8);
  } else {
#pragma unroll
    for (int ks = 0; ks < HD / 4; ++ks) qf_f[ks] = qkv[(long long)qc * rowstride + ks * 4 + g];
  }

  f32x4 o[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f, a_run = 0.f;

  const int ntile = (S + 63) / 64;
  for (int kt = 0; kt < ntile; ++kt) {
    const int kv0 = kt * 64;
    __syncthreads();   // previous tile fully consumed
    // ---- stage K and V tiles (zero fill past S)
    if constexpr (BF) {
      constexpr int NCH = HD / 8;                        // 16-B chunks per row
      for (int c = tid; c < 64 * NCH; c += 256) {
        const int r = c / NCH, ch = c % NCH;
        uint4 kz = {0u, 0u, 0u, 0u}, vz = {0u, 0u, 0u, 0u};
        if (kv0 + r < S) {
          const T* base = qkv + (long long)(kv0 + r) * rowstride + ch * 8;
          kz = *(const uint4*)(base + (long long)H * HD);
          vz = *(const uint4*)(base + 2LL * H * HD);
        }
        *(uint4*)(sK + rowimg_off<HD>(r, ch)) = kz;
        *(uint4*)(sV + trimg_off<HD>(r, ch * 8)) = vz;
      }
    } else {
      constexpr int NCH = HD / 4;
      for (int c = tid; c < 64 * NCH; c += 256) {
        const int r = c / NCH, ch = c % NCH;
        uint4 kz = {0u, 0u, 0u, 0u}, vz = {0u, 0u, 0u, 0u};
        if (kv0 + r < S) {
          const T* base = qkv + (long long)(kv0 + r) * rowstride + ch * 4;
          kz = *(const uint4*)(base + (long long)H * HD);
          vz = *(const uint4*)(base + 2LL * H * HD);
        }
        *(uint4*)((float*)sK + r * FP32_PITCH(HD) + ch * 4) = kz;
        *(uint4*)((float*)sV + r * FP32_PITCH(HD) + ch * 4) = vz;
      }
    }
    if (tid < 64) {
      const int kv = kv0 + tid;
      sPos[tid] = kv < S ? (a.pos ? (float)a.pos[(long long)n * S + kv] : (float)kv) : 0.f;
    }
    __syncthreads();

    // ---- S^T tiles: s[j][r] = score(kv = kv0+16j+4g+r, q)
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BF) {
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          const bf16x8 kf = *(const bf16x8*)(sK + rowimg_off<HD>(16 * j + fr, ks * 4 + g));
          acc = A2V_MFMA_16x16x32(kf, qf_b[ks], acc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < HD / 4; ++ks) {
          const float kf = ((const float*)sK)[(16 * j + fr) * FP32_PITCH(HD) + ks * 4 + g];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qf_f[ks], acc, 0, 0, 0);
        }
      }
      s[j] = acc;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kvl = 16 * j + 4 * g + r;
        float v = s[j][r] * a.scale + coef * fabsf(pos_q - sPos[kvl]);
        v = (kv0 + kvl < S) ? v : -INFINITY;
        s[j][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float ps = 0.f, pa = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __expf(s[j][r] - m_new);
        ps += p;                                  // the normaliser is dropout-free
        pa = fmaf(p, fabsf(pos_q - sPos[16 * j + 4 * g + r]), pa);
        if (a.drop.thresh) {
          const unsigned rt = drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qc));
          p = drop_keep2(a.drop, rt, (unsigned)(kv0 + 16 * j + 4 * g + r)) ? p * a.drop.scale : 0.f;
        }
        s[j][r] = p;
      }
    l_run = l_run * alpha + ps;
    a_run = a_run * alpha + pa;
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] *= alpha;

    // ---- O^T += V^T P^T
    if constexpr (BF) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (bf16)s[2 * st][r]; pf[4 + r] = (bf16)s[2 * st + 1][r]; }
        const int qq = (lane & 15) >> 2, pp = lane & 3;
        const int r_lo = 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const bf16x8 vf = tr_frag(sV + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sV + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
          o[i] = A2V_MFMA_16x16x32(vf, pf, o[i], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kvl = 16 * j + 4 * g + r;
#pragma unroll
          for (int i = 0; i < ND; ++i) {
            const float vf = ((const float*)sV)[kvl * FP32_PITCH(HD) + 16 * i + fr];
            o[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, s[j][r], o[i], 0, 0, 0);
          }
        }
    }
  }
  // ---- finish: O = O^T / l ; lse = m + log(l)
  float l_tot = l_run;
  l_tot += __shfl_xor(l_tot, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = 1.f / l_tot;
  if (q < S) {
    T* orow = (T*)a.out + ((long long)n * S + q) * ((long long)H * HD) + (long long)h * HD;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      f32x4 v = o[i] * inv;
      Vec4<T>::store(orow + 16 * i + 4 * g, v);
    }
    if (g == 0 && a.lse) a.lse[((long long)n * H + h) * S + q] = m_run + __logf(l_tot);
  }
  if (a.mdist) {
    float a_tot = a_run;
    a_tot += __shfl_xor(a_tot, 16, 64);
    a_tot += __shfl_xor(a_tot, 32, 64);
    if (q < S && g == 0) a.mdist[((long long)n * H + h) * S + q] = a_tot * inv;
  }
}

// ---------------------------------------------------------------------------------------------
// bf16x3 forward (dtype A2V_BF16X3: fp32 q / k / v / out, the parity mode on the bf16 matrix cores).  The kernel above with every
// product split: x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), S^T = K_hi Q_hi^T + K_lo Q_hi^T + K_hi Q_lo^T and
// O^T = V_hi^T P_hi^T + V_lo^T P_hi^T + V_hi^T P_lo^T -- three exact bf16 products per tile, fp32 accumulation, 2^-16 relative --
// 48 v_mfma_f32_16x16x32_bf16 (768 pipe cycles) per 64-key tile and wave against 128 v_mfma_f32_16x16x4_f32 (4096 cycles).
// K / V tiles are split when they are staged (four bf16 images in LDS), Q when it is loaded, P in registers; softmax is fp32.
__device__ __forceinline__ void split8(const float* v, bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = (bf16)v[i];
    l[i] = (bf16)(v[i] - (float)h[i]);
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_x3_kernel(AttnArgs a) {
  constexpr int ND = HD / 16, NCH = HD / 8;
  constexpr int TILE_BYTES = 64 * HD * 2;
  __shared__ __attribute__((aligned(16))) char sKh[TILE_BYTES];
  __shared__ __attribute__((aligned(16))) char sKl[TILE_BYTES];
  __shared__ __attribute__((aligned(16))) char sVh[TILE_BYTES];
  __shared__ __attribute__((aligned(16))) char sVl[TILE_BYTES];
  __shared__ float sPos[64];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 64, h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const float* qkv = (const float*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const float coef = -a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f);

  const int q = q0 + wid * 16 + fr;
  const int qc = q < S ? q : S - 1;
  const float pos_q = a.pos ? (float)a.pos[(long long)n * S + qc] : (float)qc;
  bf16x8 qh[HD / 32], ql[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks) {
    float v[8];
    Vec8<float>::load(qkv + (long long)qc * rowstride + ks * 32 + g * 8, v);
    split8(v, qh[ks], ql[ks]);
  }
  f32x4 o[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f, a_run = 0.f;

  const int ntile = (S + 63) / 64;
  // ALiBi band limit, as the bf16 long-sequence forward's (positions = arange, a.qk_bound given): key tiles wholly beyond
  // d_cut = (2B + 32) / (slope * alibi_scale) of every query row of this workgroup carry softmax weights below e^-32 of the row sum --
  // 3e-11 over 2000 keys, three orders below an fp32 ulp of the sums -- and are not visited.  B from the exact fp32 row norms.
  int kt_lo = 0, kt_end = ntile;
  if (!a.pos && a.qk_bound) {
    const float c_nat = a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f);
    if (c_nat > 0.f) {
      float q2 = 0.f, k2 = 0.f;
#pragma unroll
      for (int z = 0; z < A2V_QKB_SPLIT; ++z) {
        const float* const bz = a.qk_bound + (((long long)z * a.N + n) * H + h) * 2;
        q2 = fmaxf(q2, bz[0]);
        k2 = fmaxf(k2, bz[1]);
      }
      const float dcut = (2.f * sqrtf(q2 * k2) * a.scale * 1.01f + a.band_nats) / c_nat;
      if (dcut < 1.0e6f) {
        const int dc = (int)dcut + 1;
        kt_lo = max(0, (q0 - dc) / 64);
        kt_end = min(ntile, (q0 + 63 + dc) / 64 + 1);
        kt_lo = min(kt_lo, min(q0 / 64, ntile - 1));          // (the tile that holds the query rows is always visited)
        kt_end = max(kt_end, min(q0 / 64 + 1, ntile));
      }
    }
  }
  for (int kt = kt_lo; kt < kt_end; ++kt) {
    const int kv0 = kt * 64;
    __syncthreads();
    // (fetching the next tile's rows into registers under this tile's math was measured: +54 registers, three -> two waves per SIMD,
    //  197 against 195 ms per 8-clip step: not kept)
    for (int c = tid; c < 64 * NCH; c += 256) {
      const int r = c / NCH, ch = c % NCH;
      float kv_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, vv_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (kv0 + r < S) {
        const float* base = qkv + (long long)(kv0 + r) * rowstride + ch * 8;
        Vec8<float>::load(base + (long long)H * HD, kv_);
        Vec8<float>::load(base + 2LL * H * HD, vv_);
      }
      bf16x8 kh_, kl_, vh_, vl_;
      split8(kv_, kh_, kl_);
      split8(vv_, vh_, vl_);
      *(bf16x8*)(sKh + rowimg_off<HD>(r, ch)) = kh_;
      *(bf16x8*)(sKl + rowimg_off<HD>(r, ch)) = kl_;
      *(bf16x8*)(sVh + trimg_off<HD>(r, ch * 8)) = vh_;
      *(bf16x8*)(sVl + trimg_off<HD>(r, ch * 8)) = vl_;
    }
    if (tid < 64) {
      const int kv = kv0 + tid;
      sPos[tid] = kv < S ? (a.pos ? (float)a.pos[(long long)n * S + kv] : (float)kv) : 0.f;
    }
    __syncthreads();

    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const bf16x8 kh_ = *(const bf16x8*)(sKh + rowimg_off<HD>(16 * j + fr, ks * 4 + g));
        const bf16x8 kl_ = *(const bf16x8*)(sKl + rowimg_off<HD>(16 * j + fr, ks * 4 + g));
        acc = A2V_MFMA_16x16x32(kl_, qh[ks], acc, 0, 0, 0);
        acc = A2V_MFMA_16x16x32(kh_, ql[ks], acc, 0, 0, 0);
        acc = A2V_MFMA_16x16x32(kh_, qh[ks], acc, 0, 0, 0);
      }
      s[j] = acc;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kvl = 16 * j + 4 * g + r;
        float v = s[j][r] * a.scale + coef * fabsf(pos_q - sPos[kvl]);
        v = (kv0 + kvl < S) ? v : -INFINITY;
        s[j][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float ps = 0.f, pa = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __expf(s[j][r] - m_new);
        ps += p;
        pa = fmaf(p, fabsf(pos_q - sPos[16 * j + 4 * g + r]), pa);
        if (a.drop.thresh) {
          const unsigned rt = drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qc));
          p = drop_keep2(a.drop, rt, (unsigned)(kv0 + 16 * j + 4 * g + r)) ? p * a.drop.scale : 0.f;
        }
        s[j][r] = p;
      }
    l_run = l_run * alpha + ps;
    a_run = a_run * alpha + pa;
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] *= alpha;

#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8 ph, pl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p0 = s[2 * st][r], p1 = s[2 * st + 1][r];
        ph[r] = (bf16)p0; pl[r] = (bf16)(p0 - (float)ph[r]);
        ph[4 + r] = (bf16)p1; pl[4 + r] = (bf16)(p1 - (float)ph[4 + r]);
      }
      const int qq = (lane & 15) >> 2, pp = lane & 3;
      const int r_lo = 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const bf16x8 vh_ = tr_frag(sVh + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sVh + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
        const bf16x8 vl_ = tr_frag(sVl + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sVl + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
        o[i] = A2V_MFMA_16x16x32(vl_, ph, o[i], 0, 0, 0);
        o[i] = A2V_MFMA_16x16x32(vh_, pl, o[i], 0, 0, 0);
        o[i] = A2V_MFMA_16x16x32(vh_, ph, o[i], 0, 0, 0);
      }
    }
  }
  float l_tot = l_run;
  l_tot += __shfl_xor(l_tot, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = 1.f / l_tot;
  if (q < S) {
    float* orow = (float*)a.out + ((long long)n * S + q) * ((long long)H * HD) + (long long)h * HD;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      f32x4 v = o[i] * inv;
      Vec4<float>::store(orow + 16 * i + 4 * g, v);
    }
    if (g == 0 && a.lse) a.lse[((long long)n * H + h) * S + q] = m_run + __logf(l_tot);
  }
  if (a.mdist) {
    float a_tot = a_run;
    a_tot += __shfl_xor(a_tot, 16, 64);
    a_tot += __shfl_xor(a_tot, 32, 64);
    if (q < S && g == 0) a.mdist[((long long)n * H + h) * S + q] = a_tot * inv;
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 forward, software pipelined: the K/V tile kt+1 is fetched into registers while tile kt (in LDS) is
// consumed and is written into the other LDS buffer afterwards -> one barrier per tile and no exposed global
// latency (the un-pipelined kernel above sat in s_waitcnt/s_barrier for ~50 % of its wave cycles).
// Scores are kept in the log2 domain (scale*log2e and coef*log2e folded into one fma) so that p = exp2(s - m)
// is a v_sub + v_exp; the key-padding select only runs on the last tile.
__device__ __forceinline__ float max3f(float a, float b, float c) {   // no canonicalising v_max in front (MFMA / fma results are never sNaN)
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ float max2f(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// max over the lanes l, l^16, l^32, l^48 (the four 16-lane rows that hold one query's keys) on the VALU:
// v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows between two registers, no LDS round trip
// (ds_bpermute: two dependent ~100-cycle LDS trips per query block and tile, in front of the whole softmax)
__device__ __forceinline__ float rows_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float w = max2f(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const unsigned x = __float_as_uint(w);
  auto t = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return max2f(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

// QB = 16-row query blocks per wave (workgroup = 64*QB query rows): with QB = 2 every K / V fragment read from
// LDS feeds two MFMAs -- the QB = 1 kernel was LDS-bandwidth bound (16 KB of fragment reads per 16 MFMAs).
// MD: also form c_i = sum_j P_ij |pos_i - pos_j| (AttnArgs::mdist) -- calls that keep the LSE for a backward.
template <int HD, bool DROP, int QB, bool MSUM = false, bool MD = false>
__global__ __launch_bounds__(256, (QB == 2 && !DROP && !MD) ? 3 : 1) void attn_fwd_bf16_kernel(AttnArgs a) {
  typedef bf16 T;
  // MSUM (forward-only calls without dropout: the teacher; a2v_debug_set(18, 0) switches it off): row sums of P by a ones-MFMA over
  // the bf16 P, -4 % time.  Calls that keep the LSE for a backward use fp32 sums of the fp32 P on the VALU, as the reference's softmax.
  constexpr bool VSUM = DROP || !MSUM;
  constexpr int ND = HD / 16;
  constexpr int NCH = HD / 8;                 // 16-B chunks per row
  constexpr int TILE_BYTES = 64 * HD * 2;
  constexpr int CPT = 64 * NCH / 256;         // chunks per thread per tile (2 for HD 64, 1 for HD 32)
  __shared__ __attribute__((aligned(16))) char sK[2][TILE_BYTES];
  __shared__ __attribute__((aligned(16))) char sV[2][TILE_BYTES];
  __shared__ float sPos[2][64];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int S = a.S, H = a.H;
  int q0, h, n;
  if (a.xcd_order) {
    // 1-D launch, XCD-aware: workgroup ids go round-robin over the 8 XCDs, so ids with equal (id & 7) share an L2.  All query
    // blocks of one (n, h) get ids of one residue class: its K / V (512 KB at S = 2000) are fetched into ONE L2 instead of eight
    // (fabric reads per launch 2.26 GB -> see profiles/, 23 x the 98 MB of qkv before).
    const int nqb = (S + 64 * QB - 1) / (64 * QB);
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int hl = (k / nqb) * 8 + xcd;
    if (hl >= a.N * H) return;
    q0 = (k % nqb) * (64 * QB);
    n = hl / H;
    h = (hl % H + n) % H;        // rotate the heads over the XCDs from clip to clip: with the ALiBi band limit a head's work goes
                                 // with its slope, and a fixed head -> XCD map left the XCDs of the steep heads idle early
  } else {
    q0 = blockIdx.x * (64 * QB); h = blockIdx.y; n = blockIdx.z;
  }
  const long long rowstride = 3LL * H * HD;
  const T* qkv = (const T*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = a.scale * LOG2E;                                            // accumulator units -> log2 units
  const float c_rel = -a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f) / a.scale;       // ALiBi slope in accumulator units
  const float thr_acc = 4.f / scale2;

  int q[QB], qc[QB];
  float pos_q[QB], m_run[QB], l_run[QB], a_run[QB];
  bf16x8 qf[QB][HD / 32];
  f32x4 o[QB][ND];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    q[b] = q0 + (wid * QB + b) * 16 + fr;
    qc[b] = q[b] < S ? q[b] : S - 1;
    pos_q[b] = a.pos ? (float)a.pos[(long long)n * S + qc[b]] : (float)qc[b];
    m_run[b] = -INFINITY;
    l_run[b] = 0.f;
    a_run[b] = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) qf[b][ks] = *(const bf16x8*)(qkv + (long long)qc[b] * rowstride + ks * 32 + g * 8);
#pragma unroll
    for (int i = 0; i < ND; ++i) o[b][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  bf16x8 kreg[CPT], vreg[CPT];   // native vectors: an array of HIP uint4 structs that is not zero-initialised stays an alloca (scratch)
  const int ntile = (S + 63) / 64, nfull = S / 64;

  // ---- ALiBi without per-score arithmetic (positions = arange, the teacher).  For a key tile wholly BEFORE the workgroup's
  // query rows |q - k| = q - k, wholly AFTER it k - q, so the bias splits into a per-row constant, which softmax does not see,
  // and a per-key term that enters the QK MFMA as its C operand straight from LDS:
  //   before:  t = acc + c(q - k) = [acc - c(k - qref)] + R      R = c (q - qref), c = c_rel (< 0), qref = q0
  //   after :  t = acc + c(k - q) = [acc + c(k - qref)] - R
  // Only the QB tiles that overlap the query rows take the |.| form (in true units t).  m_run lives in the units of the
  // current phase and is shifted by R at the two phase changes (l and O are relative to it and do not move).  The producer of
  // tile kt+1 writes what that tile's phase consumes into sPos: mul * pos + add.  Explicit positions (a.pos, the student's kept
  // frames) are not assumed sorted: every tile is a |.| tile.
  const bool sorted = a.pos == nullptr;
  const int kd0 = sorted ? min(q0 / 64, ntile) : 0;             // tiles [kd0, kd1) overlap the query rows
  const int kd1 = sorted ? min(q0 / 64 + QB, ntile) : ntile;
  // ---- ALiBi band limit (positions = arange, a.qk_bound given).  With B >= |q_i . k_j| scale for every pair of this (n, h) and
  // c = slope_h max(alibi_scale_h, 0), a key j at distance |i - j| > d scores s_ij <= B - c d while the row's own diagonal key
  // scores s_ii >= -B: its softmax weight is below exp(2B - c d) of the row sum.  Beyond d_cut = (2B + L) / c that is e^-L per key
  // (AttnArgs::band_nats: L = 20 in the bf16 kernels -- 4e-6 of the row sum over 2000 keys, 500 times below the 2^-9 rounding of every
  // P the kernel feeds to the matrix cores; L = 32 in the bf16x3 kernel -- 3e-11, below one fp32 ulp of the sums the reference
  // forms), so key tiles wholly beyond d_cut of every query row of this workgroup are not visited.  B comes from the row norms (attn_qk_bound_kernel), so the limit
  // is a bound, not a heuristic: it follows the learned alibi_scale and the data (c -> 0 or large B: every tile is visited).
  int kt_lo = 0, kt_end = ntile;
  if (sorted && a.qk_bound) {
    const float c_nat = a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f);
    if (c_nat > 0.f) {
      float q2 = 0.f, k2 = 0.f;
#pragma unroll
      for (int z = 0; z < A2V_QKB_SPLIT; ++z) {
        const float* const bz = a.qk_bound + (((long long)z * a.N + n) * H + h) * 2;
        q2 = fmaxf(q2, bz[0]);
        k2 = fmaxf(k2, bz[1]);
      }
      const float bnd = sqrtf(q2 * k2) * a.scale * 1.01f;       // Cauchy-Schwarz, + 1 %
      const float dcut = (2.f * bnd + a.band_nats) / c_nat;
      if (dcut < 1.0e6f) {
        const int dc = (int)dcut + 1;
        kt_lo = max(0, (q0 - dc) / 64);
        kt_end = min(ntile, (q0 + 64 * QB - 1 + dc) / 64 + 1);
        kt_lo = min(kt_lo, kd0);                                // (the tiles that overlap the query rows are always visited)
        kt_end = max(kt_end, kd1);
      }
    }
  }
  const float qref = (float)q0;
  float pmul = 1.f, padd = 0.f;                                 // uniform: sPos = pmul * pos + padd for the tile being fetched
  auto pos_form = [&](int kt_) {
    if (kt_ < kd0) { pmul = -c_rel; padd = c_rel * qref; }
    else if (kt_ < kd1) { pmul = 1.f; padd = 0.f; }
    else { pmul = c_rel; padd = -c_rel * qref; }
  };

  // loop-invariant per-thread offsets of the tile copy (global elements, LDS bytes)
  unsigned goff[CPT];
  int kst[CPT], vst[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int c = tid + 256 * u;
    const int r = c / NCH, ch = c % NCH;
    goff[u] = (unsigned)(r * (int)rowstride + ch * 8);
    kst[u] = rowimg_off<HD>(r, ch);
    vst[u] = trimg_off<HD>(r, ch * 8);
  }
  // The fetch of tile kt+1 is ONLY loads into registers nothing else writes: no zero-initialised destination, no guard and no
  // conversion of the loaded position -- each of those made hipcc put an s_waitcnt vmcnt(0) right behind the loads (a VALU write
  // to a register a load of the previous trip may still target; a use of the loaded position), which parked every wave for the
  // full global latency once per tile (SQ_WAIT_ANY 50 % of the wave cycles).  Rows past S re-read row S-1 (finite data); their
  // scores are masked to -inf in the TAIL tile, so their V rows meet p = 0.
  int praw = 0;
#define AF_LOAD(kt_)                                                                              \
  {                                                                                               \
    const int kv0_ = (kt_) * 64;                                                                  \
    const T* tbase = qkv + (long long)kv0_ * rowstride;                                           \
    unsigned go_[CPT];                                                                            \
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) go_[u] = goff[u];                             \
    if ((kt_) >= nfull) {                            /* uniform, rare: ragged last tile */        \
      _Pragma("unroll") for (int u = 0; u < CPT; ++u) {                                           \
        const int c = tid + 256 * u;                                                              \
        go_[u] = (unsigned)(min(c / NCH, S - 1 - kv0_) * (int)rowstride + (c % NCH) * 8);         \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) {                                             \
      kreg[u] = *(const bf16x8*)(tbase + go_[u] + H * HD);                                         \
      vreg[u] = *(const bf16x8*)(tbase + go_[u] + 2 * H * HD);                                     \
    }                                                                                             \
    if (a.pos && tid < 64) praw = a.pos[(long long)n * S + min(kv0_ + tid, S - 1)];               \
  }
#define AF_STORE(kt_)                                                                             \
  {                                                                                               \
    const int buf_ = (kt_) & 1;                                                                   \
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) {                                             \
      *(bf16x8*)(sK[0] + buf_ * TILE_BYTES + kst[u]) = kreg[u];                                    \
      *(bf16x8*)(sV[0] + buf_ * TILE_BYTES + vst[u]) = vreg[u];                                    \
    }                                                                                             \
    if (tid < 64) {                                                                               \
      const int kv = (kt_) * 64 + tid;                                                            \
      pos_form(kt_);                                                                              \
      sPos[buf_][tid] = fmaf(a.pos ? (float)praw : (float)kv, pmul, padd);                        \
    }                                                                                             \
  }
  AF_LOAD(kt_lo);
  AF_STORE(kt_lo);
  __syncthreads();

  // row sums of P on the matrix cores (MSUM kernels): a ones fragment as the A operand of two more MFMAs per query
  // block gives sum_k p[k][q] in every row of the accumulator -- 32 v_add per tile and the final cross-lane reduction leave
  // the VALU, which is what bounds this kernel (SQ_ACTIVE_INST_VALU 83 % of the SIMD's cycles at 290 VALU per tile)
  f32x4 lacc[QB];
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
#pragma unroll
  for (int b = 0; b < QB; ++b) lacc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // One key tile.  PH: 0 = |.| form, 1 = per-key term only (before / after the query rows).  TAIL (compile-time): the padded
  // keys of a ragged last tile are masked to -inf -- as a run-time test inside one loop body the compiler if-converted it into
  // 32 v_cndmask + 15 v_cmp executed for EVERY tile.
  auto key_tile = [&](int kt, auto ph_tag, auto tail_tag) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int buf = kt & 1;
    const int kv0 = kt * 64;
    const bool more = kt + 1 < kt_end;
    if (more) AF_LOAD(kt + 1);
    const char* const bK = sK[buf];
    const char* const bV = sV[buf];

    // ---- S^T tiles in accumulator units (raw q.k + bias/scale): s[b][j][r] for kv = kv0+16j+4g+r
    f32x4 s[QB][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 kf[HD / 32];
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) kf[ks] = *(const bf16x8*)(bK + rowimg_off<HD>(16 * j + fr, ks * 4 + g));
      const f32x4 pk = *(const f32x4*)(&sPos[buf][16 * j + 4 * g]);
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        if constexpr (PH == 1) {
          f32x4 acc = pk;                            // the bias IS the accumulator's initial value
#pragma unroll
          for (int ks = 0; ks < HD / 32; ++ks) acc = A2V_MFMA_16x16x32(kf[ks], qf[b][ks], acc, 0, 0, 0);
          s[b][j] = acc;
        } else {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < HD / 32; ++ks) acc = A2V_MFMA_16x16x32(kf[ks], qf[b][ks], acc, 0, 0, 0);
          // acc + c_rel |pos_q - pos_k| = v_sub + one v_fma with the |.| source modifier.  This file is built with
          // -fno-slp-vectorize: paired into v_pk_fma_f32 (no modifiers) the abs costs a separate v_and per element and the
          // packed op more issue time next to MFMAs than two plain fmas (MI355X_MICROARCH.md cycle constants).
#pragma unroll
          for (int r = 0; r < 4; ++r) s[b][j][r] = __builtin_fmaf(__builtin_fabsf(pos_q[b] - pk[r]), c_rel, acc[r]);
        }
      }
    }
    if constexpr (PH == 1) {
      // The scores of this form are MFMA results read DIRECTLY by the inline-asm v_max3 / v_max below.  hipcc inserts the wait
      // states an XDL write needs before a VALU read (11 for this 8-pass MFMA) only in front of instructions it can see into,
      // not inline asm: depending on the schedule the row maximum was formed from registers the last MFMAs had not written yet
      // (garbage running maxima -> p underflow -> 0/0 with steep slopes; a build with two more live variables was correct).
      // The pad is tied to the results of the last four MFMAs; every earlier one is > 16 issue cycles behind.
      asm volatile("s_nop 7\n\ts_nop 4" : "+v"(s[0][3]), "+v"(s[QB - 1][3]), "+v"(s[0][2]), "+v"(s[QB - 1][2]));
    }
    if (TAIL) {   // last tile: padded keys -> -inf
#pragma unroll
      for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + 16 * j + 4 * g + r >= S) s[b][j][r] = -INFINITY;
    }
    float mx[QB];
    bool grow = false;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float v = max3f(s[b][0][0], s[b][0][1], s[b][0][2]);
      v = max3f(v, s[b][0][3], s[b][1][0]);
      v = max3f(v, s[b][1][1], s[b][1][2]);
      v = max3f(v, s[b][1][3], s[b][2][0]);
      v = max3f(v, s[b][2][1], s[b][2][2]);
      v = max3f(v, s[b][2][3], s[b][3][0]);
      v = max3f(v, s[b][3][1], s[b][3][2]);
      v = max2f(v, s[b][3][3]);
      mx[b] = rows_max(v);
      grow |= mx[b] > m_run[b] + thr_acc;
    }
    // deferred maximum: rescale only when some row's maximum grew by more than 2^4 (p stays <= 16: harmless
    // for fp32 sums and for bf16's scale-free relative precision); ONE wave-uniform decision for all query blocks
    if (__any(grow)) {
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        const float m_new = fmaxf(m_run[b], mx[b]);
        const float alpha = __builtin_amdgcn_exp2f((m_run[b] - m_new) * scale2);
        m_run[b] = m_new;
        if constexpr (VSUM) l_run[b] *= alpha; else lacc[b] *= alpha;
        if constexpr (MD) a_run[b] *= alpha;
#pragma unroll
        for (int i = 0; i < ND; ++i) o[b][i] *= alpha;
      }
    }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const float nm = -m_run[b] * scale2;
      float ps = 0.f, pa = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // key positions of this lane's four scores: sPos holds raw positions in the |.| tiles, the per-key BIAS in the
        // others (positions = arange there: the key's position is its index)
        f32x4 kp = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MD) {
          if constexpr (PH == 1) {
            const float k0 = (float)(kv0 + 16 * j + 4 * g);
            kp = (f32x4){k0, k0 + 1.f, k0 + 2.f, k0 + 3.f};
          } else {
            kp = *(const f32x4*)(&sPos[buf][16 * j + 4 * g]);
          }
        }
        unsigned hp[2] = {0u, 0u};
        if (DROP) {                                 // keys kv0 + 16 j + 4 g + {0,1} and + {2,3}: one hash per pair
          const unsigned rt = drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qc[b]));
          hp[0] = drop_hash_pair(rt, (unsigned)(kv0 + 16 * j + 4 * g));
          hp[1] = drop_hash_pair(rt, (unsigned)(kv0 + 16 * j + 4 * g + 2));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[b][j][r], scale2, nm));
          if (VSUM) ps += p;                        // the normaliser is dropout-free
          if constexpr (MD) pa = __builtin_fmaf(__builtin_fabsf(pos_q[b] - kp[r]), p, pa);
          if (DROP) {
            const unsigned h16 = (r & 1) ? (hp[r >> 1] >> 16) : (hp[r >> 1] & 0xFFFFu);
            p = h16 >= (a.drop.thresh >> 16) ? p * a.drop.scale : 0.f;
          }
          s[b][j][r] = p;
        }
      }
      if (VSUM) l_run[b] += ps;
      if constexpr (MD) a_run[b] += pa;
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8 pf[QB];
#pragma unroll
      for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[b][r] = (bf16)s[b][2 * st][r]; pf[b][4 + r] = (bf16)s[b][2 * st + 1][r]; }
      const int qq = (lane & 15) >> 2, pp = lane & 3;
      const int r_lo = 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const bf16x8 vf = tr_frag(bV + trimg_off<HD>(r_lo, 16 * i + 4 * pp), bV + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
#pragma unroll
        for (int b = 0; b < QB; ++b) o[b][i] = A2V_MFMA_16x16x32(vf, pf[b], o[b][i], 0, 0, 0);
      }
      if constexpr (!VSUM) {
#pragma unroll
        for (int b = 0; b < QB; ++b) lacc[b] = A2V_MFMA_16x16x32(ones, pf[b], lacc[b], 0, 0, 0);
      }
    }
    // buffer buf^1 was last read in iteration kt-1; every wave has passed that iteration's barrier
    if (more) AF_STORE(kt + 1);
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> PhAbs;
  typedef std::integral_constant<int, 1> PhKey;
  {
    int kt = kt_lo;
    for (; kt < kd0; ++kt) key_tile(kt, PhKey{}, std::false_type{});            // before the query rows: always whole tiles
    if (kd0 > 0) {                                                               // units t' -> t
#pragma unroll
      for (int b = 0; b < QB; ++b) m_run[b] += c_rel * (pos_q[b] - qref);
    }
    for (; kt < min(kd1, nfull); ++kt) key_tile(kt, PhAbs{}, std::false_type{});
    if (kt < kd1) { key_tile(kt, PhAbs{}, std::true_type{}); ++kt; }
    if (kt < kt_end) {                                                           // units t -> t''
#pragma unroll
      for (int b = 0; b < QB; ++b) m_run[b] += c_rel * (pos_q[b] - qref);
      for (; kt < min(nfull, kt_end); ++kt) key_tile(kt, PhKey{}, std::false_type{});
      if (kt < kt_end) key_tile(kt, PhKey{}, std::true_type{});               // (kt == nfull < ntile: the ragged last tile)
#pragma unroll
      for (int b = 0; b < QB; ++b) m_run[b] -= c_rel * (pos_q[b] - qref);        // back to true units for the LSE
    }
  }
#undef AF_LOAD
#undef AF_STORE
  // ---- finish: O = O^T / l ; lse = ln2 * (m*scale2 + log2 l)  (natural-log LSE, as the backward expects)
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    float l_tot;
    if constexpr (VSUM) {
      l_tot = l_run[b];
      l_tot += __shfl_xor(l_tot, 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
    } else {
      l_tot = lacc[b][0];                            // every accumulator row holds the whole row sum
    }
    const float inv = 1.f / l_tot;
    if (q[b] < S) {
      T* orow = (T*)a.out + ((long long)n * S + q[b]) * ((long long)H * HD) + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        f32x4 v = o[b][i] * inv;
        Vec4<T>::store(orow + 16 * i + 4 * g, v);
      }
      if (g == 0 && a.lse)
        a.lse[((long long)n * H + h) * S + q[b]] = (m_run[b] * scale2 + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }
    if constexpr (MD) {
      float a_tot = a_run[b];
      a_tot += __shfl_xor(a_tot, 16, 64);
      a_tot += __shfl_xor(a_tot, 32, 64);
      if (q[b] < S && g == 0) a.mdist[((long long)n * H + h) * S + q[b]] = a_tot * inv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16, HD = 64, long sequences, round 5: the same kernel on 32 x 32 x 16 MFMAs.  attn_fwd_bf16_kernel<64, ., 2> is bound by the
// SIMD's vector ISSUE port (SQ_ACTIVE_INST_VALU 68 %, SQ_WAIT_INST_ANY 40 % of the wave cycles): per 64-key tile a wave issues
// 36 v_mfma_f32_16x16x32_bf16, each holding the port for 8 of its 16 cycles, next to ~180 VALU instructions.  A
// v_mfma_f32_32x32x16_bf16 does twice the work for the same 8 issue cycles.  Here a wave owns 32 query rows as ONE 32-wide
// column block: S^T = K Q^T as two 32 x 32 tiles per key tile (a lane holds 32 scores of ONE query: one permlane32 swap
// finishes the row maximum instead of two swaps per 16-row block), O^T = V^T P^T as two 32-dim row blocks.  20 MFMA issues per
// tile instead of 36, the running statistics once per lane instead of twice.  LDS images, tile fetch, phases (ALiBi as the C
// operand outside the diagonal tiles), band limit and deferred maximum are the 16 x 16 kernel's.
//   S^T accumulator (jj = 0, 1: keys 32 jj ..): element i of lane (l31, h2) = key 32 jj + 8 (i >> 2) + 4 h2 + (i & 3), query l31
//   P^T as the next B operand: MFMA (jj, t) takes elements 8 t .. 8 t + 7: k-slot (h2, e) = key 32 jj + 16 t + 8 (e >> 2) + 4 h2 + (e & 3)
//   V^T as its A operand: 16-lane group (h2, dhalf) reads keys 32 jj + 16 t + {0, 8} + 4 h2 + 0..3 at dims 16 dd + 32 dhalf + 0..15
//   (the two groups of one 32-lane half read column blocks that differ by 2: with the image's XOR swizzle their banks are disjoint)
//   O^T accumulator dd: element i = dim 16 dd + 32 (i >> 3) + 8 ((i >> 2) & 1) + 4 h2 + (i & 3), query l31
#define A32_LDS (4 * 64 * 64 * 2 + 2 * 64 * 4)
template <bool DROP, bool MSUM = false, bool MD = false>
__global__ __launch_bounds__(256, (!DROP && !MD) ? 3 : 1) void attn_fwd32_bf16_kernel(AttnArgs a) {
  typedef bf16 T;
  constexpr int HD = 64;
  constexpr bool VSUM = DROP || !MSUM;
  constexpr int NCH = HD / 8;
  constexpr int TILE_BYTES = 64 * HD * 2;
  constexpr int CPT = 64 * NCH / 256;
  extern __shared__ __attribute__((aligned(128))) char smem[];     // A32_LDS bytes (the fragment addresses XOR bits 5-6 of a row base)
  char (*const sK)[TILE_BYTES] = (char (*)[TILE_BYTES])smem;
  char (*const sV)[TILE_BYTES] = (char (*)[TILE_BYTES])(smem + 2 * TILE_BYTES);
  float (*const sPos)[64] = (float (*)[64])(smem + 4 * TILE_BYTES);   // per-key term of key k of the tile at (k & 32) | 16 ((k >> 2) & 1) | 4 ((k >> 3) & 3) | (k & 3):
                                                                       // the 16 keys of a lane's accumulator are 64 contiguous bytes

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h2 = lane >> 5;
  const int S = a.S, H = a.H;
  int q0, h, n;
  if (a.xcd_order) {
    const int nqb = (S + 127) / 128;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int hl = (k / nqb) * 8 + xcd;
    if (hl >= a.N * H) return;
    q0 = (k % nqb) * 128;
    n = hl / H;
    h = (hl % H + n) % H;
  } else {
    q0 = blockIdx.x * 128; h = blockIdx.y; n = blockIdx.z;
  }
  const long long rowstride = 3LL * H * HD;
  const T* qkv = (const T*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = a.scale * LOG2E;
  const float c_rel = -a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f) / a.scale;
  const float thr_acc = 4.f / scale2;

  const int q = q0 + wid * 32 + l31;
  const int qc = q < S ? q : S - 1;
  const float pos_q = a.pos ? (float)a.pos[(long long)n * S + qc] : (float)qc;
  float m_run = -INFINITY, l_run = 0.f, a_run = 0.f;
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qkv + (long long)qc * rowstride + kk * 16 + h2 * 8);
  f32x16 o[2];
#pragma unroll
  for (int dd = 0; dd < 2; ++dd)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dd][i] = 0.f;

  const int ntile = (S + 63) / 64, nfull = S / 64;
  const bool sorted = a.pos == nullptr;
  const int kd0 = sorted ? min(q0 / 64, ntile) : 0;
  const int kd1 = sorted ? min(q0 / 64 + 2, ntile) : ntile;
  int kt_lo = 0, kt_end = ntile;
  if (sorted && a.qk_bound) {                                    // ALiBi band limit: see attn_fwd_bf16_kernel
    const float c_nat = a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f);
    if (c_nat > 0.f) {
      float q2 = 0.f, k2 = 0.f;
#pragma unroll
      for (int z = 0; z < A2V_QKB_SPLIT; ++z) {
        const float* const bz = a.qk_bound + (((long long)z * a.N + n) * H + h) * 2;
        q2 = fmaxf(q2, bz[0]);
        k2 = fmaxf(k2, bz[1]);
      }
      const float bnd = sqrtf(q2 * k2) * a.scale * 1.01f;
      const float dcut = (2.f * bnd + a.band_nats) / c_nat;
      if (dcut < 1.0e6f) {
        const int dc = (int)dcut + 1;
        kt_lo = max(0, (q0 - dc) / 64);
        kt_end = min(ntile, (q0 + 127 + dc) / 64 + 1);
        kt_lo = min(kt_lo, kd0);
        kt_end = max(kt_end, kd1);
      }
    }
  }
  const float qref = (float)q0;
  float pmul = 1.f, padd = 0.f;
  auto pos_form = [&](int kt_) {
    if (kt_ < kd0) { pmul = -c_rel; padd = c_rel * qref; }
    else if (kt_ < kd1) { pmul = 1.f; padd = 0.f; }
    else { pmul = c_rel; padd = -c_rel * qref; }
  };

  // Tile fetch by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at consecutive LDS addresses): wave w, instruction u
  // fills LDS rows 8 (2 w + u) .. + 8 of the K and of the V image; the images' XOR swizzles are applied on the GLOBAL side (lane l
  // fetches the chunk that belongs at position l & 7 of its row).  No staging registers, no ds_write, no per-tile address VALU:
  // the tile's base is scalar arithmetic, the lane's offsets are loop-invariant.
  unsigned koff[CPT], voff[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int r = (wid * CPT + u) * 8 + (lane >> 3), p_ = lane & 7;
    const int kc = (p_ ^ (r >> 1)) & 7;                                   // rowimg_off: position = chunk ^ ((r >> 1) & 7)
    const int vc = ((((p_ >> 1) ^ (r >> 1)) & 3) << 1) | (p_ & 1);        // trimg_off: 32-B block = (col >> 4) ^ ((r >> 1) & 3)
    koff[u] = (unsigned)((r * (int)rowstride + kc * 8) * 2);
    voff[u] = (unsigned)((r * (int)rowstride + vc * 8) * 2);
  }
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  int praw = 0;
#define A32_LOAD(kt_)                                                                             \
  {                                                                                               \
    const int kv0_ = (kt_) * 64;                                                                  \
    const int buf_ = (kt_) & 1;                                                                   \
    const char* const kb_ = (const char*)(qkv + (long long)kv0_ * rowstride + H * HD);            \
    const char* const vb_ = kb_ + H * HD * 2;                                                     \
    unsigned ko_[CPT], vo_[CPT];                                                                  \
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) { ko_[u] = koff[u]; vo_[u] = voff[u]; }       \
    if ((kt_) >= nfull) {                            /* uniform, rare: ragged last tile */        \
      _Pragma("unroll") for (int u = 0; u < CPT; ++u) {                                           \
        const int r = (wid * CPT + u) * 8 + (lane >> 3), rc = min(r, S - 1 - kv0_);               \
        ko_[u] = koff[u] - (unsigned)((r - rc) * (int)rowstride * 2);                             \
        vo_[u] = voff[u] - (unsigned)((r - rc) * (int)rowstride * 2);                             \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) {                                             \
      __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)(kb_ + ko_[u]), (af_lds_void_t*)(sK[0] + buf_ * TILE_BYTES + (widu * CPT + u) * 1024), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((af_gbl_cvoid_t*)(vb_ + vo_[u]), (af_lds_void_t*)(sV[0] + buf_ * TILE_BYTES + (widu * CPT + u) * 1024), 16, 0, 0); \
    }                                                                                             \
    if (a.pos && tid < 64) praw = a.pos[(long long)n * S + min(kv0_ + tid, S - 1)];               \
  }
#define A32_STORE(kt_)                                                                            \
  {                                                                                               \
    const int buf_ = (kt_) & 1;                                                                   \
    if (tid < 64) {                                                                               \
      const int kv = (kt_) * 64 + tid;                                                            \
      pos_form(kt_);                                                                              \
      sPos[buf_][(tid & 32) | ((tid & 4) << 2) | ((tid >> 1) & 12) | (tid & 3)] = fmaf(a.pos ? (float)praw : (float)kv, pmul, padd); \
    }                                                                                             \
  }
  A32_LOAD(kt_lo);
  A32_STORE(kt_lo);
  __syncthreads();

  f32x16 lacc;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
  const unsigned rt = DROP ? drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qc)) : 0u;
  const int qq = (lane & 15) >> 2, pp = lane & 3, dhalf = (lane >> 4) & 1;
  // lane parts of the LDS addresses.  K fragment (kk, jj): row 32 jj + l31, chunk (2 kk + h2) ^ ((l31 >> 1) & 7) = (2 kk) ^ y:
  // byte offset l31 * 128 + 16 y, XOR 32 kk (row bases are multiples of 128), + 4096 jj as the immediate.
  const unsigned krow_off = (unsigned)(l31 * 128 + ((h2 ^ ((l31 >> 1) & 7)) << 4));
  const unsigned kxor[4] = {0u, 32u, 64u, 96u};
  // V^T fragment (step, dd, lo / hi): row 16 step + 4 h2 + qq (+ 8), 32-B block (dd + 2 dhalf) ^ ((2 h2 + (qq >> 1)) & 3), + 8 pp bytes;
  // dd flips bit 5, step and lo / hi are immediates.
  const unsigned vrow_off = (unsigned)((4 * h2 + qq) * 128 + ((((2 * dhalf) ^ (2 * h2 + (qq >> 1))) & 3) << 5) + 8 * pp);

  auto key_tile = [&](int kt, auto ph_tag, auto tail_tag) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int buf = kt & 1;
    const int kv0 = kt * 64;
    const bool more = kt + 1 < kt_end;
    if (more) A32_LOAD(kt + 1);
    const char* const bK = sK[buf];
    const char* const bV = sV[buf];

    // ---- S^T tiles in accumulator units
    f32x16 s[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      if constexpr (PH == 1) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {                 // the per-key bias IS the accumulator's initial value.  (Four 16-B loads: as ONE
          // 64-B vector load hipcc puts s_waitcnt vmcnt(0) -- the tile DMA issued a moment ago -- in front of it.)
          const f32x4 pk = *(const f32x4*)(&sPos[buf][32 * jj + 16 * h2 + 4 * ii]);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[jj][4 * ii + r] = pk[r];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[jj][i] = 0.f;
      }
    }
    // K fragments by raw-asm ds_read_b128 THREE ahead of their MFMAs behind counted lgkmcnt waits (left to the compiler every
    // read sits directly in front of its MFMA behind lgkmcnt(0): eight exposed LDS round trips per tile); fragment i = (kk, jj) =
    // (i >> 1, i & 1): the two accumulator chains alternate.  An LDS operation the compiler issues in between only makes a
    // counted wait stricter (LDS returns in order), never weaker.
    {
      const unsigned kb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(const_cast<char*>(bK)) + krow_off;
      bf16x8 kr[5];
#define A32_KREAD(i_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kr[(i_) % 5]) : "v"(kb_lds ^ kxor[(i_) >> 1]), "n"(((i_) & 1) * 4096))
      A32_KREAD(0); A32_KREAD(1); A32_KREAD(2); A32_KREAD(3);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // fragment i + 4 goes out AFTER MFMA i: in the per-key form the first MFMA's C operand comes from the compiler's own sPos
        // reads, whose lgkmcnt(0) would otherwise wait for a fragment issued a moment ago
        if (i < 5) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(kr[i % 5]));
        else if (i == 5) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kr[i % 5]));
        else if (i == 6) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(kr[i % 5]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kr[i % 5]));
        s[i & 1] = A2V_MFMA_32x32x16(kr[i % 5], qf[i >> 1], s[i & 1], 0, 0, 0);
        if (i + 4 < 8) A32_KREAD(i + 4);
      }
#undef A32_KREAD
    }
    // V^T fragments: vr[step & 1][2 dd + {lo, hi}]; step 0's reads go out here, under the softmax
    const unsigned vb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(const_cast<char*>(bV)) + vrow_off;
    s16x4 vr[2][4];
#define A32_VREAD(st_)                                                                                                              \
  {                                                                                                                                 \
    _Pragma("unroll") for (int dd_ = 0; dd_ < 2; ++dd_) {                                                                           \
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[(st_) & 1][2 * dd_]) : "v"(vb_lds ^ (dd_ << 5)), "n"((st_) * 2048));        \
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[(st_) & 1][2 * dd_ + 1]) : "v"(vb_lds ^ (dd_ << 5)), "n"((st_) * 2048 + 1024)); \
    }                                                                                                                               \
  }
    A32_VREAD(0);
    if constexpr (PH == 0) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const f32x4 pk = *(const f32x4*)(&sPos[buf][32 * jj + 16 * h2 + 4 * ii]);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[jj][4 * ii + r] = __builtin_fmaf(__builtin_fabsf(pos_q - pk[r]), c_rel, s[jj][4 * ii + r]);
        }
    } else {
      // the inline-asm v_max3 below reads MFMA results the compiler cannot see a hazard for (see attn_fwd_bf16_kernel): a 16-pass
      // MFMA needs 18 wait states before a VALU read of its result
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s[0]), "+v"(s[1]));
    }
    if (TAIL) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kv0 + 32 * jj + 8 * (i >> 2) + 4 * h2 + (i & 3) >= S) s[jj][i] = -INFINITY;
    }
    float mx;
    {
      float v0 = max3f(s[0][0], s[0][1], s[0][2]);
      float v1 = max3f(s[1][0], s[1][1], s[1][2]);
      v0 = max3f(v0, s[0][3], s[0][4]);
      v1 = max3f(v1, s[1][3], s[1][4]);
      v0 = max3f(v0, s[0][5], s[0][6]);
      v1 = max3f(v1, s[1][5], s[1][6]);
      v0 = max3f(v0, s[0][7], s[0][8]);
      v1 = max3f(v1, s[1][7], s[1][8]);
      v0 = max3f(v0, s[0][9], s[0][10]);
      v1 = max3f(v1, s[1][9], s[1][10]);
      v0 = max3f(v0, s[0][11], s[0][12]);
      v1 = max3f(v1, s[1][11], s[1][12]);
      v0 = max3f(v0, s[0][13], s[0][14]);
      v1 = max3f(v1, s[1][13], s[1][14]);
      v0 = max3f(v0, s[0][15], s[1][15]);
      const float v = max2f(v0, v1);
      const unsigned u = __float_as_uint(v);
      auto rr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      mx = max2f(__uint_as_float(rr[0]), __uint_as_float(rr[1]));
    }
    const bool grow = mx > m_run + thr_acc;
    if (__any(grow)) {                                   // deferred maximum, one wave-uniform decision
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale2);
      m_run = m_new;
      if constexpr (VSUM) l_run *= alpha; else lacc *= alpha;
      if constexpr (MD) a_run *= alpha;
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) o[dd] *= alpha;
    }
    {
      const float nm = -m_run * scale2;
      float ps = 0.f, pa = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int kbase = kv0 + 32 * jj + 8 * ii + 4 * h2;
          f32x4 kp = {0.f, 0.f, 0.f, 0.f};
          if constexpr (MD) {
            if constexpr (PH == 1) {
              const float k0 = (float)kbase;
              kp = (f32x4){k0, k0 + 1.f, k0 + 2.f, k0 + 3.f};
            } else {
              kp = *(const f32x4*)(&sPos[buf][32 * jj + 16 * h2 + 4 * ii]);
            }
          }
          unsigned hp[2] = {0u, 0u};
          if (DROP) {
            hp[0] = drop_hash_pair(rt, (unsigned)kbase);
            hp[1] = drop_hash_pair(rt, (unsigned)(kbase + 2));
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(fmaf(s[jj][4 * ii + r], scale2, nm));
            if (VSUM && DROP) ps += p;                 // (the normaliser is dropout-free; without dropout the sums are formed beside the PV MFMAs)
            if constexpr (MD) pa = __builtin_fmaf(__builtin_fabsf(pos_q - kp[r]), p, pa);
            if (DROP) {
              const unsigned h16 = (r & 1) ? (hp[r >> 1] >> 16) : (hp[r >> 1] & 0xFFFFu);
              p = h16 >= (a.drop.thresh >> 16) ? p * a.drop.scale : 0.f;
            }
            s[jj][4 * ii + r] = p;
          }
        }
      if (VSUM && DROP) l_run += ps;
      if constexpr (MD) a_run += pa;
    }

    // ---- O^T += V^T P^T: step st = (jj, t) = (st >> 1, st & 1); its four transposing reads (two fragments) are issued one step
    // ahead by raw asm (the ds_read_tr builtin makes hipcc wait for the tile DMA, vmcnt(0), in front of the first of them)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      if (st + 1 < 4) A32_VREAD(st + 1);
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[e] = (bf16)s[st >> 1][8 * (st & 1) + e];
      if constexpr (VSUM && !DROP) {                     // this step's eight weights into the row sum, next to its MFMAs
        const int j_ = st >> 1, e_ = 8 * (st & 1);
        l_run += ((s[j_][e_] + s[j_][e_ + 1]) + (s[j_][e_ + 2] + s[j_][e_ + 3])) + ((s[j_][e_ + 4] + s[j_][e_ + 5]) + (s[j_][e_ + 6] + s[j_][e_ + 7]));
      }
      // (l_run rides along: the eight adds above stay in this step, beside its MFMAs, instead of sinking to the end of the tile)
      if (st + 1 < 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vr[st & 1][0]), "+v"(vr[st & 1][1]), "+v"(vr[st & 1][2]), "+v"(vr[st & 1][3]), "+v"(l_run));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[st & 1][0]), "+v"(vr[st & 1][1]), "+v"(vr[st & 1][2]), "+v"(vr[st & 1][3]), "+v"(l_run));
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        union { struct { s16x4 a, b; } h; bf16x8 v; } vf;
        vf.h.a = vr[st & 1][2 * dd]; vf.h.b = vr[st & 1][2 * dd + 1];
        o[dd] = A2V_MFMA_32x32x16(vf.v, pf, o[dd], 0, 0, 0);
      }
      if constexpr (!VSUM) lacc = A2V_MFMA_32x32x16(ones, pf, lacc, 0, 0, 0);
    }
#undef A32_VREAD
    if (more) A32_STORE(kt + 1);
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> PhAbs;
  typedef std::integral_constant<int, 1> PhKey;
  {
    int kt = kt_lo;
    for (; kt < kd0; ++kt) key_tile(kt, PhKey{}, std::false_type{});
    if (kd0 > 0) m_run += c_rel * (pos_q - qref);
    for (; kt < min(kd1, nfull); ++kt) key_tile(kt, PhAbs{}, std::false_type{});
    if (kt < kd1) { key_tile(kt, PhAbs{}, std::true_type{}); ++kt; }
    if (kt < kt_end) {
      m_run += c_rel * (pos_q - qref);
      for (; kt < min(nfull, kt_end); ++kt) key_tile(kt, PhKey{}, std::false_type{});
      if (kt < kt_end) key_tile(kt, PhKey{}, std::true_type{});
      m_run -= c_rel * (pos_q - qref);
    }
  }
#undef A32_LOAD
#undef A32_STORE
  float l_tot;
  if constexpr (VSUM) {
    const unsigned u = __float_as_uint(l_run);
    auto rr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    l_tot = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
  } else {
    l_tot = lacc[0];
  }
  const float inv = 1.f / l_tot;
  if (q < S) {
    T* orow = (T*)a.out + ((long long)n * S + q) * ((long long)H * HD) + (long long)h * HD;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        f32x4 v = {o[dd][4 * ii] * inv, o[dd][4 * ii + 1] * inv, o[dd][4 * ii + 2] * inv, o[dd][4 * ii + 3] * inv};
        Vec4<T>::store(orow + 16 * dd + 32 * (ii >> 1) + 8 * (ii & 1) + 4 * h2, v);
      }
    if (h2 == 0 && a.lse)
      a.lse[((long long)n * H + h) * S + q] = (m_run * scale2 + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  }
  if constexpr (MD) {
    const unsigned u = __float_as_uint(a_run);
    auto rr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a_tot = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    if (q < S && h2 == 0) a.mdist[((long long)n * H + h) * S + q] = a_tot * inv;
  }
}

// ---------------------------------------------------------------------------------------------
// bf16, HD = 64, S <= 160 (the student's kept frames: 140 in round 2's seeded masks, 143 .. 155 as the sampler draws them at 96 ..
// 768 rows): whole-sequence forward.  One workgroup of up to FOUR waves per (n, h) -- a wave without query rows leaves at once --;
// K and V of all S keys are staged once (the tiled kernel above re-fetched them for each of its three 64-row query blocks
// and padded 140 to 192 on both axes), each wave owns 48 query rows (three 16-row blocks) against every key, and the whole
// score row is in registers: ONE pass of softmax (no running maximum, no rescale), one barrier in the kernel.
#define AF48_ROWS 160             // K / V images: ten 16-key groups (an odd last group's MFMA partner is a zero group)
#define AF48_MAXS 160
#define AF48_LDS (2 * AF48_ROWS * 128 + AF48_ROWS * 4)
__device__ __forceinline__ float rows_sum(float v) {    // sum over the lanes l, l^16, l^32, l^48 (see rows_max)
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float w = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned x = __float_as_uint(w);
  auto t = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

#include "a2v_attn_seq2.h"

template <bool DROP, bool MD = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_seq48_kernel(AttnArgs a) {
  typedef bf16 T;
  constexpr int HD = 64, ND = 4, QB = 3, NKG = 10;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sK = smem;                                   // [160][64] row image
  char* const sV = sK + AF48_ROWS * 128;                   // [160][64] tr image
  float* const sPos = (float*)(sV + AF48_ROWS * 128);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const T* qkv = (const T*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = a.scale * LOG2E;
  const float c_rel = -a.slopes[h] * fmaxf(a.alibi_scale[h], 0.f) / a.scale;       // ALiBi slope in accumulator units
  const int nkg = (S + 15) >> 4;                           // 16-key groups with keys (<= 9)

  // ---- every global load of the kernel first: K / V rows (zero past S), positions, this wave's Q fragments
  {
    constexpr int NQ = (AF48_ROWS * 8 + 191) / 192;        // (192 or 256 threads: sized for the smaller launch)
    const int nthr = blockDim.x;
    uint4 kz[NQ], vz[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int c = tid + nthr * u, r = c >> 3, ch = c & 7;
      kz[u] = make_uint4(0u, 0u, 0u, 0u);
      vz[u] = kz[u];
      if (r < S) {
        const T* base = qkv + (long long)r * rowstride + ch * 8;
        kz[u] = *(const uint4*)(base + H * HD);
        vz[u] = *(const uint4*)(base + 2 * H * HD);
      }
    }
    if (tid < AF48_ROWS) sPos[tid] = tid < S ? (a.pos ? (float)a.pos[(long long)n * S + tid] : (float)tid) : 0.f;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int c = tid + nthr * u, r = c >> 3, ch = c & 7;
      if (c < AF48_ROWS * 8) {
        *(uint4*)(sK + rowimg_off<HD>(r, ch)) = kz[u];
        *(uint4*)(sV + trimg_off<HD>(r, ch * 8)) = vz[u];
      }
    }
  }
  int q[QB], qc[QB];
  float pos_q[QB];
  bf16x8 qf[QB][2];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    q[b] = (wid * QB + b) * 16 + fr;
    qc[b] = q[b] < S ? q[b] : S - 1;
    pos_q[b] = a.pos ? (float)a.pos[(long long)n * S + qc[b]] : (float)qc[b];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[b][ks] = *(const bf16x8*)(qkv + (long long)qc[b] * rowstride + ks * 32 + g * 8);
  }
  __syncthreads();
  if (wid * 48 >= S) return;                               // (S <= 96: a wave without query rows; no barrier follows)

  // ---- S^T in accumulator units: s[b][j][r] for key 16j + 4g + r, query block b
  f32x4 s[QB][NKG];
#pragma unroll
  for (int j = 0; j < NKG; ++j) {
    if (j < nkg) {                                         // uniform
      bf16x8 kf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[ks] = *(const bf16x8*)(sK + rowimg_off<HD>(16 * j + fr, ks * 4 + g));
      const f32x4 pk = *(const f32x4*)(sPos + 16 * j + 4 * g);
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = A2V_MFMA_16x16x32(kf[ks], qf[b][ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __builtin_fmaf(__builtin_fabsf(pos_q[b] - pk[r]), c_rel, acc[r]);
          s[b][j][r] = (16 * j + 4 * g + r < S) ? v : -INFINITY;       // only the last group has keys past S
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < QB; ++b) s[b][j] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
  }
  // ---- softmax over the whole row, dropout, P -> bf16 in place
  float m_row[QB], l_row[QB], a_row[QB];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    float mx = max3f(s[b][0][0], s[b][0][1], s[b][0][2]);
    mx = max2f(mx, s[b][0][3]);
#pragma unroll
    for (int j = 1; j < NKG; ++j) {
      mx = max3f(mx, s[b][j][0], s[b][j][1]);
      mx = max3f(mx, s[b][j][2], s[b][j][3]);
    }
    mx = rows_max(mx);
    m_row[b] = mx;
    const float nm = -mx * scale2;
    const unsigned rt = DROP ? drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qc[b])) : 0u;
    float ps = 0.f, pa = 0.f;
#pragma unroll
    for (int j = 0; j < NKG; ++j) {
      unsigned hp[2] = {0u, 0u};
      if (DROP) {                                   // keys 16 j + 4 g + {0,1} and + {2,3}: one hash per pair
        hp[0] = drop_hash_pair(rt, (unsigned)(16 * j + 4 * g));
        hp[1] = drop_hash_pair(rt, (unsigned)(16 * j + 4 * g + 2));
      }
      f32x4 kp = {0.f, 0.f, 0.f, 0.f};
      if constexpr (MD) kp = *(const f32x4*)(sPos + 16 * j + 4 * g);      // (rows past S: p = 0)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][j][r], scale2, nm));
        ps += p;                                    // the normaliser is dropout-free
        if constexpr (MD) pa = __builtin_fmaf(__builtin_fabsf(pos_q[b] - kp[r]), p, pa);
        if (DROP) {
          const unsigned h16 = (r & 1) ? (hp[r >> 1] >> 16) : (hp[r >> 1] & 0xFFFFu);
          p = h16 >= (a.drop.thresh >> 16) ? p * a.drop.scale : 0.f;
        }
        s[b][j][r] = p;
      }
    }
    l_row[b] = rows_sum(ps);
    a_row[b] = MD ? rows_sum(pa) : 0.f;
  }
  // ---- O^T = V^T P^T over 32-key chunks (the tenth key group is the zero rows 144..159 against p = 0)
  f32x4 o[QB][ND];
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int i = 0; i < ND; ++i) o[b][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
  for (int st = 0; st < 5; ++st) {
    if (32 * st < S) {                                     // uniform
      bf16x8 pf[QB];
#pragma unroll
      for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[b][r] = (bf16)s[b][2 * st][r];
          pf[b][4 + r] = 2 * st + 1 < NKG ? (bf16)s[b][2 * st + 1 < NKG ? 2 * st + 1 : 0][r] : (bf16)0.f;
        }
      const int r_lo = 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const bf16x8 vf = tr_frag(sV + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sV + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
#pragma unroll
        for (int b = 0; b < QB; ++b) o[b][i] = A2V_MFMA_16x16x32(vf, pf[b], o[b][i], 0, 0, 0);
      }
    }
  }
  // ---- finish: O = O^T / l ; lse = ln2 * (m*scale2 + log2 l)  (natural-log LSE, as the backward expects)
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const float inv = 1.f / l_row[b];
    if (q[b] < S) {
      T* orow = (T*)a.out + ((long long)n * S + q[b]) * ((long long)H * HD) + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        f32x4 v = o[b][i] * inv;
        Vec4<T>::store(orow + 16 * i + 4 * g, v);
      }
      if (g == 0 && a.lse)
        a.lse[((long long)n * H + h) * S + q[b]] = (m_row[b] * scale2 + __builtin_amdgcn_logf(l_row[b])) * 0.6931471805599453f;
      if (MD && g == 0) a.mdist[((long long)n * H + h) * S + q[b]] = a_row[b] * inv;
    }
  }
}

int g_a2v_attn_msum = 1;     // a2v_debug_set(18, v): forward-only long-sequence calls (the teacher) sum P on the matrix cores (bf16 P; -4 % kernel
                             // time; 0: fp32 sums on the VALU).  On since round 3: targets, every gradient and the backward error of d alibi_scale
                             // are unchanged at all four full-size comparisons (profiles/r03_parity_report.jsonl, rows "msum"); what round 2
                             // read as a loss of accuracy (d alibi_scale 0.16 -> 0.27 at T = 11 025) is that gradient's condition number (673)
                             // times rounding noise either way
int g_a2v_attn_xcd = 1;      // a2v_debug_set(17, v): long-sequence forward: XCD-aware 1-D workgroup order on / off
int g_a2v_attn_fwd32 = 1;    // a2v_debug_set(36, v): long-sequence bf16 forward on 32 x 32 x 16 MFMAs (0: the 16 x 16 x 32 kernel)
// The round-6 kernels always walk ten 16-row groups: they pay where the sequence HAS ten (145 .. 160 kept frames: what the sampler draws
// at the shipped recipe, 143 .. 155); with nine or fewer the round-5 kernels (three waves, 48-row blocks) waste less -- measured at
// S = 140: forward 320 against 333 us, backward 782 against 788 (profiles/r06_attn_bench.txt).  a2v_debug_set(39, 2) sends every
// S <= 160 to the round-6 kernels (tests), 0 none.
static inline bool as2_takes(int S);
int g_a2v_attn_seq2 = 1;     // a2v_debug_set(39, v): S <= 160 bf16 forward AND backward take the round-6 five-wave kernels (0: the round-5 ones)
static inline bool as2_takes(int S) { return S <= AS2_ROWS && (g_a2v_attn_seq2 == 2 || (g_a2v_attn_seq2 == 1 && S > 144)); }
int g_a2v_attn_fwd48 = 1;    // a2v_debug_set(16, v): S <= 144 bf16 forward takes the whole-sequence three-wave kernel (0: the tiled one)
// Ingredients of the bound of |q_i . k_j| * scale over all pairs of one (n, h): max_i |q_i|^2 and max_j |k_j|^2 (Cauchy-Schwarz in
// the attention kernel).  8 lanes share a row (16 B each); reads q and k once: 65 MB at 32 x 2000 x 16 heads.
#define A2V_QKB_MAX (A2V_WS_QKB_BYTES / 4)              // the bounds live at A2V_WS_QKB of the stream's workspace
template <int HD, typename T = bf16>
__global__ __launch_bounds__(256) void attn_qk_bound_kernel(const T* __restrict__ qkv, float* __restrict__ bound, int S, int H) {
  static_assert(HD == 64, "8 lanes x 8 elements per head row");
  // One workgroup = one clip (blockIdx.y) x one of A2V_QKB_SPLIT row ranges (blockIdx.x); a thread owns one 16-B chunk of the
  // [q | k] part of a row -- 2 x H x 8 chunks, read as one contiguous 4-KB piece per row at H = 16 (per-head workgroups read
  // 128-B pieces at a 6-KB stride: 1.3 TB/s) -- and keeps the running maximum of its head's |q|^2 or |k|^2 over the rows.
  const int n = blockIdx.y, tid = threadIdx.x;
  const int nch = 16 * H;                                  // chunks of q and of k per row: 8 H each
  const int rows_z = (S + A2V_QKB_SPLIT - 1) / A2V_QKB_SPLIT;
  const int rbeg = blockIdx.x * rows_z, rend = min(S, rbeg + rows_z);
  const long long rowstride = 3LL * H * HD;
  for (int c = tid; c < nch; c += 256) {
    const T* base = qkv + (long long)n * S * rowstride + c * 8;         // q chunks 0 .. 8H-1, k chunks 8H .. 16H-1: contiguous
    float mx = 0.f;
    for (int r0 = rbeg; r0 < rend; r0 += 8) {              // eight rows' loads in flight (rows past the range repeat its last row)
      Raw8<T> v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u].load(base + (long long)min(r0 + u, rend - 1) * rowstride);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float sq = 0.f, e8[8];
        v[u].unpack(e8);
#pragma unroll
        for (int i = 0; i < 8; ++i) sq += e8[i] * e8[i];
        // sum over the 8 lanes of the head row on the VALU (DPP: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror)
#define QKB_DPP_ADD(ctrl) sq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), ctrl, 0xF, 0xF, true))
        QKB_DPP_ADD(0xB1);
        QKB_DPP_ADD(0x4E);
        QKB_DPP_ADD(0x141);
#undef QKB_DPP_ADD
        mx = fmaxf(mx, sq);
      }
    }
    if ((c & 7) == 0 && rbeg < rend) {
      const int h = (c >> 3) % H, which = (c >> 3) / H;    // 0: q, 1: k
      bound[(((long long)blockIdx.x * gridDim.y + n) * H + h) * 2 + which] = mx;
    } else if ((c & 7) == 0) {
      const int h = (c >> 3) % H, which = (c >> 3) / H;
      bound[(((long long)blockIdx.x * gridDim.y + n) * H + h) * 2 + which] = 0.f;
    }
  }
}
// The same bound from the qkv product's own epilogue (F_QKNORM, a2v_gemm_p8.hip): per clip and 32-column segment the maximum over the
// rows of the segment's squared norm.  A head row is two segments a, b:  |q_i . k_j| <= |q_a,i| |k_a,j| + |q_b,i| |k_b,j| <=
// sqrt(max_i q_a,i^2 max_j k_a,j^2) + sqrt(max_i q_b,i^2 max_j k_b,j^2) =: B' -- a bound as before (Cauchy-Schwarz per segment; the
// maxima of a and b may sit in different rows, which is all it loses).  The forward kernels read (q2, k2) per (n, h) from
// A2V_QKB_SPLIT slots and form sqrt(max q2 * max k2): slot 0 gets (B'^2, 1), the others zero.
__global__ __launch_bounds__(256) void attn_qk_norm_fold_kernel(const float* __restrict__ norms, float* __restrict__ bound, int N, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // over (SPLIT, N, H, 2)
  if (idx >= A2V_QKB_SPLIT * N * H * 2) return;
  const int which = idx & 1, h = (idx >> 1) % H, n = ((idx >> 1) / H) % N, z = (idx >> 1) / (H * N);
  float v = 0.f;
  if (z == 0) {
    const float* const q = norms + (long long)n * (6 * H) + 2 * h;
    const float* const k = q + 2 * H;
    const float b = sqrtf(q[0] * k[0]) + sqrtf(q[1] * k[1]);
    v = which ? 1.f : b * b;
  }
  bound[idx] = v;
}
int g_a2v_attn_band = 1;     // a2v_debug_set(19, v): ALiBi band limit of the long-sequence forward (positions = arange) on / off;
                             // v >= 8: on, with exp(-v) as the bf16 kernel's weight bound (experiments)
#define A2V_BAND_NATS_BF16 20.f  // bf16 kernels: 2000 keys x e^-20 = 4e-6 of the row sum, 500 times below the bf16 rounding of P itself (2^-9)

template <int HD, bool DROP>
static void attn_fwd_bf16_launch(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  a.xcd_order = 0;
  if constexpr (HD == 64) {
    if (as2_takes(a.S)) {                                   // round 6: persistent ten-wave kernels (a2v_attn_seq2.h)
      const long long np = (long long)a.N * a.H;
      const dim3 grid((unsigned)(np < 256 ? np : 256));     // persistent: one workgroup per CU walks the (n, h) sequences
      const bool wm = DROP && a.dropmask != nullptr;
#define AS2F_GO(MD_, WM_)                                                                                                         \
  {                                                                                                                               \
    (void)hipFuncSetAttribute((const void*)attn_seq2_fwd_kernel<DROP, MD_, WM_>, hipFuncAttributeMaxDynamicSharedMemorySize, AS2F_LDS); \
    hipLaunchKernelGGL((attn_seq2_fwd_kernel<DROP, MD_, WM_>), grid, dim3(AS2F_NT), AS2F_LDS, st, a, a.dropmask);                  \
  }
      if (a.mdist) { if (wm) AS2F_GO(true, DROP) else AS2F_GO(true, false) }
      else AS2F_GO(false, false)                            // (no mask without the statistics a backward needs: forward-only call)
#undef AS2F_GO
      return;
    }
    if (a.S <= AF48_MAXS && g_a2v_attn_fwd48) {
      const int nthr = a.S > 144 ? 256 : 192;             // 48 query rows per wave
      if (a.mdist) hipLaunchKernelGGL((attn_fwd_seq48_kernel<DROP, true>), dim3(1, a.H, a.N), dim3(nthr), AF48_LDS, st, a);
      else hipLaunchKernelGGL((attn_fwd_seq48_kernel<DROP, false>), dim3(1, a.H, a.N), dim3(nthr), AF48_LDS, st, a);
      return;
    }
  }
  if (a.S >= 512) {   // long sequences (the teacher's 2000 frames): 128 query rows per workgroup
    const long long nqb = (a.S + 127) / 128, heads8 = ((long long)a.N * a.H + 7) / 8 * 8;
    a.qk_bound = nullptr;
    if constexpr (HD == 64) {
      char* const ws = (g_a2v_attn_band && !a.pos && (long long)a.N * a.H * 2 * A2V_QKB_SPLIT <= A2V_QKB_MAX) ? a2v_stream_ws(st) : nullptr;
      if (ws) {
        float* const bp = (float*)(ws + A2V_WS_QKB);
        if (a.qk_norm) hipLaunchKernelGGL(attn_qk_norm_fold_kernel, dim3((A2V_QKB_SPLIT * a.N * a.H * 2 + 255) / 256), dim3(256), 0, st, a.qk_norm, bp, a.N, a.H);
        else hipLaunchKernelGGL((attn_qk_bound_kernel<64>), dim3(A2V_QKB_SPLIT, a.N), dim3(256), 0, st, (const bf16*)a.qkv, bp, a.S, a.H);
        a.qk_bound = bp;
      }
    }
    const bool xo = g_a2v_attn_xcd && heads8 * nqb < (1ll << 31);
    a.xcd_order = xo ? 1 : 0;
    const dim3 grid = xo ? dim3((unsigned)(heads8 * nqb)) : dim3((a.S + 127) / 128, a.H, a.N);
    if constexpr (HD == 64) {
      if (g_a2v_attn_fwd32) {                               // the 32 x 32 x 16 form (round 5)
        if (a.mdist) hipLaunchKernelGGL((attn_fwd32_bf16_kernel<DROP, false, true>), grid, dim3(256), A32_LDS, st, a);
        // (row sums on the matrix cores, a2v_debug_set(18, 2): four more 32 x 32 x 16 MFMAs per tile compute 32 identical rows; the
        // kernel runs at the power cap, where their energy costs more clock than the 32 v_add they replace: 593 against 578 us)
        else if (!DROP && !a.lse && g_a2v_attn_msum == 2) hipLaunchKernelGGL((attn_fwd32_bf16_kernel<DROP, !DROP>), grid, dim3(256), A32_LDS, st, a);
        else hipLaunchKernelGGL((attn_fwd32_bf16_kernel<DROP, false>), grid, dim3(256), A32_LDS, st, a);
        return;
      }
    }
    if (a.mdist) hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, DROP, 2, false, true>), grid, dim3(256), 0, st, a);
    else if (!DROP && !a.lse && g_a2v_attn_msum) hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, DROP, 2, !DROP>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, DROP, 2, false>), grid, dim3(256), 0, st, a);
  } else if (a.mdist) {
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, DROP, 1, false, true>), dim3((a.S + 63) / 64, a.H, a.N), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, DROP, 1>), dim3((a.S + 63) / 64, a.H, a.N), dim3(256), 0, st, a);
  }
}

// =============================================================================================
//                                            backward
// =============================================================================================
// delta[n,h,s] = sum_d dO[n,s,h,d] * O[n,s,h,d]
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ d_o,
                                                         float* __restrict__ delta, int N, int S, int H) {
  const long long total = (long long)N * S * H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H);
    const long long ns = i / H;
    const T* po = o + ns * ((long long)H * HD) + (long long)h * HD;
    const T* pd = d_o + ns * ((long long)H * HD) + (long long)h * HD;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 x = Vec4<T>::load(po + d), y = Vec4<T>::load(pd + d);
      acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    const int n = (int)(ns / S), s = (int)(ns - (long long)n * S);
    delta[((long long)n * H + h) * S + s] = acc;
  }
}

// d loss / d alibi_scale_h: every workgroup of a backward launch contributes to one of only H addresses -- as plain
// atomics that is thousands of same-address read-modify-writes in a row, which the L2 serialises (60 us of a 160 us
// launch at N*H = 1536 workgroups; a "last workgroup folds" counter has the same problem).  Each workgroup instead
// stores its partial sum to its own slot of a device scratch and a one-block-per-head kernel folds them afterwards:
// no contention, and the sum order is fixed (deterministic gradient).
#define A2V_DAL_PART_MAX (A2V_WS_DAL_BYTES / 4)        // the partial sums live at A2V_WS_DAL of the stream's workspace
__device__ __forceinline__ void dalibi_finish(float dal_wave, float* dalibi, int h, float* use_part, float slope, float asc) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  if (!use_part) {                                      // too many workgroups for the scratch: direct atomics
    if ((tid & 63) == 0 && dalibi && asc >= 0.f) atomicAdd(dalibi + h, dal_wave * slope);
    return;
  }
  if ((tid & 63) == 0) red[tid >> 6] = dal_wave;
  __syncthreads();
  if (tid == 0) {
    float sum = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += red[w];
    const float v = asc >= 0.f ? sum * slope : 0.f;
    use_part[((long long)h * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x] = v;      // blockIdx.y = head
  }
}
__global__ __launch_bounds__(256) void dalibi_fold_kernel(float* __restrict__ dalibi, int per_head, const float* __restrict__ g_dal_part) {
  __shared__ float red[4];
  const int h = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < per_head; i += 256) s += g_dal_part[(long long)h * per_head + i];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) dalibi[h] += red[0] + red[1] + red[2] + red[3];
}


// NT = 0: one workgroup per 64-key block, dQ summed across key blocks with fp32 atomics into dq32 (any S).
// NT > 0: S <= 64*NT; one workgroup per (n, h) walks every key block, dQ of all NT query tiles stays in
//         registers and is stored once into the q slot of dqkv: no atomics, no dq32 zero-fill, no cast pass
//         (the student's 140-token sequences: 324 us -> see DESIGN.md).
template <typename T, int HD, int NT>
__global__ __launch_bounds__(256, NT > 0 ? 2 : 1) void attn_bwd_kernel(AttnBwdArgs a) {
  constexpr bool BF = std::is_same<T, bf16>::value;
  constexpr int ND = HD / 16;
  constexpr int TILE_BYTES = BF ? 64 * HD * 2 : 64 * FP32_PITCH(HD) * 4;
  constexpr int DS_BYTES = BF ? 64 * 64 * 2 : 64 * 68 * 4;
  // dynamic LDS carve-up (all offsets multiples of 16 B)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sQ = smem;                                   // [q][d] row image
  char* const sDO = sQ + TILE_BYTES;                       // [q][d] row image of dO
  char* const sKt = sDO + TILE_BYTES;                      // K tile for dQ (tr image for bf16)
  char* const sDS = sKt + TILE_BYTES;                      // dS [q][kv]
  float* const sLse = (float*)(sDS + DS_BYTES);
  float* const sDelta = sLse + 64;
  float* const sPosQ = sDelta + 64;
  float* const sMd = sPosQ + 64;
  char* const sQt = (char*)(sMd + 64);                     // bf16 only: tr image of Q
  char* const sDOt = sQt + TILE_BYTES;                     // bf16 only: tr image of dO

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const long long orow = (long long)H * HD;
  const T* qkv = (const T*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const T* dO = (const T*)a.d_out + (long long)n * S * orow + (long long)h * HD;
  const float slope = a.slopes[h];
  const float asc = a.alibi_scale[h];
  const float coef = -slope * fmaxf(asc, 0.f);

  static_assert(NT == 0 || NT == 3, "whole-sequence mode keeps exactly three dQ tiles");
  // dQ of the three query tiles: separately named so that the (rolled) query loop selects them with v_cndmask
  // instead of dynamic indexing (scratch) or a 3x unroll (370 VGPRs)
  f32x4 dq0[ND], dq1[ND], dq2[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) { dq0[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dq1[i] = dq0[i]; dq2[i] = dq0[i]; }
  float dal = 0.f;
  const int ntile = (S + 63) / 64;
  const int nkb = NT > 0 ? ntile : 1;
  for (int kb = 0; kb < nkb; ++kb) {
  const int kv0 = (NT > 0 ? kb : (int)blockIdx.x) * 64;
  if (NT > 0 && kb > 0) __syncthreads();   // the previous key block's dQ product has finished with sKt / sDS
  const int kv = kv0 + wid * 16 + fr;            // this lane's key column
  const int kvc = kv < S ? kv : S - 1;
  const bool kv_ok = kv < S;
  const float pos_kv = a.pos ? (float)a.pos[(long long)n * S + kvc] : (float)kvc;

  // K / V fragments as B operands (k = d, col = kv): lane (kv=fr, g)
  bf16x8 kf_b[BF ? HD / 32 : 1], vf_b[BF ? HD / 32 : 1];
  float kf_f[BF ? 1 : HD / 4], vf_f[BF ? 1 : HD / 4];
  if constexpr (BF) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      kf_b[ks] = *(const bf16x8*)(qkv + (long long)kvc * rowstride + (long long)H * HD + ks * 32 + g * 8);
      vf_b[ks] = *(const bf16x8*)(qkv + (long long)kvc * rowstride + 2LL * H * HD + ks * 32 + g * 8);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < HD / 4; ++ks) {
      kf_f[ks] = qkv[(long long)kvc * rowstride + (long long)H * HD + ks * 4 + g];
      vf_f[ks] = qkv[(long long)kvc * rowstride + 2LL * H * HD + ks * 4 + g];
    }
  }
  // K tile of the whole block for the dQ product
  if constexpr (BF) {
    constexpr int NCH = HD / 8;
    for (int c = tid; c < 64 * NCH; c += 256) {
      const int r = c / NCH, ch = c % NCH;
      uint4 kz = {0u, 0u, 0u, 0u};
      if (kv0 + r < S) kz = *(const uint4*)(qkv + (long long)(kv0 + r) * rowstride + (long long)H * HD + ch * 8);
      *(uint4*)(sKt + trimg_off<HD>(r, ch * 8)) = kz;
    }
  } else {
    constexpr int NCH = HD / 4;
    for (int c = tid; c < 64 * NCH; c += 256) {
      const int r = c / NCH, ch = c % NCH;
      uint4 kz = {0u, 0u, 0u, 0u};
      if (kv0 + r < S) kz = *(const uint4*)(qkv + (long long)(kv0 + r) * rowstride + (long long)H * HD + ch * 4);
      *(uint4*)((float*)sKt + r * FP32_PITCH(HD) + ch * 4) = kz;
    }
  }

  f32x4 dk[ND], dv[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
  for (int qt = 0; qt < ntile; ++qt) {
    const int q0 = qt * 64;
    __syncthreads();
    // ---- stage Q and dO tiles (+ lse, delta, pos)
    if constexpr (BF) {
      constexpr int NCH = HD / 8;
      for (int c = tid; c < 64 * NCH; c += 256) {
        const int r = c / NCH, ch = c % NCH;
        uint4 qz = {0u, 0u, 0u, 0u}, dz = {0u, 0u, 0u, 0u};
        if (q0 + r < S) {
          qz = *(const uint4*)(qkv + (long long)(q0 + r) * rowstride + ch * 8);
          dz = *(const uint4*)(dO + (long long)(q0 + r) * orow + ch * 8);
        }
        *(uint4*)(sQ + rowimg_off<HD>(r, ch)) = qz;
        *(uint4*)(sQt + trimg_off<HD>(r, ch * 8)) = qz;
        *(uint4*)(sDO + rowimg_off<HD>(r, ch)) = dz;
        *(uint4*)(sDOt + trimg_off<HD>(r, ch * 8)) = dz;
      }
    } else {
      constexpr int NCH = HD / 4;
      for (int c = tid; c < 64 * NCH; c += 256) {
        const int r = c / NCH, ch = c % NCH;
        uint4 qz = {0u, 0u, 0u, 0u}, dz = {0u, 0u, 0u, 0u};
        if (q0 + r < S) {
          qz = *(const uint4*)(qkv + (long long)(q0 + r) * rowstride + ch * 4);
          dz = *(const uint4*)(dO + (long long)(q0 + r) * orow + ch * 4);
        }
        *(uint4*)((float*)sQ + r * FP32_PITCH(HD) + ch * 4) = qz;
        *(uint4*)((float*)sDO + r * FP32_PITCH(HD) + ch * 4) = dz;
      }
    }
    if (tid < 64) {
      const int qq = q0 + tid;
      const bool ok = qq < S;
      const long long si = ((long long)n * H + h) * S + (ok ? qq : 0);
      sLse[tid] = ok ? a.lse[si] : INFINITY;     // exp(x - inf) = 0 -> padded query rows contribute nothing
      sDelta[tid] = ok ? a.delta[si] : 0.f;
      sPosQ[tid] = ok ? (a.pos ? (float)a.pos[(long long)n * S + qq] : (float)qq) : 0.f;
      sMd[tid] = (ok && a.mdist) ? a.mdist[si] : 0.f;
    }
    __syncthreads();

    // ---- S, P, dP, dS for the 4 query sub-blocks; rows q = 16i+4g+r, col kv = fr
    f32x4 p[4], ds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BF) {
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          const bf16x8 qa = *(const bf16x8*)(sQ + rowimg_off<HD>(16 * i + fr, ks * 4 + g));
          const bf16x8 da = *(const bf16x8*)(sDO + rowimg_off<HD>(16 * i + fr, ks * 4 + g));
          sacc = A2V_MFMA_16x16x32(qa, kf_b[ks], sacc, 0, 0, 0);
          pacc = A2V_MFMA_16x16x32(da, vf_b[ks], pacc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < HD / 4; ++ks) {
          const float qa = ((const float*)sQ)[(16 * i + fr) * FP32_PITCH(HD) + ks * 4 + g];
          const float da = ((const float*)sDO)[(16 * i + fr) * FP32_PITCH(HD) + ks * 4 + g];
          sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, kf_f[ks], sacc, 0, 0, 0);
          pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(da, vf_f[ks], pacc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * i + 4 * g + r;
        const float dist = fabsf(sPosQ[ql] - pos_kv);
        const float sc = sacc[r] * a.scale + coef * dist;
        const float pv = kv_ok ? __expf(sc - sLse[ql]) : 0.f;
        float pdrop = pv, dpe = pacc[r];
        if (a.drop.thresh) {                       // same mask as the forward: P_drop = P*m/(1-p), dP = (dO V^T)*m/(1-p)
          const int qg = q0 + ql < S ? q0 + ql : S - 1;
          const bool keep = drop_keep2(a.drop, drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qg)), (unsigned)kvc);
          pdrop = keep ? pv * a.drop.scale : 0.f;
          dpe = keep ? pacc[r] * a.drop.scale : 0.f;
        }
        const float dsv = pv * (dpe - sDelta[ql]);
        p[i][r] = pdrop;
        ds[i][r] = dsv;
        dal -= dsv * (dist - sMd[ql]);
      }
    }
    // ---- dS -> LDS (for dQ), as [q][kv]
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = 16 * i + 4 * g + r;
          const int col = wid * 16 + fr;
          *(bf16*)(sDS + ql * 128 + ((((col >> 3) ^ ((ql >> 1) & 7)) & 7) << 4) + ((col & 7) << 1)) = (bf16)ds[i][r];
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) ((float*)sDS)[(16 * i + 4 * g + r) * 68 + wid * 16 + fr] = ds[i][r];
    }

    // ---- dV^T += dO^T P ; dK^T += Q^T dS      (k = q, permuted order shared by both operands)
    if constexpr (BF) {
      const int qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf, df;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[r] = (bf16)p[2 * st][r]; pf[4 + r] = (bf16)p[2 * st + 1][r];
          df[r] = (bf16)ds[2 * st][r]; df[4 + r] = (bf16)ds[2 * st + 1][r];
        }
        const int r_lo = 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const bf16x8 dof = tr_frag(sDOt + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sDOt + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
          const bf16x8 qf = tr_frag(sQt + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sQt + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
          dv[i] = A2V_MFMA_16x16x32(dof, pf, dv[i], 0, 0, 0);
          dk[i] = A2V_MFMA_16x16x32(qf, df, dk[i], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = 16 * j + 4 * g + r;
#pragma unroll
          for (int i = 0; i < ND; ++i) {
            const float dof = ((const float*)sDO)[ql * FP32_PITCH(HD) + 16 * i + fr];
            const float qf = ((const float*)sQ)[ql * FP32_PITCH(HD) + 16 * i + fr];
            dv[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(dof, p[j][r], dv[i], 0, 0, 0);
            dk[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf, ds[j][r], dk[i], 0, 0, 0);
          }
        }
    }
    __syncthreads();   // dS tile complete

    // ---- dQ[q rows 16*wid..][d] += dS K * scale : A = dS rows, B = K (k = kv, col = d)
    f32x4 dq[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (NT > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[i][r] = qt == 0 ? dq0[i][r] : (qt == 1 ? dq1[i][r] : dq2[i][r]);
      }
    }
    if constexpr (BF) {
      const int qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int rq = wid * 16 + fr;
        const int ch = ks * 4 + g;
        const bf16x8 af = *(const bf16x8*)(sDS + rq * 128 + (((ch ^ ((rq >> 1) & 7)) & 7) << 4));
        const int r_lo = 32 * ks + 8 * g + qq, r_hi = r_lo + 4;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const bf16x8 kf = tr_frag(sKt + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sKt + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
          dq[i] = A2V_MFMA_16x16x32(af, kf, dq[i], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float af = ((const float*)sDS)[(wid * 16 + fr) * 68 + ks * 4 + g];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const float kf = ((const float*)sKt)[(ks * 4 + g) * FP32_PITCH(HD) + 16 * i + fr];
          dq[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, kf, dq[i], 0, 0, 0);
        }
      }
    }
    // D: row q = q0 + 16*wid + 4g + r, col d = 16i + fr
    if constexpr (NT > 0) {
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dq0[i][r] = qt == 0 ? dq[i][r] : dq0[i][r];
          dq1[i][r] = qt == 1 ? dq[i][r] : dq1[i][r];
          dq2[i][r] = qt == 2 ? dq[i][r] : dq2[i][r];
        }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq2 = q0 + wid * 16 + 4 * g + r;
        if (qq2 < S) {
          float* dst = a.dq32 + (((long long)n * S + qq2) * H + h) * HD;
#pragma unroll
          for (int i = 0; i < ND; ++i) atomicAdd(dst + 16 * i + fr, dq[i][r] * a.scale);
        }
      }
    }
  }
  // ---- write dK, dV: D rows d = 16i+4g+r, col kv = fr
  if (kv_ok) {
    T* dkv = (T*)a.dqkv + ((long long)n * S + kv) * rowstride + (long long)h * HD;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      f32x4 kk = dk[i] * a.scale;
      Vec4<T>::store(dkv + (long long)H * HD + 16 * i + 4 * g, kk);
      Vec4<T>::store(dkv + 2LL * H * HD + 16 * i + 4 * g, dv[i]);
    }
  }
  }   // key blocks
  if constexpr (NT > 0) {   // dQ straight into the q slot of dqkv
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq2 = t * 64 + wid * 16 + 4 * g + r;
        if (qq2 < S) {
          T* dst = (T*)a.dqkv + ((long long)n * S + qq2) * rowstride + (long long)h * HD;
#pragma unroll
          for (int i = 0; i < ND; ++i) dst[16 * i + fr] = (T)((t == 0 ? dq0[i][r] : (t == 1 ? dq1[i][r] : dq2[i][r])) * a.scale);
        }
      }
  }
  // ---- d alibi_scale_h = slope_h * [scale_h >= 0] * sum dS * (-dist)
  dal = wave_sum(dal);
  dalibi_finish(dal, a.dalibi, h, a.dal_part, slope, asc);
}

// =============================================================================================
template <typename T, int HD>
static int attn_fwd_launch(const AttnArgs& a, hipStream_t st) {
  dim3 grid((a.S + 63) / 64, a.H, a.N);
  hipLaunchKernelGGL((attn_fwd_kernel<T, HD>), grid, dim3(256), 0, st, a);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// Words (u32) per (n, h) of the dropout bit mask a2v_attention_fwd writes and a2v_attention_bwd reads for this (dtype, S, HD), or 0
// where the kernels regenerate the decisions from the counter hash instead (the pointer is then ignored).
extern "C" int a2v_attention_dropmask_words(int dtype, int S, int HD) {
  return (dtype == A2V_BF16 && HD == 64 && as2_takes(S)) ? AS2_ROWS * 4 * 2 : 0;
}

extern "C" int a2v_attention_fwd(int dtype, const void* qkv, void* out, float* lse, float* mdist, unsigned* dropmask, const float* qk_norms,
                                 const int* pos, const float* slopes, const float* alibi_scale, int N, int S, int H, int HD, float scale,
                                 float drop_p, unsigned drop_seed, hipStream_t stream) {
  if (N <= 0 || S <= 0 || H <= 0 || !slopes || !alibi_scale || drop_p < 0.f || drop_p >= 1.f) return A2V_ERR_ARG;
  AttnArgs a{qkv, out, lse, pos, slopes, alibi_scale, N, S, H, scale, make_drop(drop_p, drop_seed), 0, nullptr, mdist, 32.f,
             a2v_attention_dropmask_words(dtype, S, HD) ? dropmask : nullptr, (dtype == A2V_BF16 && HD == 64) ? qk_norms : nullptr};
  if (dtype == A2V_BF16) {
    a.band_nats = g_a2v_attn_band >= 8 ? (float)g_a2v_attn_band : A2V_BAND_NATS_BF16;
    const bool dr = a.drop.thresh != 0;
    if (HD == 64 && dr) attn_fwd_bf16_launch<64, true>(a, stream);
    else if (HD == 64) attn_fwd_bf16_launch<64, false>(a, stream);
    else if (HD == 32 && dr) attn_fwd_bf16_launch<32, true>(a, stream);
    else if (HD == 32) attn_fwd_bf16_launch<32, false>(a, stream);
    else return A2V_ERR_ARG;
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16X3) {          // fp32 storage, split products on the bf16 matrix cores
    const dim3 grid((a.S + 63) / 64, a.H, a.N);
    a.qk_bound = nullptr;
    if (HD == 64 && a.S >= 512 && g_a2v_attn_band && !a.pos && (long long)a.N * a.H * 2 * A2V_QKB_SPLIT <= A2V_QKB_MAX) {
      char* const ws = a2v_stream_ws(stream);
      if (ws) {
        float* const bp = (float*)(ws + A2V_WS_QKB);
        hipLaunchKernelGGL((attn_qk_bound_kernel<64, float>), dim3(A2V_QKB_SPLIT, a.N), dim3(256), 0, stream, (const float*)a.qkv, bp, a.S, a.H);
        a.qk_bound = bp;
      }
    }
    if (HD == 64) hipLaunchKernelGGL((attn_fwd_x3_kernel<64>), grid, dim3(256), 0, stream, a);
    else if (HD == 32) hipLaunchKernelGGL((attn_fwd_x3_kernel<32>), grid, dim3(256), 0, stream, a);
    else return A2V_ERR_UNSUPPORTED;
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_F32) {
    if (HD == 64) return attn_fwd_launch<float, 64>(a, stream);
    if (HD == 32) return attn_fwd_launch<float, 32>(a, stream);
    if (HD == 16) return attn_fwd_launch<float, 16>(a, stream);
    return A2V_ERR_ARG;
  }
  return A2V_ERR_DTYPE;
}

// ---------------------------------------------------------------------------------------------
// bf16, HD = 64, S <= 192: the whole (n, h) sequence is resident.  Q and dO (all S rows, zero padded to 192) sit in
// LDS once, each as ONE image that serves both the row reads (S, dP) and the transposed reads (dK^T, dV^T):
// 16-B chunks XOR-swizzled by (row & 7).  Per key block only the K tile for dQ is staged; inside the query loop
// there is no global load and one barrier per (key block, query tile) step (dS double buffered), where the tiled
// kernel above re-staged Q / dO for every key block and sat in s_waitcnt / s_barrier for 61 % of its cycles.
// delta = rowsum(dO * O) is computed here too (no separate pass).

#define ABS_ROWS 192
#define ABS_LDS (2 * ABS_ROWS * 128 + 64 * 128 + 2 * 64 * 128 + 4 * ABS_ROWS * 4 + 3 * 64 * 4)

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_seq_kernel(AttnBwdArgs a, const bf16* __restrict__ o_fwd) {
  constexpr int HD = 64, ND = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sQ = smem;                                   // [192][64] dual-use image
  char* const sDO = sQ + ABS_ROWS * 128;
  char* const sKt = sDO + ABS_ROWS * 128;                  // [64][64] tr image of the current key block
  char* const sDS = sKt + 64 * 128;                        // 2 x [64 q][64 kv] row image
  float* const sLse = (float*)(sDS + 2 * 64 * 128);
  float* const sDelta = sLse + ABS_ROWS;
  float* const sPosQ = sDelta + ABS_ROWS;
  float* const sMd = sPosQ + ABS_ROWS;
  float* const sBias = sMd + ABS_ROWS;                     // [3][64]: column sums of this (n, h)'s dQ / dK / dV (bias gradient)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const long long orow = (long long)H * HD;
  const bf16* qkv = (const bf16*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const bf16* dO = (const bf16*)a.d_out + (long long)n * S * orow + (long long)h * HD;
  if (a.bias_sum && tid < 192) sBias[tid] = 0.f;           // published by the barriers of the first key block
  const bf16* Of = o_fwd + (long long)n * S * orow + (long long)h * HD;
  const float slope = a.slopes[h];
  const float asc = a.alibi_scale[h];
  const float coef = -slope * fmaxf(asc, 0.f);

  // ---- one round trip for everything the first key block needs: Q / dO (all rows, zero padded), this thread's O row
  // (delta), lse / pos, and key block 0's K / V fragments and K tile; later key blocks are fetched one block ahead,
  // under the previous block's math (the kernel used to sit through three dependent global round trips before its
  // first MFMA and two more per key block, with only two workgroups per CU to hide them)
  const int ntile = (S + 63) / 64;
  bf16x8 kf_n[2], vf_n[2];
  uint4 kt_n[2];
  float pos_n;
#define ABS_FETCH_KB(kb_)                                                                                      \
  {                                                                                                             \
    const int kvn_ = (kb_) * 64 + wid * 16 + fr;                                                                \
    const int kvc_ = kvn_ < S ? kvn_ : S - 1;                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                          \
      kf_n[ks] = *(const bf16x8*)(qkv + (long long)kvc_ * rowstride + (long long)H * HD + ks * 32 + g * 8);     \
      vf_n[ks] = *(const bf16x8*)(qkv + (long long)kvc_ * rowstride + 2LL * H * HD + ks * 32 + g * 8);          \
    }                                                                                                           \
    pos_n = a.pos ? (float)a.pos[(long long)n * S + kvc_] : (float)kvc_;                                        \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                             \
      const int c = tid + 256 * u, r = c >> 3, ch = c & 7;                                                      \
      kt_n[u] = make_uint4(0u, 0u, 0u, 0u);                                                                     \
      if ((kb_) * 64 + r < S) kt_n[u] = *(const uint4*)(qkv + (long long)((kb_) * 64 + r) * rowstride + (long long)H * HD + ch * 8); \
    }                                                                                                           \
  }
  {
    uint4 qz[ABS_ROWS * 8 / 256], dz[ABS_ROWS * 8 / 256];      // all loads in flight before the first LDS write
    uint4 oz[8];
#pragma unroll
    for (int u = 0; u < ABS_ROWS * 8 / 256; ++u) {
      const int c = tid + 256 * u, r = c >> 3, ch = c & 7;
      qz[u] = make_uint4(0u, 0u, 0u, 0u);
      dz[u] = qz[u];
      if (r < S) {
        qz[u] = *(const uint4*)(qkv + (long long)r * rowstride + ch * 8);
        dz[u] = *(const uint4*)(dO + (long long)r * orow + ch * 8);
      }
    }
    const bool rok = tid < S;                       // thread tid < ABS_ROWS owns query row tid for delta / lse / pos
#pragma unroll
    for (int u = 0; u < 8; ++u) oz[u] = rok ? *(const uint4*)(Of + (long long)tid * orow + u * 8) : make_uint4(0u, 0u, 0u, 0u);
    float lse_r = INFINITY, pos_r = 0.f, md_r = 0.f; // exp(x - inf) = 0: padded queries vanish
    if (rok) {
      lse_r = a.lse[((long long)n * H + h) * S + tid];
      pos_r = a.pos ? (float)a.pos[(long long)n * S + tid] : (float)tid;
      if (a.mdist) md_r = a.mdist[((long long)n * H + h) * S + tid];
    }
    ABS_FETCH_KB(0)
#pragma unroll
    for (int u = 0; u < ABS_ROWS * 8 / 256; ++u) {
      const int c = tid + 256 * u, r = c >> 3, ch = c & 7;
      *(uint4*)(sQ + dual_off(r, ch)) = qz[u];
      *(uint4*)(sDO + dual_off(r, ch)) = dz[u];
    }
    __syncthreads();                                // dO image complete: delta reads row tid from it
    if (tid < ABS_ROWS) {
      float dl = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bf16x8 ov = *(const bf16x8*)&oz[u];
        const bf16x8 dv_ = *(const bf16x8*)(sDO + dual_off(tid, u));
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += (float)ov[i] * (float)dv_[i];
      }
      sDelta[tid] = dl;
      sLse[tid] = lse_r;
      sPosQ[tid] = pos_r;
      sMd[tid] = md_r;
    }
  }

  f32x4 dq0[ND], dq1[ND], dq2[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) { dq0[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dq1[i] = dq0[i]; dq2[i] = dq0[i]; }
  float dal = 0.f;
  const int qq = (lane & 15) >> 2, pp = lane & 3;

#pragma unroll 1
  for (int kb = 0; kb < ntile; ++kb) {
    const int kv0 = kb * 64;
    const int kv = kv0 + wid * 16 + fr;            // this lane's key column
    const int kvc = kv < S ? kv : S - 1;
    const bool kv_ok = kv < S;
    const float pos_kv = pos_n;
    bf16x8 kf_b[2], vf_b[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { kf_b[ks] = kf_n[ks]; vf_b[ks] = vf_n[ks]; }
    __syncthreads();                               // previous key block's dQ products are done with sKt
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + 256 * u, r = c >> 3, ch = c & 7;
      *(uint4*)(sKt + trimg_off<HD>(r, ch * 8)) = kt_n[u];
    }
    __syncthreads();                               // (first pass: also publishes delta / lse / pos)
    if (kb + 1 < ntile) ABS_FETCH_KB(kb + 1)        // next key block's operands arrive under this block's math

    f32x4 dk[ND], dv[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int qt = 0; qt < ntile; ++qt) {
      const int q0 = qt * 64;
      char* const dsb = sDS + (qt & 1) * (64 * 128);
      // ---- S, P, dP, dS for the 4 query sub-blocks; rows q = q0 + 16i + 4g + r, col kv = fr
      // only 16-row query sub-blocks below S and key columns below S do work (S = 140: the third tile holds 12 rows)
      const int ni = min(4, (S - q0 + 15) >> 4);
      const bool wave_has_keys = kv0 + wid * 16 < S;         // wave-uniform
      f32x4 p[4], ds[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ds[i] = p[i];
        if (i >= ni || !wave_has_keys) continue;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 qa = *(const bf16x8*)(sQ + dual_off(q0 + 16 * i + fr, ks * 4 + g));
          const bf16x8 da = *(const bf16x8*)(sDO + dual_off(q0 + 16 * i + fr, ks * 4 + g));
          sacc = A2V_MFMA_16x16x32(qa, kf_b[ks], sacc, 0, 0, 0);
          pacc = A2V_MFMA_16x16x32(da, vf_b[ks], pacc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = q0 + 16 * i + 4 * g + r;
          const float dist = fabsf(sPosQ[ql] - pos_kv);
          const float sc = sacc[r] * a.scale + coef * dist;
          const float pv = kv_ok ? __expf(sc - sLse[ql]) : 0.f;
          float pdrop = pv, dpe = pacc[r];
          if (DROP) {                                // same mask as the forward
            const int qg = ql < S ? ql : S - 1;
            const bool keep = drop_keep2(a.drop, drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + qg)), (unsigned)kvc);
            pdrop = keep ? pv * a.drop.scale : 0.f;
            dpe = keep ? pacc[r] * a.drop.scale : 0.f;
          }
          const float dsv = pv * (dpe - sDelta[ql]);
          p[i][r] = pdrop;
          ds[i][r] = dsv;
          dal -= dsv * (dist - sMd[ql]);
        }
      }
      // ---- dS -> LDS (for dQ), as [q][kv]
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = 16 * i + 4 * g + r;
          const int col = wid * 16 + fr;
          *(bf16*)(dsb + ql * 128 + ((((col >> 3) ^ ((ql >> 1) & 7)) & 7) << 4) + ((col & 7) << 1)) = (bf16)ds[i][r];
        }
      // ---- dV^T += dO^T P ; dK^T += Q^T dS      (k = q, permuted order shared by both operands)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        if (2 * st >= ni || !wave_has_keys) continue;
        bf16x8 pf, df;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[r] = (bf16)p[2 * st][r]; pf[4 + r] = (bf16)p[2 * st + 1][r];
          df[r] = (bf16)ds[2 * st][r]; df[4 + r] = (bf16)ds[2 * st + 1][r];
        }
        const int r_lo = q0 + 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const int chl = 2 * i + (pp >> 1), sub = (pp & 1) << 3;       // column 16 i + 4 pp: chunk, byte inside it
          const bf16x8 dof = tr_frag(sDO + dual_off(r_lo, chl) + sub, sDO + dual_off(r_hi, chl) + sub);
          const bf16x8 qf = tr_frag(sQ + dual_off(r_lo, chl) + sub, sQ + dual_off(r_hi, chl) + sub);
          dv[i] = A2V_MFMA_16x16x32(dof, pf, dv[i], 0, 0, 0);
          dk[i] = A2V_MFMA_16x16x32(qf, df, dk[i], 0, 0, 0);
        }
      }
      __syncthreads();   // dS tile complete (and the other dS buffer's readers of step - 1 are past)

      // ---- dQ[q rows q0 + 16 wid ..][d] += dS K : A = dS rows, B = K (k = kv, col = d)
      f32x4 dq[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[i][r] = qt == 0 ? dq0[i][r] : (qt == 1 ? dq1[i][r] : dq2[i][r]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (q0 + wid * 16 >= S || kv0 + 32 * ks >= S) continue;     // no valid query rows in this wave / no keys in this half
        const int rq = wid * 16 + fr;
        const int ch = ks * 4 + g;
        const bf16x8 af = *(const bf16x8*)(dsb + rq * 128 + (((ch ^ ((rq >> 1) & 7)) & 7) << 4));
        const int r_lo = 32 * ks + 8 * g + qq, r_hi = r_lo + 4;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const bf16x8 kf = tr_frag(sKt + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sKt + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
          dq[i] = A2V_MFMA_16x16x32(af, kf, dq[i], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dq0[i][r] = qt == 0 ? dq[i][r] : dq0[i][r];
          dq1[i][r] = qt == 1 ? dq[i][r] : dq1[i][r];
          dq2[i][r] = qt == 2 ? dq[i][r] : dq2[i][r];
        }
    }
    // ---- dK, dV of this key block: D rows d = 16i+4g+r, col kv = fr
    if (kv_ok) {
      bf16* dkv = (bf16*)a.dqkv + ((long long)n * S + kv) * rowstride + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        f32x4 kk = dk[i] * a.scale;
        Vec4<bf16>::store(dkv + (long long)H * HD + 16 * i + 4 * g, kk);
        Vec4<bf16>::store(dkv + 2LL * H * HD + 16 * i + 4 * g, dv[i]);
      }
    }
    if (a.bias_sum) {          // column sums over this block's keys of what was stored (bf16-rounded), 16 key lanes -> 1
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sk = kv_ok ? (float)(bf16)(dk[i][r] * a.scale) : 0.f;
          float sv_ = kv_ok ? (float)(bf16)dv[i][r] : 0.f;
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { sk += __shfl_xor(sk, o, 64); sv_ += __shfl_xor(sv_, o, 64); }
          if (fr == 0) {
            atomicAdd(&sBias[64 + 16 * i + 4 * g + r], sk);
            atomicAdd(&sBias[128 + 16 * i + 4 * g + r], sv_);
          }
        }
    }
  }
  // ---- dQ into the q slot of dqkv: rows q = 64 t + 16 wid + 4g + r, col d = 16i + fr
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq2 = t * 64 + wid * 16 + 4 * g + r;
      if (qq2 < S) {
        bf16* dst = (bf16*)a.dqkv + ((long long)n * S + qq2) * rowstride + (long long)h * HD;
#pragma unroll
        for (int i = 0; i < ND; ++i) dst[16 * i + fr] = (bf16)((t == 0 ? dq0[i][r] : (t == 1 ? dq1[i][r] : dq2[i][r])) * a.scale);
      }
    }
  if (a.bias_sum) {            // dQ column sums: this lane's rows (t, r), then the 4 row groups g, then the 4 waves through LDS
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      float sq = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq2 = t * 64 + wid * 16 + 4 * g + r;
          const float v = (t == 0 ? dq0[i][r] : (t == 1 ? dq1[i][r] : dq2[i][r])) * a.scale;
          sq += qq2 < S ? (float)(bf16)v : 0.f;
        }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (g == 0) atomicAdd(&sBias[16 * i + fr], sq);
    }
    __syncthreads();
    if (tid < 192) atomicAdd(a.bias_sum + (long long)(tid >> 6) * H * HD + (long long)h * HD + (tid & 63), sBias[tid]);
  }
  dal = wave_sum(dal);
  dalibi_finish(dal, a.dalibi, h, a.dal_part, slope, asc);
}

// ---------------------------------------------------------------------------------------------
// bf16, HD = 64, S <= 144 (the student's 140 kept frames): the same whole-sequence backward on 48-row tiles and THREE
// waves.  With 64-key blocks and four waves S = 140 is 2.19 blocks: in the third block one wave had keys and three sat
// at the two barriers of each of its three query steps (SQ_WAIT_ANY 50 % of the wave cycles, MFMA busy 0.06).  Here a
// key block is 48 keys (16 per wave; 140 = 2.92 blocks) and there is NO query-tile loop with barriers inside a key block:
//   phase A  every wave walks ALL query rows (pairs of 16-row sub-blocks) against its 16 keys: S, P, dP, dS; dV^T / dK^T
//            accumulate in registers; dS of all rows goes to LDS ([q][key], zero beyond key 48)
//   barrier
//   phase B  dQ: wave w owns the 16-row query groups w, w+3, w+6 (static accumulators, no select on a loop index) and
//            multiplies their dS rows with the key block's K (transposed reads of the K tile)
//   barrier
// = 2 barriers per key block (6 per (n, h)) instead of 2 per (key block, query tile) step (18).  Per-row operands (lse in
// log2 units, delta, position, dropout row term) are formed once into LDS and read as one 16-B vector per four rows.
// The next key block's K / V fragments and K tile are fetched by loads NOTHING else touches until the block change (no
// zero-initialised destinations, no guards: rows past S re-read row S-1 and are masked by kv_ok / zero dS columns).
#define AB48_QROWS 160            // Q / dO images: ten 16-row groups (an odd last sub-block's partner is a zero group)
#define AB48_ROWS 160             // S <= 160: ten query groups (wave w owns w, w + 3, w + 6, w + 9), up to four 48-key blocks
#define AB48_LDS (2 * AB48_QROWS * 128 + 64 * 128 + AB48_ROWS * 128 + 5 * AB48_ROWS * 4 + 3 * 64 * 4)

// NW waves: a key block is 16 NW keys (48 or 64).  The key blocks of a sequence run one after the other, so what counts is their
// number: S = 152 (what the sampler draws at 768 rows) is 4 blocks of 48 but 3 of 64 -- the launcher picks NW = 4 where that saves a
// block (S <= 64, 97 .. 128, 145 .. 160) and NW = 3 otherwise.
template <bool DROP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_seq48_kernel(AttnBwdArgs a, const bf16* __restrict__ o_fwd) {
  constexpr int HD = 64, ND = 4, NT = 64 * NW, KB = 16 * NW, NTQ = (AB48_ROWS / 16 + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sQ = smem;                                   // [160][64] dual-use image
  char* const sDO = sQ + AB48_QROWS * 128;
  char* const sKt = sDO + AB48_QROWS * 128;                // [64][64] tr image of the key block's K (rows 48.. stay zero)
  char* const sDS = sKt + 64 * 128;                        // [144 q][64 kv] row image (cols 48.. stay zero)
  float* const sLse = (float*)(sDS + AB48_ROWS * 128);     // lse * log2(e); +inf on padded rows
  float* const sDelta = sLse + AB48_ROWS;
  float* const sPosQ = sDelta + AB48_ROWS;
  unsigned* const sRowT = (unsigned*)(sPosQ + AB48_ROWS);  // dropout row terms
  float* const sMd = (float*)(sRowT + AB48_ROWS);          // mean attended distance of the row (AttnBwdArgs::mdist)
  float* const sBias = sMd + AB48_ROWS;                    // [3][64]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const long long orow = (long long)H * HD;
  const bf16* qkv = (const bf16*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const bf16* dO = (const bf16*)a.d_out + (long long)n * S * orow + (long long)h * HD;
  const bf16* Of = o_fwd + (long long)n * S * orow + (long long)h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float slope = a.slopes[h];
  const float asc = a.alibi_scale[h];
  const float coef2 = -slope * fmaxf(asc, 0.f) * LOG2E;
  const float scale2 = a.scale * LOG2E;
  const int nsub = (S + 15) >> 4;                  // 16-row query groups (<= 9)
  const int npair = (nsub + 1) >> 1;
  const int nkb = (S + KB - 1) / KB;

  bf16x8 kf_n[2], vf_n[2], kt_n[2];
  int praw = 0;
#define AB48_FETCH(kb_)                                                                                         \
  {                                                                                                             \
    const int kvc_ = min((kb_) * KB + wid * 16 + fr, S - 1);                                                    \
    const bf16* krow_ = qkv + (long long)kvc_ * rowstride + (long long)H * HD + g * 8;                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                          \
      kf_n[ks] = *(const bf16x8*)(krow_ + ks * 32);                                                             \
      vf_n[ks] = *(const bf16x8*)(krow_ + (long long)H * HD + ks * 32);                                         \
    }                                                                                                           \
    if (a.pos) praw = a.pos[(long long)n * S + kvc_];                                                           \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                             \
      const int c = tid + NT * u, r = c >> 3, ch = c & 7;                                                       \
      kt_n[u] = *(const bf16x8*)(qkv + (long long)min((kb_) * KB + r, S - 1) * rowstride + (long long)H * HD + ch * 8); \
    }                                                                                                           \
  }
  {
    constexpr int NQ = (AB48_QROWS * 8 + NT - 1) / NT;     // 7 (5) chunks of Q and of dO per thread
    uint4 qz[NQ], dz[NQ], oz[8];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int c = tid + NT * u, r = c >> 3, ch = c & 7;
      qz[u] = make_uint4(0u, 0u, 0u, 0u);
      dz[u] = qz[u];
      if (r < S) {
        qz[u] = *(const uint4*)(qkv + (long long)r * rowstride + ch * 8);
        dz[u] = *(const uint4*)(dO + (long long)r * orow + ch * 8);
      }
    }
    const bool rok = tid < S;                       // thread tid < 144 owns query row tid for delta / lse / pos / row term
#pragma unroll
    for (int u = 0; u < 8; ++u) oz[u] = rok ? *(const uint4*)(Of + (long long)tid * orow + u * 8) : make_uint4(0u, 0u, 0u, 0u);
    float lse_r = INFINITY, pos_r = 0.f, md_r = 0.f; // exp2(x - inf) = 0: padded queries vanish
    if (rok) {
      lse_r = a.lse[((long long)n * H + h) * S + tid] * LOG2E;
      pos_r = a.pos ? (float)a.pos[(long long)n * S + tid] : (float)tid;
      if (a.mdist) md_r = a.mdist[((long long)n * H + h) * S + tid];
    }
    AB48_FETCH(0)
    if (a.bias_sum && tid < 192) sBias[tid] = 0.f;  // 3 x 64 sums
    if (NW == 3 && tid < 128) *(uint4*)(sKt + 48 * 128 + tid * 16) = make_uint4(0u, 0u, 0u, 0u);         // key rows 48..63: zero for good
#pragma unroll
    for (int u = 0; u < (AB48_ROWS * 8 + NT - 1) / NT; ++u)
      if (tid + NT * u < AB48_ROWS * 8) *(uint4*)(sDS + (tid + NT * u) * 16) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int c = tid + NT * u, r = c >> 3, ch = c & 7;
      if (c < AB48_QROWS * 8) {
        *(uint4*)(sQ + dual_off(r, ch)) = qz[u];
        *(uint4*)(sDO + dual_off(r, ch)) = dz[u];
      }
    }
    __syncthreads();                                // dO image complete: delta reads row tid from it
    if (tid < AB48_ROWS) {
      float dl = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bf16x8 ov = *(const bf16x8*)&oz[u];
        const bf16x8 dv_ = *(const bf16x8*)(sDO + dual_off(tid, u));
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += (float)ov[i] * (float)dv_[i];
      }
      sDelta[tid] = dl;
      sLse[tid] = lse_r;
      sPosQ[tid] = pos_r;
      sMd[tid] = md_r;
      sRowT[tid] = drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + min(tid, S - 1)));
    }
  }

  f32x4 dq[NTQ][ND];
#pragma unroll
  for (int t = 0; t < NTQ; ++t)
#pragma unroll
    for (int i = 0; i < ND; ++i) dq[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dal = 0.f;
  int ds_off[4];                                   // dS image: byte offset of (row 4g + r, column 16 wid + fr); + 128 * first row
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ql = 4 * g + r, col = wid * 16 + fr;
    ds_off[r] = ql * 128 + ((((col >> 3) ^ ((ql >> 1) & 7)) & 7) << 4) + ((col & 7) << 1);
  }

#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int kv = kb * KB + wid * 16 + fr;        // this lane's key column
    const bool kv_ok = kv < S;
    const int kvc = kv_ok ? kv : S - 1;
    const float pos_kv = a.pos ? (float)praw : (float)kvc;
    bf16x8 kf_b[2], vf_b[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { kf_b[ks] = kf_n[ks]; vf_b[ks] = vf_n[ks]; }
    // (the barrier that ended the previous key block's phase B -- or the prologue's -- protects sKt and sDS)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + NT * u, r = c >> 3, ch = c & 7;
      *(bf16x8*)(sKt + trimg_off<HD>(r, ch * 8)) = kt_n[u];
    }
    if (kb + 1 < nkb) AB48_FETCH(kb + 1)           // next key block's operands arrive under this block's math
    if (kb == 0) __syncthreads();                  // publishes delta / lse / pos / row terms (first block only)

    f32x4 dk[ND], dv[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // ---------------- phase A: all query rows against this wave's 16 keys, two 16-row groups per trip (one basic block:
    // the second group's LDS reads and MFMAs interleave with the first group's VALU); an odd last group runs alone
    auto rows16 = [&](int rb, f32x4& p_o, f32x4& ds_o) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 qa = *(const bf16x8*)(sQ + dual_off(rb + fr, ks * 4 + g));
        const bf16x8 da = *(const bf16x8*)(sDO + dual_off(rb + fr, ks * 4 + g));
        sacc = A2V_MFMA_16x16x32(qa, kf_b[ks], sacc, 0, 0, 0);
        pacc = A2V_MFMA_16x16x32(da, vf_b[ks], pacc, 0, 0, 0);
      }
      const f32x4 lse4 = *(const f32x4*)(sLse + rb + 4 * g);
      const f32x4 del4 = *(const f32x4*)(sDelta + rb + 4 * g);
      const f32x4 pq4 = *(const f32x4*)(sPosQ + rb + 4 * g);
      const f32x4 md4 = *(const f32x4*)(sMd + rb + 4 * g);
      const uint4 rt4 = *(const uint4*)(sRowT + rb + 4 * g);
      const unsigned rts[4] = {rt4.x, rt4.y, rt4.z, rt4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dist = __builtin_fabsf(pq4[r] - pos_kv);
        const float e2 = __builtin_fmaf(sacc[r], scale2, __builtin_fmaf(dist, coef2, -lse4[r]));
        const float pv = kv_ok ? __builtin_amdgcn_exp2f(e2) : 0.f;
        float mlt = 1.f;                             // dropout as ONE multiplier (0 or 1/(1-p)), same mask as the forward
        if (DROP) mlt = drop_keep2(a.drop, rts[r], (unsigned)kvc) ? a.drop.scale : 0.f;
        const float dsv = pv * __builtin_fmaf(pacc[r], mlt, -del4[r]);
        p_o[r] = pv * mlt;
        ds_o[r] = dsv;
        dal = __builtin_fmaf(-dsv, dist - md4[r], dal);
        // dS -> LDS as [q][kv] (for dQ): row rb + 4g + r, column 16 wid + fr (per-lane offsets formed once, ds_off[])
        *(bf16*)(sDS + rb * 128 + ds_off[r]) = (bf16)dsv;
      }
    };
    auto pair_products = [&](int pr, const f32x4& p0, const f32x4& p1, const f32x4& d0, const f32x4& d1) {
      // dV^T += dO^T P ; dK^T += Q^T dS      (k = q, permuted order shared by both operands)
      bf16x8 pf, df;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (bf16)p0[r]; pf[4 + r] = (bf16)p1[r];
        df[r] = (bf16)d0[r]; df[4 + r] = (bf16)d1[r];
      }
      const int r_lo = 32 * pr + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int chl = 2 * i + (pp >> 1), sub = (pp & 1) << 3;       // column 16 i + 4 pp: chunk, byte inside it
        const bf16x8 dof = tr_frag(sDO + dual_off(r_lo, chl) + sub, sDO + dual_off(r_hi, chl) + sub);
        const bf16x8 qf = tr_frag(sQ + dual_off(r_lo, chl) + sub, sQ + dual_off(r_hi, chl) + sub);
        dv[i] = A2V_MFMA_16x16x32(dof, pf, dv[i], 0, 0, 0);
        dk[i] = A2V_MFMA_16x16x32(qf, df, dk[i], 0, 0, 0);
      }
    };
#pragma unroll 1
    for (int pr = 0; pr < (nsub >> 1); ++pr) {
      f32x4 p0, p1, d0, d1;
      rows16(32 * pr, p0, d0);
      rows16(32 * pr + 16, p1, d1);
      pair_products(pr, p0, p1, d0, d1);
    }
    if (nsub & 1) {
      f32x4 p0, d0;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rows16(16 * (nsub - 1), p0, d0);
      pair_products(nsub >> 1, p0, z, d0, z);
    }
    // dK, dV of this key block: D rows d = 16i+4g+r, col kv = fr
    if (kv_ok) {
      bf16* dkv = (bf16*)a.dqkv + ((long long)n * S + kv) * rowstride + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        f32x4 kk = dk[i] * a.scale;
        Vec4<bf16>::store(dkv + (long long)H * HD + 16 * i + 4 * g, kk);
        Vec4<bf16>::store(dkv + 2LL * H * HD + 16 * i + 4 * g, dv[i]);
      }
    }
    if (a.bias_sum) {          // column sums over this block's keys of what was stored (bf16-rounded), 16 key lanes -> 1
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sk = kv_ok ? (float)(bf16)(dk[i][r] * a.scale) : 0.f;
          float sv_ = kv_ok ? (float)(bf16)dv[i][r] : 0.f;
          sk = row16_sum(sk);                        // DPP adds inside the 16-lane row (no LDS round trips)
          sv_ = row16_sum(sv_);
          if (fr == 0) {
            atomicAdd(&sBias[64 + 16 * i + 4 * g + r], sk);
            atomicAdd(&sBias[128 + 16 * i + 4 * g + r], sv_);
          }
        }
    }
    __syncthreads();           // dS of every row and the K tile are in LDS

    // ---------------- phase B: dQ[q rows of groups wid, wid+3, wid+6][d] += dS K : A = dS rows, B = K (k = kv, col = d)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (kb * KB + 32 * ks >= S) continue;        // no keys in this half (uniform)
      const int r_lo = 32 * ks + 8 * g + qq, r_hi = r_lo + 4;
      bf16x8 kfr[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i)
        kfr[i] = tr_frag(sKt + trimg_off<HD>(r_lo, 16 * i + 4 * pp), sKt + trimg_off<HD>(r_hi, 16 * i + 4 * pp));
#pragma unroll
      for (int t = 0; t < NTQ; ++t) {
        const int qg = wid + NW * t;
        if (qg >= nsub) continue;                  // uniform
        const int rq = 16 * qg + fr;
        const int ch = ks * 4 + g;
        const bf16x8 af = *(const bf16x8*)(sDS + rq * 128 + (((ch ^ ((rq >> 1) & 7)) & 7) << 4));
#pragma unroll
        for (int i = 0; i < ND; ++i) dq[t][i] = A2V_MFMA_16x16x32(af, kfr[i], dq[t][i], 0, 0, 0);
      }
    }
    __syncthreads();           // the next key block overwrites sKt and sDS
  }
#undef AB48_FETCH
  // ---- dQ into the q slot of dqkv: rows q = 16 (wid + 3t) + 4g + r, col d = 16i + fr
#pragma unroll
  for (int t = 0; t < NTQ; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * (wid + NW * t) + 4 * g + r;
      if (qr < S) {
        bf16* dst = (bf16*)a.dqkv + ((long long)n * S + qr) * rowstride + (long long)h * HD;
#pragma unroll
        for (int i = 0; i < ND; ++i) dst[16 * i + fr] = (bf16)(dq[t][i][r] * a.scale);
      }
    }
  if (a.bias_sum) {            // dQ column sums: this lane's rows (t, r), then the 4 row groups g, then the waves through LDS
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      float sq = 0.f;
#pragma unroll
      for (int t = 0; t < NTQ; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qr = 16 * (wid + NW * t) + 4 * g + r;
          sq += qr < S ? (float)(bf16)(dq[t][i][r] * a.scale) : 0.f;
        }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (g == 0) atomicAdd(&sBias[16 * i + fr], sq);
    }
    __syncthreads();
    if (tid < 192) atomicAdd(a.bias_sum + (long long)(tid >> 6) * H * HD + (long long)h * HD + (tid & 63), sBias[tid]);
  }
  dal = wave_sum(dal);
  dalibi_finish(dal, a.dalibi, h, a.dal_part, slope, asc);
}

// ---------------------------------------------------------------------------------------------
// bf16x3 (dtype A2V_BF16X3), HD = 64, S <= 160: the whole-sequence backward above for fp32 q / k / v / dO / O / dqkv with every one
// of its five products as three exact bf16 products over split operands (x = hi + lo: X_lo Y_hi + X_hi Y_lo + X_hi Y_hi, fp32
// accumulation) -- fp32-grade gradients at a third of the bf16 matrix rate instead of the 1/16 of v_mfma_f32_16x16x4_f32, on which
// the parity mode's student attention backward ran (6 % of its step).  Q and dO live in LDS as hi and lo dual-use images, the key
// block's K tile and the dS rows likewise; K / V fragments, P and dS are split in registers.  Softmax, ALiBi, dropout, delta and
// the d alibi_scale sum are the fp32 arithmetic of the kernels above (delta from the fp32 O and dO in global memory).
#define AX3_LDS (4 * AB48_QROWS * 128 + 2 * 64 * 128 + 2 * AB48_ROWS * 128 + 5 * AB48_ROWS * 4 + 3 * 64 * 4)

template <bool DROP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_x3_kernel(AttnBwdArgs a, const float* __restrict__ o_fwd) {
  constexpr int HD = 64, ND = 4, NT = 64 * NW, KB = 16 * NW, NTQ = (AB48_ROWS / 16 + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sQh = smem;                                  // [160][64] dual-use images, hi / lo
  char* const sQl = sQh + AB48_QROWS * 128;
  char* const sDOh = sQl + AB48_QROWS * 128;
  char* const sDOl = sDOh + AB48_QROWS * 128;
  char* const sKth = sDOl + AB48_QROWS * 128;              // [64][64] tr images of the key block's K (rows 16 NW.. stay zero)
  char* const sKtl = sKth + 64 * 128;
  char* const sDSh = sKtl + 64 * 128;                      // [160 q][64 kv] row images (cols 16 NW.. stay zero)
  char* const sDSl = sDSh + AB48_ROWS * 128;
  float* const sLse = (float*)(sDSl + AB48_ROWS * 128);    // lse * log2(e); +inf on padded rows
  float* const sDelta = sLse + AB48_ROWS;
  float* const sPosQ = sDelta + AB48_ROWS;
  unsigned* const sRowT = (unsigned*)(sPosQ + AB48_ROWS);  // dropout row terms
  float* const sMd = (float*)(sRowT + AB48_ROWS);
  float* const sBias = sMd + AB48_ROWS;                    // [3][64]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const long long orow = (long long)H * HD;
  const float* qkv = (const float*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const float* dO = (const float*)a.d_out + (long long)n * S * orow + (long long)h * HD;
  const float* Of = o_fwd + (long long)n * S * orow + (long long)h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float slope = a.slopes[h];
  const float asc = a.alibi_scale[h];
  const float coef2 = -slope * fmaxf(asc, 0.f) * LOG2E;
  const float scale2 = a.scale * LOG2E;
  const int nsub = (S + 15) >> 4;
  const int nkb = (S + KB - 1) / KB;

  float kraw[2][8], vraw[2][8], ktraw[2][8];               // the next key block's operands, fp32 as loaded
  int praw = 0;
#define AX3_LD8(dst_, src_) { const f32x4 a_ = *(const f32x4*)(src_), b_ = *(const f32x4*)((src_) + 4); \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { (dst_)[e_] = a_[e_]; (dst_)[4 + e_] = b_[e_]; } }
#define AX3_FETCH(kb_)                                                                                          \
  {                                                                                                             \
    const int kvc_ = min((kb_) * KB + wid * 16 + fr, S - 1);                                                    \
    const float* krow_ = qkv + (long long)kvc_ * rowstride + (long long)H * HD + g * 8;                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                          \
      AX3_LD8(kraw[ks], krow_ + ks * 32)                                                                        \
      AX3_LD8(vraw[ks], krow_ + (long long)H * HD + ks * 32)                                                    \
    }                                                                                                           \
    if (a.pos) praw = a.pos[(long long)n * S + kvc_];                                                           \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                             \
      const int c = tid + NT * u, r = c >> 3, ch = c & 7;                                                       \
      AX3_LD8(ktraw[u], qkv + (long long)min((kb_) * KB + min(r, KB - 1), S - 1) * rowstride + (long long)H * HD + ch * 8) \
    }                                                                                                           \
  }
  {
    constexpr int NQ = (AB48_QROWS * 8 + NT - 1) / NT;     // chunks (8 elements) of Q and of dO per thread
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int c = tid + NT * u, r = c >> 3, ch = c & 7;
      if (c < AB48_QROWS * 8) {
        float qv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dv_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r < S) {
          AX3_LD8(qv, qkv + (long long)r * rowstride + ch * 8)
          AX3_LD8(dv_, dO + (long long)r * orow + ch * 8)
        }
        bf16x8 hh, ll;
        split8(qv, hh, ll);
        *(bf16x8*)(sQh + dual_off(r, ch)) = hh;
        *(bf16x8*)(sQl + dual_off(r, ch)) = ll;
        split8(dv_, hh, ll);
        *(bf16x8*)(sDOh + dual_off(r, ch)) = hh;
        *(bf16x8*)(sDOl + dual_off(r, ch)) = ll;
      }
    }
    AX3_FETCH(0)
    if (a.bias_sum && tid < 192) sBias[tid] = 0.f;
    // key rows / dS columns beyond the block's 16 NW keys: zero for good
    if (NW == 3 && tid < 128) {
      *(uint4*)(sKth + 48 * 128 + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
      *(uint4*)(sKtl + 48 * 128 + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < (AB48_ROWS * 8 + NT - 1) / NT; ++u)
      if (tid + NT * u < AB48_ROWS * 8) {
        *(uint4*)(sDSh + (tid + NT * u) * 16) = make_uint4(0u, 0u, 0u, 0u);
        *(uint4*)(sDSl + (tid + NT * u) * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
    if (tid < AB48_ROWS) {                           // per-row operands; delta from the fp32 O and dO
      const bool rok = tid < S;
      float dl = 0.f, lse_r = INFINITY, pos_r = 0.f, md_r = 0.f;
      if (rok) {
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
          const f32x4 ov = *(const f32x4*)(Of + (long long)tid * orow + 4 * u);
          const f32x4 dv4 = *(const f32x4*)(dO + (long long)tid * orow + 4 * u);
          dl += ov[0] * dv4[0] + ov[1] * dv4[1] + ov[2] * dv4[2] + ov[3] * dv4[3];
        }
        lse_r = a.lse[((long long)n * H + h) * S + tid] * LOG2E;
        pos_r = a.pos ? (float)a.pos[(long long)n * S + tid] : (float)tid;
        if (a.mdist) md_r = a.mdist[((long long)n * H + h) * S + tid];
      }
      sDelta[tid] = dl;
      sLse[tid] = lse_r;
      sPosQ[tid] = pos_r;
      sMd[tid] = md_r;
      sRowT[tid] = drop_row_term(a.drop.seed, (unsigned)(((long long)n * H + h) * S + min(tid, S - 1)));
    }
  }

  f32x4 dq[NTQ][ND];
#pragma unroll
  for (int t = 0; t < NTQ; ++t)
#pragma unroll
    for (int i = 0; i < ND; ++i) dq[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dal = 0.f;
  int ds_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ql = 4 * g + r, col = wid * 16 + fr;
    ds_off[r] = ql * 128 + ((((col >> 3) ^ ((ql >> 1) & 7)) & 7) << 4) + ((col & 7) << 1);
  }

#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int kv = kb * KB + wid * 16 + fr;
    const bool kv_ok = kv < S;
    const int kvc = kv_ok ? kv : S - 1;
    const float pos_kv = a.pos ? (float)praw : (float)kvc;
    bf16x8 kfh[2], kfl[2], vfh[2], vfl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { split8(kraw[ks], kfh[ks], kfl[ks]); split8(vraw[ks], vfh[ks], vfl[ks]); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + NT * u, r = c >> 3, ch = c & 7;
      if (r < KB) {
        bf16x8 hh, ll;
        split8(ktraw[u], hh, ll);
        *(bf16x8*)(sKth + trimg_off<HD>(r, ch * 8)) = hh;
        *(bf16x8*)(sKtl + trimg_off<HD>(r, ch * 8)) = ll;
      }
    }
    if (kb + 1 < nkb) AX3_FETCH(kb + 1)
    if (kb == 0) __syncthreads();                  // Q / dO images and the per-row operands

    f32x4 dk[ND], dv[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    auto rows16 = [&](int rb, f32x4& p_o, f32x4& ds_o) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o_ = dual_off(rb + fr, ks * 4 + g);
        const bf16x8 qh_ = *(const bf16x8*)(sQh + o_), ql_ = *(const bf16x8*)(sQl + o_);
        const bf16x8 dh_ = *(const bf16x8*)(sDOh + o_), dl_ = *(const bf16x8*)(sDOl + o_);
        sacc = A2V_MFMA_16x16x32(ql_, kfh[ks], sacc, 0, 0, 0);
        sacc = A2V_MFMA_16x16x32(qh_, kfl[ks], sacc, 0, 0, 0);
        sacc = A2V_MFMA_16x16x32(qh_, kfh[ks], sacc, 0, 0, 0);
        pacc = A2V_MFMA_16x16x32(dl_, vfh[ks], pacc, 0, 0, 0);
        pacc = A2V_MFMA_16x16x32(dh_, vfl[ks], pacc, 0, 0, 0);
        pacc = A2V_MFMA_16x16x32(dh_, vfh[ks], pacc, 0, 0, 0);
      }
      const f32x4 lse4 = *(const f32x4*)(sLse + rb + 4 * g);
      const f32x4 del4 = *(const f32x4*)(sDelta + rb + 4 * g);
      const f32x4 pq4 = *(const f32x4*)(sPosQ + rb + 4 * g);
      const f32x4 md4 = *(const f32x4*)(sMd + rb + 4 * g);
      const uint4 rt4 = *(const uint4*)(sRowT + rb + 4 * g);
      const unsigned rts[4] = {rt4.x, rt4.y, rt4.z, rt4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dist = __builtin_fabsf(pq4[r] - pos_kv);
        const float e2 = __builtin_fmaf(sacc[r], scale2, __builtin_fmaf(dist, coef2, -lse4[r]));
        const float pv = kv_ok ? __builtin_amdgcn_exp2f(e2) : 0.f;
        float mlt = 1.f;
        if (DROP) mlt = drop_keep2(a.drop, rts[r], (unsigned)kvc) ? a.drop.scale : 0.f;
        const float dsv = pv * __builtin_fmaf(pacc[r], mlt, -del4[r]);
        p_o[r] = pv * mlt;
        ds_o[r] = dsv;
        dal = __builtin_fmaf(-dsv, dist - md4[r], dal);
        const bf16 dh1 = (bf16)dsv;
        *(bf16*)(sDSh + rb * 128 + ds_off[r]) = dh1;
        *(bf16*)(sDSl + rb * 128 + ds_off[r]) = (bf16)(dsv - (float)dh1);
      }
    };
    auto pair_products = [&](int pr, const f32x4& p0, const f32x4& p1, const f32x4& d0, const f32x4& d1) {
      float pv8[8], dv8[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { pv8[r] = p0[r]; pv8[4 + r] = p1[r]; dv8[r] = d0[r]; dv8[4 + r] = d1[r]; }
      bf16x8 pfh, pfl, dfh, dfl;
      split8(pv8, pfh, pfl);
      split8(dv8, dfh, dfl);
      const int r_lo = 32 * pr + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int chl = 2 * i + (pp >> 1), sub = (pp & 1) << 3;
        const int o_lo = dual_off(r_lo, chl) + sub, o_hi = dual_off(r_hi, chl) + sub;
        const bf16x8 doh = tr_frag(sDOh + o_lo, sDOh + o_hi), dol = tr_frag(sDOl + o_lo, sDOl + o_hi);
        const bf16x8 qfh = tr_frag(sQh + o_lo, sQh + o_hi), qfl = tr_frag(sQl + o_lo, sQl + o_hi);
        dv[i] = A2V_MFMA_16x16x32(dol, pfh, dv[i], 0, 0, 0);
        dv[i] = A2V_MFMA_16x16x32(doh, pfl, dv[i], 0, 0, 0);
        dv[i] = A2V_MFMA_16x16x32(doh, pfh, dv[i], 0, 0, 0);
        dk[i] = A2V_MFMA_16x16x32(qfl, dfh, dk[i], 0, 0, 0);
        dk[i] = A2V_MFMA_16x16x32(qfh, dfl, dk[i], 0, 0, 0);
        dk[i] = A2V_MFMA_16x16x32(qfh, dfh, dk[i], 0, 0, 0);
      }
    };
#pragma unroll 1
    for (int pr = 0; pr < (nsub >> 1); ++pr) {
      f32x4 p0, p1, d0, d1;
      rows16(32 * pr, p0, d0);
      rows16(32 * pr + 16, p1, d1);
      pair_products(pr, p0, p1, d0, d1);
    }
    if (nsub & 1) {
      f32x4 p0, d0;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rows16(16 * (nsub - 1), p0, d0);
      pair_products(nsub >> 1, p0, z, d0, z);
    }
    if (kv_ok) {
      float* dkv = (float*)a.dqkv + ((long long)n * S + kv) * rowstride + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        *(f32x4*)(dkv + (long long)H * HD + 16 * i + 4 * g) = dk[i] * a.scale;
        *(f32x4*)(dkv + 2LL * H * HD + 16 * i + 4 * g) = dv[i];
      }
    }
    if (a.bias_sum) {
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sk = kv_ok ? dk[i][r] * a.scale : 0.f;
          float sv_ = kv_ok ? dv[i][r] : 0.f;
          sk = row16_sum(sk);
          sv_ = row16_sum(sv_);
          if (fr == 0) {
            atomicAdd(&sBias[64 + 16 * i + 4 * g + r], sk);
            atomicAdd(&sBias[128 + 16 * i + 4 * g + r], sv_);
          }
        }
    }
    __syncthreads();           // dS of every row and the K tile are in LDS

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (kb * KB + 32 * ks >= S) continue;
      const int r_lo = 32 * ks + 8 * g + qq, r_hi = r_lo + 4;
      bf16x8 kh_[ND], kl_[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int o_lo = trimg_off<HD>(r_lo, 16 * i + 4 * pp), o_hi = trimg_off<HD>(r_hi, 16 * i + 4 * pp);
        kh_[i] = tr_frag(sKth + o_lo, sKth + o_hi);
        kl_[i] = tr_frag(sKtl + o_lo, sKtl + o_hi);
      }
#pragma unroll
      for (int t = 0; t < NTQ; ++t) {
        const int qg = wid + NW * t;
        if (qg >= nsub) continue;
        const int rq = 16 * qg + fr;
        const int ch = ks * 4 + g;
        const int o_ = rq * 128 + (((ch ^ ((rq >> 1) & 7)) & 7) << 4);
        const bf16x8 ah_ = *(const bf16x8*)(sDSh + o_), al_ = *(const bf16x8*)(sDSl + o_);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          dq[t][i] = A2V_MFMA_16x16x32(al_, kh_[i], dq[t][i], 0, 0, 0);
          dq[t][i] = A2V_MFMA_16x16x32(ah_, kl_[i], dq[t][i], 0, 0, 0);
          dq[t][i] = A2V_MFMA_16x16x32(ah_, kh_[i], dq[t][i], 0, 0, 0);
        }
      }
    }
    __syncthreads();           // the next key block overwrites the K tile and dS
  }
#undef AX3_FETCH
#undef AX3_LD8
#pragma unroll
  for (int t = 0; t < NTQ; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * (wid + NW * t) + 4 * g + r;
      if (qr < S) {
        float* dst = (float*)a.dqkv + ((long long)n * S + qr) * rowstride + (long long)h * HD;
#pragma unroll
        for (int i = 0; i < ND; ++i) dst[16 * i + fr] = dq[t][i][r] * a.scale;
      }
    }
  if (a.bias_sum) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      float sq = 0.f;
#pragma unroll
      for (int t = 0; t < NTQ; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qr = 16 * (wid + NW * t) + 4 * g + r;
          sq += qr < S ? dq[t][i][r] * a.scale : 0.f;
        }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (g == 0) atomicAdd(&sBias[16 * i + fr], sq);
    }
    __syncthreads();
    if (tid < 192) atomicAdd(a.bias_sum + (long long)(tid >> 6) * H * HD + (long long)h * HD + (tid & 63), sBias[tid]);
  }
  dal = wave_sum(dal);
  dalibi_finish(dal, a.dalibi, h, a.dal_part, slope, asc);
}

// ---------------------------------------------------------------------------------------------
// bf16, HD = 64, long sequences (finetuning attends over all T = 2000 frames): two passes, no atomics, no fp32 dQ
// scratch.  MODE 0 (dK, dV): a workgroup owns 128 keys (32 per wave, as B-operand fragments in registers) and
// streams the queries; MODE 1 (dQ): it owns 128 queries and streams the keys.  The streamed side comes through LDS
// in 64-row tiles, double buffered (global -> registers during the tile's math -> LDS, one barrier per tile), each
// tile ONE dual-use image (row reads for S / dP, transposed reads for the gradient products).  Each A fragment read
// from LDS feeds both 16-row owner groups of the wave, which halves LDS traffic per MFMA against one group per wave
// (the limit of the tiled kernel above: 8 KB of fragment reads per 8 MFMAs).  S and P are recomputed in both passes
// (7 matmuls instead of 5) -- cheaper than 2 GB of fp32 atomics per layer at S = 2000.
//   MODE 0: S[q][kv] = X1(Q) . own1(K), dP = X2(dO) . own2(V); dV^T += tr(X2) . P, dK^T += tr(X1) . dS
//   MODE 1: S^T[kv][q] = X1(K) . own1(Q), dP^T = X2(V) . own2(dO); dQ^T += tr(X1) . dS^T
#define ABL_LDS (2 * 16384 + 2 * 4 * 64 * 4)

template <int MODE, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_long_kernel(AttnBwdArgs a) {
  constexpr int HD = 64, ND = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const sRowBase = (float*)(smem + 2 * 16384);     // [buf][neg | delta | pos | mean distance][64] of the streamed rows

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const int h = blockIdx.y, n = blockIdx.z;
  const int S = a.S, H = a.H;
  const long long rowstride = 3LL * H * HD;
  const long long orow = (long long)H * HD;
  const bf16* qkv = (const bf16*)a.qkv + (long long)n * S * rowstride + (long long)h * HD;
  const bf16* dO = (const bf16*)a.d_out + (long long)n * S * orow + (long long)h * HD;
  const float slope = a.slopes[h];
  const float asc = a.alibi_scale[h];
  const float coef = -slope * fmaxf(asc, 0.f);
  const long long nh = (long long)n * H + h;

  // streamed operands and owner operands
  const bf16* x1p = MODE == 0 ? qkv : qkv + (long long)H * HD;
  const bf16* x2p = MODE == 0 ? dO : qkv + 2LL * H * HD;
  const long long x2s = MODE == 0 ? orow : rowstride;
  const bf16* o1p = MODE == 0 ? qkv + (long long)H * HD : qkv;
  const bf16* o2p = MODE == 0 ? qkv + 2LL * H * HD : dO;
  const long long o2s = MODE == 0 ? rowstride : orow;

  // ---- owner side: 2 groups of 16 rows per wave
  bf16x8 own1[2][2], own2[2][2];
  float cneg[2], cdel[2], cpos[2];
  unsigned own_term[2];
  int own_sh[2];
  int orow_i[2];
#pragma unroll
  for (int og = 0; og < 2; ++og) {
    const int o = blockIdx.x * 128 + wid * 32 + og * 16 + fr;
    const int oc = o < S ? o : S - 1;
    orow_i[og] = o;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      own1[og][ks] = *(const bf16x8*)(o1p + (long long)oc * rowstride + ks * 32 + g * 8);
      own2[og][ks] = *(const bf16x8*)(o2p + (long long)oc * o2s + ks * 32 + g * 8);
    }
    cpos[og] = a.pos ? (float)a.pos[(long long)n * S + oc] : (float)oc;
    own_term[og] = MODE == 0 ? drop_col_term((unsigned)oc) : drop_row_term(a.drop.seed, (unsigned)(nh * S) + (unsigned)oc);
    own_sh[og] = (oc & 1) ? 16 : 0;
    if (MODE == 0) {
      cneg[og] = o < S ? 0.f : -INFINITY;
      cdel[og] = 0.f;
    } else {
      cneg[og] = o < S ? -a.lse[nh * S + oc] : -INFINITY;
      cdel[og] = a.delta[nh * S + oc];
    }
  }

  f32x4 accA[2][ND], accB[2][ND];
#pragma unroll
  for (int og = 0; og < 2; ++og)
#pragma unroll
    for (int i = 0; i < ND; ++i) { accA[og][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[og][i] = accA[og][i]; }
  float dal = 0.f;

  const int ntile = (S + 63) / 64;
  uint4 r1[2], r2[2];
  float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;

#define ABL_FETCH(t_)                                                                               \
  {                                                                                                 \
    const int s0_ = (t_) * 64;                                                                      \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                 \
      const int c = tid + 256 * u, r = c >> 3, ch = c & 7;                                          \
      r1[u] = make_uint4(0u, 0u, 0u, 0u);                                                           \
      r2[u] = r1[u];                                                                                \
      if (s0_ + r < S) {                                                                            \
        r1[u] = *(const uint4*)(x1p + (long long)(s0_ + r) * rowstride + ch * 8);                   \
        r2[u] = *(const uint4*)(x2p + (long long)(s0_ + r) * x2s + ch * 8);                         \
      }                                                                                             \
    }                                                                                               \
    if (tid < 64) {                                                                                 \
      const int row = s0_ + tid;                                                                    \
      const bool ok = row < S;                                                                      \
      const int rc = ok ? row : S - 1;                                                              \
      rs2 = a.pos ? (float)a.pos[(long long)n * S + rc] : (float)rc;                                \
      if (MODE == 0) {                                                                              \
        rs0 = ok ? -a.lse[nh * S + rc] : -INFINITY;                                                 \
        rs1 = a.delta[nh * S + rc];                                                                 \
        rs3 = a.mdist ? a.mdist[nh * S + rc] : 0.f;                                                 \
      } else {                                                                                      \
        rs0 = ok ? 0.f : -INFINITY;                                                                 \
        rs1 = 0.f;                                                                                  \
      }                                                                                             \
    }                                                                                               \
  }
#define ABL_COMMIT(buf_)                                                                            \
  {                                                                                                 \
    char* const b1 = smem + (buf_) * 16384;                                                         \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                 \
      const int c = tid + 256 * u, r = c >> 3, ch = c & 7;                                          \
      *(uint4*)(b1 + dual_off(r, ch)) = r1[u];                                                      \
      *(uint4*)(b1 + 8192 + dual_off(r, ch)) = r2[u];                                               \
    }                                                                                               \
    if (tid < 64) {                                                                                 \
      float* const sr = sRowBase + (buf_) * 256;                                                    \
      sr[tid] = rs0; sr[64 + tid] = rs1; sr[128 + tid] = rs2; sr[192 + tid] = rs3;                  \
    }                                                                                               \
  }

  ABL_FETCH(0)
  ABL_COMMIT(0)
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    const int s0 = t * 64;
    if (t + 1 < ntile) ABL_FETCH(t + 1)
    const char* const X1 = smem + cur * 16384;
    const char* const X2 = X1 + 8192;
    const float* const sNeg = sRowBase + cur * 256;
    const float* const sDel = sNeg + 64;
    const float* const sPos = sNeg + 128;
    const float* const sMdr = sNeg + 192;

    // two halves of the streamed tile (32 rows each), rolled: scores + softmax gradient of the half, then its
    // gradient products -- only one half's P / dS fragments are live, and the rolled loop keeps the scheduler from
    // hoisting the second half's fragment reads over the first half's math (register spills otherwise)
#pragma unroll 1
    for (int st = 0; st < 2; ++st) {
      bf16x8 pk[2], dk_[2];              // P and dS as B operands, per owner group
      f32x4 pprev[2], dprev[2];
#pragma unroll
      for (int ih = 0; ih < 2; ++ih) {
        const int rb = 32 * st + 16 * ih;                 // first streamed row of this 16-row sub-block
        bf16x8 a1[2], a2[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          a1[ks] = *(const bf16x8*)(X1 + dual_off(rb + fr, ks * 4 + g));
          a2[ks] = *(const bf16x8*)(X2 + dual_off(rb + fr, ks * 4 + g));
        }
        float rneg[4], rdel[4], rpos[4], rmd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = rb + 4 * g + r;
          rneg[r] = sNeg[rl];
          rdel[r] = MODE == 0 ? sDel[rl] : 0.f;
          rpos[r] = sPos[rl];
          rmd[r] = MODE == 0 ? sMdr[rl] : 0.f;
        }
        // dropout hash terms of the streamed rows (rows past S never matter: their P is 0)
        unsigned rterm[4] = {0u, 0u, 0u, 0u}, hp[2][2] = {{0u, 0u}, {0u, 0u}};
        if (DROP) {
          if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rterm[r] = drop_row_term(a.drop.seed, (unsigned)(nh * S) + (unsigned)(s0 + rb + 4 * g + r));
          } else {                                   // streamed rows are keys: 4 g + {0,1} and {2,3} are hash pairs
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const unsigned ct = drop_col_term((unsigned)(s0 + rb + 4 * g + 2 * pr));
#pragma unroll
              for (int og = 0; og < 2; ++og) hp[og][pr] = drop_finalize(own_term[og] + ct);
            }
          }
        }
#pragma unroll
        for (int og = 0; og < 2; ++og) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            sacc = A2V_MFMA_16x16x32(a1[ks], own1[og][ks], sacc, 0, 0, 0);
            pacc = A2V_MFMA_16x16x32(a2[ks], own2[og][ks], pacc, 0, 0, 0);
          }
          f32x4 pcur, dcur;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dist = fabsf(rpos[r] - cpos[og]);
            const float sc = sacc[r] * a.scale + coef * dist;
            const float pv = __expf(sc + rneg[r] + cneg[og]);
            float pdrop = pv, dpe = pacc[r];
            if (DROP) {                                // same mask as the forward (2-D pair hash, a2v_common.h)
              unsigned hh;
              if (MODE == 0) hh = (drop_finalize(rterm[r] + own_term[og]) >> own_sh[og]) & 0xFFFFu;
              else hh = (r & 1) ? (hp[og][r >> 1] >> 16) : (hp[og][r >> 1] & 0xFFFFu);
              const bool keep = hh >= (a.drop.thresh >> 16);
              pdrop = keep ? pv * a.drop.scale : 0.f;
              dpe = keep ? pacc[r] * a.drop.scale : 0.f;
            }
            const float dsv = pv * (dpe - rdel[r] - cdel[og]);
            pcur[r] = pdrop;
            dcur[r] = dsv;
            if (MODE == 0) dal -= dsv * (dist - rmd[r]);
          }
          if (ih) {
            bf16x8 pf, df;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pf[r] = (bf16)pprev[og][r]; pf[4 + r] = (bf16)pcur[r];
              df[r] = (bf16)dprev[og][r]; df[4 + r] = (bf16)dcur[r];
            }
            pk[og] = pf;
            dk_[og] = df;
          } else {
            pprev[og] = pcur;
            dprev[og] = dcur;
          }
        }
      }
      // ---- gradient products of this half: k = streamed row (permuted order shared by both operands)
      const int r_lo = 32 * st + 4 * g + qq, r_hi = r_lo + 16;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int chl = 2 * i + (pp >> 1), sub = (pp & 1) << 3;       // column 16 i + 4 pp: chunk, byte inside it
        const bf16x8 x1f = tr_frag(X1 + dual_off(r_lo, chl) + sub, X1 + dual_off(r_hi, chl) + sub);
#pragma unroll
        for (int og = 0; og < 2; ++og)
          accB[og][i] = A2V_MFMA_16x16x32(x1f, dk_[og], accB[og][i], 0, 0, 0);
        if (MODE == 0) {
          const bf16x8 x2f = tr_frag(X2 + dual_off(r_lo, chl) + sub, X2 + dual_off(r_hi, chl) + sub);
#pragma unroll
          for (int og = 0; og < 2; ++og)
            accA[og][i] = A2V_MFMA_16x16x32(x2f, pk[og], accA[og][i], 0, 0, 0);
        }
      }
    }
    if (t + 1 < ntile) ABL_COMMIT(cur ^ 1)
    __syncthreads();
  }
#undef ABL_FETCH
#undef ABL_COMMIT

  // ---- results: D rows d = 16 i + 4 g + r, column = owner row
#pragma unroll
  for (int og = 0; og < 2; ++og) {
    const int o = orow_i[og];
    if (o < S) {
      bf16* dst = (bf16*)a.dqkv + ((long long)n * S + o) * rowstride + (long long)h * HD;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const f32x4 kk = accB[og][i] * a.scale;
        if (MODE == 0) {
          Vec4<bf16>::store(dst + (long long)H * HD + 16 * i + 4 * g, kk);
          Vec4<bf16>::store(dst + 2LL * H * HD + 16 * i + 4 * g, accA[og][i]);
        } else {
          Vec4<bf16>::store(dst + 16 * i + 4 * g, kk);
        }
      }
    }
  }
  if (MODE == 0) {
    dal = wave_sum(dal);
    dalibi_finish(dal, a.dalibi, h, a.dal_part, slope, asc);
  }
}

#define A2V_ATTN_SEQ_TILES 3      // sequences up to 192 tokens take the atomic-free whole-sequence backward (bf16)

int g_a2v_attn_seq48 = 1;    // a2v_debug_set(15, v): S <= 144 takes the 48-row / three-wave whole-sequence backward (0: the 64-row one)
int g_a2v_attn_long = 1;     // a2v_debug_set(8, v): two-pass long-sequence backward on / off (off: tiled kernel + dQ atomics)

template <typename T, int HD>
static int attn_bwd_launch(const AttnBwdArgs& a_in, const void* out, float* delta, hipStream_t st) {
  AttnBwdArgs a = a_in;
  // d alibi_scale: per-workgroup partials + fold kernel whenever the launch's workgroups fit the scratch
  char* const ws = a.dalibi ? a2v_stream_ws(st) : nullptr;     // (no workspace left for this stream: direct atomics)
  auto set_part = [&](const dim3& grid) {
    a.dal_part = (a.dalibi && ws && (long long)grid.x * grid.y * grid.z <= A2V_DAL_PART_MAX) ? (float*)(ws + A2V_WS_DAL) : nullptr;
  };
  auto fold = [&](const dim3& grid) {
    if (a.dal_part) hipLaunchKernelGGL(dalibi_fold_kernel, dim3(a.H), dim3(256), 0, st, a.dalibi, (int)(grid.x * grid.z), a.dal_part);
  };
  const long long total = (long long)a.N * a.S * a.H;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  constexpr bool BF = std::is_same<T, bf16>::value;
  const bool seq64 = BF && HD == 64 && a.S <= ABS_ROWS && g_a2v_attn_long != 2;       // attn_bwd_seq_kernel forms delta itself
  if (!seq64)
    hipLaunchKernelGGL((attn_delta_kernel<T, HD>), dim3(g), dim3(256), 0, st, (const T*)out, (const T*)a.d_out, delta, a.N, a.S, a.H);
  constexpr int TILE = BF ? 64 * HD * 2 : 64 * FP32_PITCH(HD) * 4;
  constexpr int LDS = (BF ? 5 : 3) * TILE + (BF ? 64 * 64 * 2 : 64 * 68 * 4) + 4 * 64 * 4;
  if constexpr (BF && HD == 64) {
    if ((a.S > ABS_ROWS && g_a2v_attn_long) || g_a2v_attn_long == 2) {       // long sequences: dK/dV pass + dQ pass, no atomics (2: every S, experiments)
      const bool dr = a.drop.thresh != 0;
      const dim3 grid((a.S + 127) / 128, a.H, a.N);
      set_part(grid);
      if (dr) {
        hipLaunchKernelGGL((attn_bwd_long_kernel<0, true>), grid, dim3(256), ABL_LDS, st, a);
        hipLaunchKernelGGL((attn_bwd_long_kernel<1, true>), grid, dim3(256), ABL_LDS, st, a);
      } else {
        hipLaunchKernelGGL((attn_bwd_long_kernel<0, false>), grid, dim3(256), ABL_LDS, st, a);
        hipLaunchKernelGGL((attn_bwd_long_kernel<1, false>), grid, dim3(256), ABL_LDS, st, a);
      }
      fold(grid);
      A2V_CHECK_LAUNCH();
      return A2V_OK;
    }
    if (as2_takes(a.S)) {                                   // round 6: persistent ten-wave kernel, two register-resident roles (a2v_attn_seq2.h)
      const bool dr = a.drop.thresh != 0;
      const dim3 pgrid(1, a.H, a.N);                        // (the d alibi_scale partials: one slot per (n, h), as the one-shot kernels)
      set_part(pgrid);
      const long long np = (long long)a.N * a.H;
      const dim3 grid((unsigned)(np < 256 ? np : 256));     // persistent: one workgroup per CU walks the (n, h) sequences
#define AS2B_GO(DM_)                                                                                                                  \
  {                                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)attn_seq2_bwd_kernel<DM_>, hipFuncAttributeMaxDynamicSharedMemorySize, AS2B_LDS);          \
    hipLaunchKernelGGL((attn_seq2_bwd_kernel<DM_>), grid, dim3(AS2B_NT), AS2B_LDS, st, a, (const bf16*)out, a.dropmask);             \
  }
      if (!dr) AS2B_GO(0) else if (a.dropmask) AS2B_GO(2) else AS2B_GO(1)
#undef AS2B_GO
      fold(pgrid);
      A2V_CHECK_LAUNCH();
      return A2V_OK;
    }
    if (a.S <= AB48_ROWS && g_a2v_attn_seq48) {
      const bool dr = a.drop.thresh != 0;
      (void)hipFuncSetAttribute((const void*)attn_bwd_seq48_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, AB48_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_seq48_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, AB48_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_seq48_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, AB48_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_seq48_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, AB48_LDS);
      const dim3 grid(1, a.H, a.N);
      set_part(grid);
      const bool four = (a.S + 63) / 64 < (a.S + 47) / 48;            // four waves where 64-key blocks save a block
      if (four) {
        if (dr) hipLaunchKernelGGL((attn_bwd_seq48_kernel<true, 4>), grid, dim3(256), AB48_LDS, st, a, (const bf16*)out);
        else hipLaunchKernelGGL((attn_bwd_seq48_kernel<false, 4>), grid, dim3(256), AB48_LDS, st, a, (const bf16*)out);
      } else {
        if (dr) hipLaunchKernelGGL((attn_bwd_seq48_kernel<true, 3>), grid, dim3(192), AB48_LDS, st, a, (const bf16*)out);
        else hipLaunchKernelGGL((attn_bwd_seq48_kernel<false, 3>), grid, dim3(192), AB48_LDS, st, a, (const bf16*)out);
      }
      fold(grid);
      A2V_CHECK_LAUNCH();
      return A2V_OK;
    }
    if (a.S <= ABS_ROWS) {
      const bool dr = a.drop.thresh != 0;
      (void)hipFuncSetAttribute((const void*)attn_bwd_seq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ABS_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_seq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ABS_LDS);
      const dim3 grid(1, a.H, a.N);
      set_part(grid);
      if (dr) hipLaunchKernelGGL((attn_bwd_seq_kernel<true>), grid, dim3(256), ABS_LDS, st, a, (const bf16*)out);
      else hipLaunchKernelGGL((attn_bwd_seq_kernel<false>), grid, dim3(256), ABS_LDS, st, a, (const bf16*)out);
      fold(grid);
      A2V_CHECK_LAUNCH();
      return A2V_OK;
    }
  }
  if (BF && a.S <= 64 * A2V_ATTN_SEQ_TILES) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T, HD, BF ? A2V_ATTN_SEQ_TILES : 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const dim3 grid(1, a.H, a.N);
    set_part(grid);
    hipLaunchKernelGGL((attn_bwd_kernel<T, HD, BF ? A2V_ATTN_SEQ_TILES : 0>), grid, dim3(256), LDS, st, a);
    fold(grid);
  } else {
    if (!a.dq32) return A2V_ERR_ARG;
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T, HD, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const dim3 grid((a.S + 63) / 64, a.H, a.N);
    set_part(grid);
    hipLaunchKernelGGL((attn_bwd_kernel<T, HD, 0>), grid, dim3(256), LDS, st, a);
    fold(grid);
  }
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// 1 when a2v_attention_bwd writes dQ directly into the q slot of dqkv for this (dtype, S) (dq32 unused, may be
// NULL); 0 when dQ is left in dq32 (fp32, zeroed by the caller) and must be cast into dqkv afterwards.
extern "C" int a2v_attention_bwd_direct_dq(int dtype, int S, int HD) {
  if (dtype == A2V_BF16X3) return HD == 64 && S <= AB48_ROWS ? 1 : 0;
  if (dtype != A2V_BF16) return 0;
  if (S <= 64 * A2V_ATTN_SEQ_TILES) return 1;
  return HD == 64 && g_a2v_attn_long ? 1 : 0;
}

// dq32 must be zero on entry; delta is (N,H,S) fp32 scratch.  dQ is left in dq32 (fp32, (N,S,H,HD));
// the caller casts it into the q slot of dqkv (a2v_cast2d).
// 1 when a2v_attention_bwd can also accumulate the column sums of dqkv (qkv_bias_grad) for this (dtype, S, HD)
extern "C" int a2v_attention_bwd_fused_bias(int dtype, int S, int HD) {
  if (dtype == A2V_BF16X3) return HD == 64 && S <= AB48_ROWS ? 1 : 0;
  return dtype == A2V_BF16 && HD == 64 && S <= ABS_ROWS && g_a2v_attn_long != 2 ? 1 : 0;
}

extern "C" int a2v_attention_bwd(int dtype, const void* qkv, const void* out, const void* d_out, const float* lse,
                                 const float* mdist, const unsigned* dropmask, float* delta, void* dqkv, float* dq32, float* dalibi, float* qkv_bias_grad, const int* pos,
                                 const float* slopes, const float* alibi_scale, int N, int S, int H, int HD, float scale,
                                 float drop_p, unsigned drop_seed, hipStream_t stream) {
  if (qkv_bias_grad && !a2v_attention_bwd_fused_bias(dtype, S, HD)) return A2V_ERR_ARG;
  if (N <= 0 || S <= 0 || H <= 0 || !slopes || !alibi_scale || !lse || !delta || drop_p < 0.f || drop_p >= 1.f)
    return A2V_ERR_ARG;
  AttnBwdArgs a{qkv, d_out, lse, delta, dqkv, dq32, dalibi, pos, slopes, alibi_scale, N, S, H, scale,
                make_drop(drop_p, drop_seed), 0, qkv_bias_grad, mdist,
                a2v_attention_dropmask_words(dtype, S, HD) ? dropmask : nullptr};
  if (dtype == A2V_BF16) {
    if (HD == 64) return attn_bwd_launch<bf16, 64>(a, out, delta, stream);
    if (HD == 32) return attn_bwd_launch<bf16, 32>(a, out, delta, stream);
    return A2V_ERR_ARG;
  }
  if (dtype == A2V_BF16X3) {          // fp32 storage; S <= 160 at HD = 64: split products on the bf16 matrix cores, else the fp32 kernels
    if (HD == 64 && S <= AB48_ROWS) {
      char* const ws = a.dalibi ? a2v_stream_ws(stream) : nullptr;
      const dim3 grid(1, H, N);
      a.dal_part = (a.dalibi && ws && (long long)grid.y * grid.z <= A2V_DAL_PART_MAX) ? (float*)(ws + A2V_WS_DAL) : nullptr;
      const bool dr = a.drop.thresh != 0;
      const bool four = (S + 63) / 64 < (S + 47) / 48;
#define AX3_GO(D_, W_)                                                                                                       \
  {                                                                                                                           \
    (void)hipFuncSetAttribute((const void*)attn_bwd_x3_kernel<D_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, AX3_LDS);   \
    hipLaunchKernelGGL((attn_bwd_x3_kernel<D_, W_>), grid, dim3(64 * W_), AX3_LDS, stream, a, (const float*)out);            \
  }
      if (four) { if (dr) AX3_GO(true, 4) else AX3_GO(false, 4) } else { if (dr) AX3_GO(true, 3) else AX3_GO(false, 3) }
#undef AX3_GO
      if (a.dal_part) hipLaunchKernelGGL(dalibi_fold_kernel, dim3(H), dim3(256), 0, stream, a.dalibi, (int)(grid.x * grid.z), a.dal_part);
      A2V_CHECK_LAUNCH();
      return A2V_OK;
    }
    dtype = A2V_F32;
  }
  if (dtype == A2V_F32) {
    if (HD == 64) return attn_bwd_launch<float, 64>(a, out, delta, stream);
    if (HD == 32) return attn_bwd_launch<float, 32>(a, out, delta, stream);
    if (HD == 16) return attn_bwd_launch<float, 16>(a, out, delta, stream);
    return A2V_ERR_ARG;
  }
  return A2V_ERR_DTYPE;
}
