// Shared device helpers for the animal2vec gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes everywhere; no 32-lane idioms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

// The 16-bit storage / MFMA operand type of this object.  The product library (liba2v_hip.so) is built with bfloat16; the SAME
// sources built with -DA2V_ELT_F16 give liba2v_hip_f16.so, in which every "bf16" below is IEEE half (11 significand bits at the
// same matrix rate): the EMA teacher's no-grad forward runs on it (the reference's own recipe is fp16,
// configs/MeerKAT/a2v_large_pretrain_best.yaml:4; profiles/r06_fp16_teacher.json: target error 2.8e-2 -> 3.4e-3).  Both libraries
// export the same C ABI (include/a2v.h); dtype code A2V_BF16 means "this library's 16-bit type".
#ifdef A2V_ELT_F16
typedef _Float16 bf16;
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2;
#define A2V_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define A2V_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define A2V_ASM_MFMA_16x16x32 "v_mfma_f32_16x16x32_f16"
#define A2V_ASM_CVT_PK "v_cvt_pk_f16_f32"
#define A2V_ELT_NAME "f16"
#else
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define A2V_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define A2V_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define A2V_ASM_MFMA_16x16x32 "v_mfma_f32_16x16x32_bf16"
#define A2V_ASM_CVT_PK "v_cvt_pk_bf16_f32"
#define A2V_ELT_NAME "bf16"
#endif
// the two 16-bit halves of a packed pair as fp32 (bf16: a shift / a mask; half: conversions)
__device__ __forceinline__ float a2v_lo16_f32(unsigned w) {
#ifdef A2V_ELT_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu));
#else
  return __builtin_bit_cast(float, w << 16);
#endif
}
__device__ __forceinline__ float a2v_hi16_f32(unsigned w) {
#ifdef A2V_ELT_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
#else
  return __builtin_bit_cast(float, w & 0xffff0000u);
#endif
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define A2V_OK 0
#define A2V_ERR_ARG (-1)
#define A2V_ERR_DTYPE (-2)
#define A2V_ERR_LAUNCH (-3)

// dtype codes used across the C ABI
#define A2V_F32 0
#define A2V_BF16 1
#define A2V_BF16X3 2        // a2v_conv_gemm only: bf16 split operand images (a2v_split_bf16), fp32 C / C2 / R
#define A2V_BF16X3S 3       // a2v_conv_gemm only: [hi ; lo] stacked images, the three products of a 64-wide grouped conv in one pass
#define A2V_ERR_UNSUPPORTED (-4)

#define A2V_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return -(int)(1000 + (int)e__);   \
  } while (0)

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to(float v) { return v; }
  static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct Cvt<bf16> {
  static __device__ __forceinline__ float to(bf16 v) { return (float)v; }
  static __device__ __forceinline__ bf16 from(float v) { return (bf16)v; }
};

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return Cvt<T>::to(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = Cvt<T>::from(v); }

// 4-wide vector load/store as floats (T = float: 16 B, T = bf16: 8 B); pointers must be aligned.
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Vec4<bf16> {
  static __device__ __forceinline__ f32x4 load(const bf16* p) {
    bf16x4 v = *(const bf16x4*)p;
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
  }
  static __device__ __forceinline__ void store(bf16* p, f32x4 v) {
    bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)p = r;
  }
};

// 8-wide (T = bf16: one 16-B access; T = float: two)
template <typename T> struct Vec8;
template <> struct Vec8<bf16> {
  static __device__ __forceinline__ void load(const bf16* p, float* o) {
    bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  static __device__ __forceinline__ void store(bf16* p, const float* o) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
    *(bf16x8*)p = v;
  }
};
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float* o) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float* o) {
    f32x4 a = {o[0], o[1], o[2], o[3]}, b = {o[4], o[5], o[6], o[7]};
    *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
  }
};

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
#define A2V_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  A2V_DPP_ADD(0xB1);     // quad_perm [1,0,3,2]
  A2V_DPP_ADD(0x4E);     // quad_perm [2,3,0,1]
  A2V_DPP_ADD(0x141);    // row_half_mirror
  A2V_DPP_ADD(0x140);    // row_mirror
#undef A2V_DPP_ADD
  return v;
}

// 64-lane sum, result in every lane.  DPP adds inside the 16-lane rows (quad_perm, row_half_mirror, row_mirror), then
// row_bcast15 / row_bcast31 carry the row sums up to lane 63, which is broadcast through an SGPR: six VALU ops and
// one v_readlane instead of six dependent ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float wave_sum(float v) {
#define A2V_DPP_ADD(ctrl, rmask, bctrl) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, bctrl))
  A2V_DPP_ADD(0xB1, 0xF, true);     // quad_perm [1,0,3,2]
  A2V_DPP_ADD(0x4E, 0xF, true);     // quad_perm [2,3,0,1]
  A2V_DPP_ADD(0x141, 0xF, true);    // row_half_mirror
  A2V_DPP_ADD(0x140, 0xF, true);    // row_mirror: every lane of a row holds the row sum
  A2V_DPP_ADD(0x142, 0xA, false);   // row_bcast15 into rows 1 and 3
  A2V_DPP_ADD(0x143, 0xC, false);   // row_bcast31 into rows 2 and 3: lane 63 holds the total
#undef A2V_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// raw 8-element load kept packed (bf16: 4 VGPRs, float: 8) so that a prefetched row costs little until it is used
template <typename T> struct Raw8;
template <> struct Raw8<bf16> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16* p) { v = *(const uint4*)p; }
  __device__ __forceinline__ void unpack(float* o) const {
    o[0] = a2v_lo16_f32(v.x); o[1] = a2v_hi16_f32(v.x);
    o[2] = a2v_lo16_f32(v.y); o[3] = a2v_hi16_f32(v.y);
    o[4] = a2v_lo16_f32(v.z); o[5] = a2v_hi16_f32(v.z);
    o[6] = a2v_lo16_f32(v.w); o[7] = a2v_hi16_f32(v.w);
  }
};
template <> struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
  __device__ __forceinline__ void unpack(float* o) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
  }
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU, as torch.nn.GELU() default.  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 6e-7 evaluated in fp32,
// i.e. below fp32 round-off of the surrounding arithmetic) -- one v_exp + one v_rcp + 6 FMAs instead of
// libm's branchy erff, which made GELU epilogues as expensive as a K=1024 main loop.
// gelu_parts: cdf = Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and e = exp(-x^2 / 2) (shared by the pdf of the gradient);
// v_rcp_f32 (1 ulp) instead of __frcp_rn, whose correctly-rounded expansion is an 11-instruction division.
// gelu_half: q = Phi(|x|) - 1/2 (in [0, 1/2)) and e = exp(-x^2 / 2).  gelu(x) = x Phi(x) = x/2 + |x| q needs no sign select;
// Phi(x) = 1/2 + copysign(q, x).
// FAST (bf16 epilogues of the persistent GEMM only, where the result is rounded to 8 bits): A-S 7.1.25, three terms,
// |abs err of erf| <= 2.5e-5 -- two FMAs fewer per element in an epilogue that is VALU-bound (112 elements per lane and tile).
template <bool FAST = false>
__device__ __forceinline__ void gelu_half(float x, float& q, float& e) {
#ifdef A2V_EXP_NOGELU     /* timing builds (wrong results): no GELU arithmetic at all -- the upper bound of what a cheaper evaluation can buy */
  q = x; e = x; return;
#endif
  const float ax = fabsf(x) * 0.70710678118654752f;
  float p;
  float t;
  if (FAST) {
    t = __builtin_amdgcn_rcpf(fmaf(0.47047f, ax, 1.f));
    p = fmaf(0.5f * 0.7478556f, t, 0.5f * -0.0958798f);                // polynomial pre-scaled by 1/2
    p = fmaf(p, t, 0.5f * 0.3480242f);
  } else {
    t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
  }
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);          // exp(-x^2/2) = 2^(-x^2 log2(e) / 2)
  q = fmaf(-(p * t), e, 0.5f);                                        // 1/2 - (1 - erf(|x|/sqrt2)) / 2
}
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
  float q;
  gelu_half<false>(x, q, e);
  cdf = 0.5f + copysignf(q, x);
}
__device__ __forceinline__ float erf_fast(float x) {
  float cdf, e;
  gelu_parts(x * 1.41421356237309505f, cdf, e);
  return 2.f * cdf - 1.f;
}
template <bool FAST = false>
__device__ __forceinline__ float gelu_t(float x) {
  float q, e;
  gelu_half<FAST>(x, q, e);
  return fmaf(fabsf(x), q, 0.5f * x);
}
template <bool FAST = false>
__device__ __forceinline__ float gelu_grad_t(float x) {
  float q, e;
  gelu_half<FAST>(x, q, e);
  return fmaf(x * 0.39894228040143268f, e, 0.5f + copysignf(q, x));
}
// gelu(x) and gelu'(x) from ONE evaluation of the shared parts (forward epilogues that keep the derivative for the backward)
__device__ __forceinline__ float gelu_with_grad(float x, float& grad) {
  float q, e;
  gelu_half<false>(x, q, e);
  grad = fmaf(x * 0.39894228040143268f, e, 0.5f + copysignf(q, x));
  return fmaf(fabsf(x), q, 0.5f * x);
}
// The same for a PAIR of values in packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of
// FMAs per issue slot; only v_rcp / v_exp and the sign / magnitude bit operations stay per component).  Same formulas, same
// constants, same operation order as gelu_with_grad: bit-identical results.
typedef float a2v_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ a2v_f32x2 gelu_with_grad2(a2v_f32x2 x, a2v_f32x2& grad) {
#ifdef A2V_EXP_NOGELU
  grad = x; return x;
#endif
  a2v_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const a2v_f32x2 axs = ax * 0.70710678118654752f;
  const a2v_f32x2 den = __builtin_elementwise_fma((a2v_f32x2){0.3275911f, 0.3275911f}, axs, (a2v_f32x2){1.f, 1.f});
  const a2v_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  a2v_f32x2 p = __builtin_elementwise_fma((a2v_f32x2){0.5f * 1.061405429f, 0.5f * 1.061405429f}, t, (a2v_f32x2){0.5f * -1.453152027f, 0.5f * -1.453152027f});
  p = __builtin_elementwise_fma(p, t, (a2v_f32x2){0.5f * 1.421413741f, 0.5f * 1.421413741f});
  p = __builtin_elementwise_fma(p, t, (a2v_f32x2){0.5f * -0.284496736f, 0.5f * -0.284496736f});
  p = __builtin_elementwise_fma(p, t, (a2v_f32x2){0.5f * 0.254829592f, 0.5f * 0.254829592f});
  const a2v_f32x2 xx = x * x * -0.72134752044448170f;
  const a2v_f32x2 e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  const a2v_f32x2 q = __builtin_elementwise_fma(-(p * t), e, (a2v_f32x2){0.5f, 0.5f});
  const a2v_f32x2 cq = {copysignf(q[0], x[0]), copysignf(q[1], x[1])};
  grad = __builtin_elementwise_fma(x * 0.39894228040143268f, e, (a2v_f32x2){0.5f, 0.5f} + cq);
  return __builtin_elementwise_fma(ax, q, x * 0.5f);
}
// gelu'(x) in 8 bits (A2V_EPI_DACT8, bf16 throughput mode only): gelu' lies in [-0.129, 1.129]; q = round(200 g') + 26 in [0, 252],
// g' ~ (q - 26) / 200 -- a grid of 1/200 on which 0 (dead units) and 1 (saturated units) are exact; |error| <= 0.0025, the size of the
// bf16 rounding (2^-9 relative) of a derivative near 1/2.  What the fc1 epilogue keeps for the backward and the fc2 input-gradient
// epilogue multiplies by: a quarter of the fc1 launch's output bytes and a third of that backward launch's extra input bytes less.
// (v_cvt_pk_u8_f32 rounds to nearest even, saturates to [0, 255] and writes one byte of a dword: one FMA + one conversion per element)
__device__ __forceinline__ unsigned dact8_put(unsigned w, unsigned byte, float d) {
  return __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d, 200.f, 26.f), byte, w);
}
__device__ __forceinline__ unsigned dact8_pack4(float a, float b, float c, float d) {
  return dact8_put(dact8_put(dact8_put(dact8_put(0u, 0u, a), 1u, b), 2u, c), 3u, d);
}
__device__ __forceinline__ void dact8_unpack4(unsigned w, float* o) {
  o[0] = fmaf((float)(w & 0xFFu), 0.005f, -0.13f);
  o[1] = fmaf((float)((w >> 8) & 0xFFu), 0.005f, -0.13f);
  o[2] = fmaf((float)((w >> 16) & 0xFFu), 0.005f, -0.13f);
  o[3] = fmaf((float)(w >> 24), 0.005f, -0.13f);
}
__device__ __forceinline__ float gelu_f(float x) { return gelu_t<false>(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_grad_t<false>(x); }

// Counter-based dropout (no mask tensor is ever stored): keep(seed, idx) is a pure function of a per-site
// seed and the element's linear index, so the backward kernels regenerate exactly the forward mask.
struct Drop {
  unsigned seed;      // per call site / micro-batch
  unsigned thresh;    // drop when hash < thresh  (thresh = p * 2^32)
  float scale;        // 1 / (1 - p); 0 threshold disables
};
__device__ __forceinline__ unsigned drop_hash(unsigned seed, unsigned long long idx) {
  unsigned u = (unsigned)idx * 0x9E3779B1u + seed + (unsigned)(idx >> 32) * 0x7F4A7C15u;
  u ^= u >> 16; u *= 0x7FEB352Du; u ^= u >> 15; u *= 0x846CA68Bu; u ^= u >> 16;
  return u;
}
// One 32-bit hash decides TWO neighbouring elements (16 bits each, as the attention sites do): the mask of element idx is
//   keep(idx) = half(idx & 1) of drop_hash(seed, idx >> 1)  >=  thresh >> 16.
// The hash is two v_mul_lo_u32 (quarter rate) + 6 more instructions -- per element it cost as much as a GELU in the GEMM epilogues
// (proj + residual + dropout: 163 us against 105 us for the plain product at 53760 x 1024 x 1024).
__device__ __forceinline__ bool drop_keep(const Drop& d, unsigned long long idx) {
  const unsigned h = drop_hash(d.seed, idx >> 1);
  return ((idx & 1) ? (h >> 16) : (h & 0xFFFFu)) >= (d.thresh >> 16);
}
__device__ __forceinline__ float drop_apply(const Drop& d, unsigned long long idx, float v) {
  return drop_keep(d, idx) ? v * d.scale : 0.f;
}
// N (even) consecutive elements from e0: N/2 hashes when e0 is even (every caller's case: rows of even length, 4- or 8-element
// groups); an odd e0 takes the per-element form of the same mask.
template <int N>
__device__ __forceinline__ void drop_apply_n(const Drop& d, unsigned long long e0, float* v) {
  if (e0 & 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = drop_apply(d, e0 + i, v[i]);
    return;
  }
  const unsigned t16 = d.thresh >> 16;
#pragma unroll
  for (int k = 0; k < N / 2; ++k) {
    const unsigned h = drop_hash(d.seed, (e0 >> 1) + k);
    v[2 * k] = (h & 0xFFFFu) >= t16 ? v[2 * k] * d.scale : 0.f;
    v[2 * k + 1] = (h >> 16) >= t16 ? v[2 * k + 1] * d.scale : 0.f;
  }
}
// Attention probabilities (N*H*S*S elements per layer, the one site where the hash itself shows up in the profile):
// 2-D form, one hash per PAIR of adjacent key columns of a (n, h, q) row -- 16 bits per element, compared with the
// upper 16 bits of the threshold (p resolved to 1/65536).  Kernels whose lanes hold consecutive keys (forward, dQ
// pass) evaluate one hash per two elements; the row term is loop invariant everywhere.
__device__ __forceinline__ unsigned drop_row_term(unsigned seed, unsigned row) { return row * 0x9E3779B1u + seed; }
__device__ __forceinline__ unsigned drop_col_term(unsigned col) { return (col >> 1) * 0x85EBCA6Bu; }
__device__ __forceinline__ unsigned drop_finalize(unsigned u) {
  u ^= u >> 16; u *= 0x7FEB352Du; u ^= u >> 15; u *= 0x846CA68Bu; u ^= u >> 16;
  return u;
}
__device__ __forceinline__ unsigned drop_hash_pair(unsigned row_term, unsigned col) {
  return drop_finalize(row_term + drop_col_term(col));
}
__device__ __forceinline__ bool drop_keep2(const Drop& d, unsigned row_term, unsigned col) {
  const unsigned h = drop_hash_pair(row_term, col);
  const unsigned h16 = (col & 1u) ? (h >> 16) : (h & 0xFFFFu);
  return h16 >= (d.thresh >> 16);
}
static inline Drop make_drop(float p, unsigned seed) {
  Drop d;
  d.seed = seed;
  d.thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  return d;
}

// Bijective XCD-aware remap of a linear workgroup id: ids that share (id % 8) run on one XCD
// (round-robin dispatch), so give each XCD a contiguous chunk of the tile space.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, loc = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
}

// Grouped ("supertile") order on top of the XCD remap: consecutive ids walk 8 m-tiles x all n-tiles column by
// column, so the ~64 tiles resident on one XCD at a time form an 8 x 8 patch that re-uses 8 A panels and 8
// B panels out of its 4 MiB L2 instead of streaming every B panel from the Infinity Cache (N = 4096: 4x less
// L2-miss traffic than row-major order).
// mode 1 ("W-stationary", a2v_debug_set(23), tiles_n % 4 == 0): the n-tiles are cut into groups of four and the ids walk ALL
// m-tiles of one n-group (8 m x 4 n patches, column by column inside a patch) before the next n-group starts: the 32
// workgroups of an XCD keep reading the same four B panels (2 MiB at K = 1024) round after round while the A panels stream,
// and the XCDs that work on different n-groups walk the same m-range at the same time (their A reads meet in the Infinity
// Cache).  mode 0 re-streams every B panel once per 8 m-tiles.
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn, int mode = 0) {
  if (mode == 1) {
    const int per_ng = tiles_m * 4;
    const int ng = id / per_ng, r = id - ng * per_ng;
    const int mg = r >> 5, r2 = r & 31;
    const int first = mg * 8;
    const int rows = tiles_m - first < 8 ? tiles_m - first : 8;
    const int c = r2 / rows;
    tn = ng * 4 + c;
    tm = first + (r2 - c * rows);
    return;
  }
  const int gsz = 8 * tiles_n;
  const int grp = id / gsz, rem = id - grp * gsz;
  const int first = grp * 8;
  const int rows = tiles_m - first < 8 ? tiles_m - first : 8;
  tn = rem / rows;
  tm = first + (rem - tn * rows);
}

// Host-side caches of per-DEVICE facts (symbol addresses, function attributes, CU counts) are arrays indexed by the current
// device: one process may drive several GPUs (tests on cuda:1, one-process multi-GPU hosts).
#define A2V_MAX_DEVICES 16
static inline int a2v_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d < 0 || d >= A2V_MAX_DEVICES) ? 0 : d;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Device scratch of the library (tile counters of the persistent GEMM, column-sum partial rows and self-cleaning column-sum copies,
// d alibi_scale partial sums, the |q.k| bounds of the long-sequence attention) exists once per STREAM: the stream argument every
// entry point takes selects the workspace, so launches on different streams never share scratch (re-entrant across streams) while
// launches on one stream are ordered anyway.  A workspace is ONE blob of A2V_WS_BYTES with the fixed layout below, either
//   * caller-owned: a2v_bind_workspace(stream, ptr, bytes) (include/a2v.h, ABI 10: a2v_query_workspace() bytes of zeroed device
//     memory) -- any number of streams, nothing static is used for them; or
//   * one of A2V_MAX_STREAMS library-owned static sets, handed to unbound streams in order of first use (a2v_stream_release gives
//     a set back); a fifth unbound stream gets nullptr and its calls take the kernels that need no scratch.
// Nothing is ever allocated by the library.
#define A2V_MAX_STREAMS 4
#define A2V_WS_CTR 0                                   /* 16 x u32: tile counters of the persistent NT GEMM (self-resetting) */
#define A2V_WS_SPREAD 256                              /* [64][3][4096] fp32: self-cleaning copies of column sums (a2v_norm.hip) */
#define A2V_WS_SPREAD_BYTES (64 * 3 * 4096 * 4)
#define A2V_WS_QKB (A2V_WS_SPREAD + A2V_WS_SPREAD_BYTES) /* 64 Ki fp32: max |q|^2, |k|^2 per (row range, n, h) (attn_qk_bound_kernel) */
#define A2V_WS_QKB_BYTES (65536 * 4)
#define A2V_WS_DAL (A2V_WS_QKB + A2V_WS_QKB_BYTES)     /* 64 Ki fp32: per-workgroup partial sums of d alibi_scale */
#define A2V_WS_DAL_BYTES (65536 * 4)
#define A2V_WS_CSPART (A2V_WS_DAL + A2V_WS_DAL_BYTES)  /* 4 Mi fp32: column-sum partial rows of the persistent NT GEMM's epilogue */
#define A2V_WS_CSPART_BYTES ((1 << 22) * 4)
#define A2V_WS_SPTAB (A2V_WS_CSPART + A2V_WS_CSPART_BYTES) /* kept-row table of conv_wgrad_sparse_kernel (rewritten by every call that uses it) */
#define A2V_WS_SPTAB_BYTES (4 << 20)
#define A2V_WS_BYTES (A2V_WS_SPTAB + A2V_WS_SPTAB_BYTES)
char* a2v_stream_ws(hipStream_t stream);               // the stream's workspace (bound or static), nullptr if none is left
