// Split-precision operand images for the bf16x3 parity mode (dtype A2V_BF16X3): x = hi + lo + r with hi = bf16(x),
// lo = bf16(x - hi), |r| <= 2^-17 |x|.  A product a b over split operands, a_hi b_hi + a_lo b_hi + a_hi b_lo, is three exact
// bf16 x bf16 products accumulated in fp32 (what v_mfma_f32_16x16x32_bf16 does) and misses a b by <= 2^-16 relative -- fp32
// grade for a 1e-3 parity bound -- at a third of the bf16 matrix rate instead of the 1/16 of the fp32 MFMA.  The images put
// the three products side by side along the reduction axis, so that the ordinary bf16 GEMM kernels compute them:
//   mode 0  (M, 3K)  [hi | lo | hi]      left operand of an NT product (activations)
//   mode 1  (M, 3K)  [hi | hi | lo]      right operand (weights)
//   mode 2  (2M, K)  [hi ; lo] stacked   operands of the weight-gradient (TN) products, which run as three launches
//                                        (dY_hi, X_hi), (dY_hi, X_lo), (dY_lo, X_hi) into the same fp32 accumulator
// Replaces nothing in the reference: it is how this library reaches the north-star's 1e-3 on the bf16 matrix cores.
#include "a2v_common.h"

template <int U>
__device__ __forceinline__ void split_store(bf16* __restrict__ out, long long M, int K, long long ldo, int mode, long long m, int c,
                                            const float* v) {
  bf16x8 h, l;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    h[q] = (bf16)v[q];
    l[q] = (bf16)(v[q] - (float)h[q]);
  }
  if (mode == 2) {
    *(bf16x8*)(out + m * ldo + c) = h;
    *(bf16x8*)(out + (M + m) * ldo + c) = l;
  } else {
    bf16* const row = out + m * ldo + c;
    *(bf16x8*)row = h;
    *(bf16x8*)(row + K) = mode == 0 ? l : h;
    *(bf16x8*)(row + 2 * K) = mode == 0 ? h : l;
  }
}

// four 8-element chunks per thread and trip, all eight 16-B loads issued before the first store (one chunk per trip ran at 3.2 TB/s)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ out, long long M, int K,
                                                         long long ldx, long long ldo, int mode) {
  const int kv = K / 8;
  const long long total = M * kv;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
    float v[4][8];
    long long mm[4];
    int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * stride;
      const long long ic = i < total ? i : total - 1;
      mm[u] = ic / kv;
      cc[u] = (int)(ic - mm[u] * kv) * 8;
      Vec8<float>::load(x + mm[u] * ldx + cc[u], v[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * stride < total) split_store<0>(out, M, K, ldo, mode, mm[u], cc[u], v[u]);
  }
}

// x (M, K) fp32 with row pitch ldx (elements) -> out bf16, layout by mode (above), output row pitch ldo (0: dense = 3 K, or K in
// mode 2; a larger pitch leaves the caller's zero padding in place: packed conv weights whose taps are padded to 64 channels).
// K % 8 == 0, 16-B aligned pointers.
extern "C" int a2v_split_bf16(const float* x, void* out, long long M, int K, long long ldx, long long ldo, int mode, hipStream_t st) {
  if (M <= 0 || K <= 0 || K % 8 || ldx % 4 || mode < 0 || mode > 2 || (((uintptr_t)x | (uintptr_t)out) & 15)) return A2V_ERR_ARG;
  if (ldo == 0) ldo = mode == 2 ? K : 3LL * K;
  if (ldo % 8 || ldo < (mode == 2 ? K : 3LL * K)) return A2V_ERR_ARG;
  long long blocks = (M * (K / 8) + 1023) / 1024;          // four chunks per thread
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (bf16*)out, M, K, ldx, ldo, mode);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}
