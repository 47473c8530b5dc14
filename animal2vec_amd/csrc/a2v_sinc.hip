// SincConv front end for gfx950 (reference nn/sinc.py + nn/utils.py:1105-1117,1430-1431):
//   a2v_sinc_filters_fwd/bwd : band-pass FIR synthesis from (low_hz_, band_hz_) and its gradient
//   a2v_sinc_fused_fwd       : reflect-pad + conv1d(1->C,k) + channel LayerNorm + PSwish in ONE pass;
//                              the (B,C,L) fp32 intermediate of the reference (40.6 MB/clip) never
//                              exists -- output is written once, channels-last (B,L,Cp).
//   a2v_sinc_fused_bwd       : recomputes the conv tile, back-propagates PSwish/LN, and reduces the
//                              filter gradient dW[c][tap] = sum_t dconv[t][c] * xpad[t+tap] on the
//                              matrix cores; per-channel parameter gradients ride in registers.
// The convolution runs on v_mfma_f32_16x16x4_f32 (exact fp32, like the reference's .float() conv).
#include "a2v_common.h"
#include <type_traits>

#define SINC_FP 66          // LDS pitch (floats) of one filter row: conflict-free A/B reads
#define SINC_MAXK 64        // taps padded to 64 (k <= 63)
#define SINC_CP 128         // padded channel count handled by one pass (C <= 128)

// ---------------------------------------------------------------------------------- filter synthesis
// filters[c][j], c < Cp (rows >= C are zero), j < k;  n_ (k/2), window (k/2) as in sinc.py:264-276
__global__ void sinc_filters_fwd_kernel(const float* __restrict__ low_hz, const float* __restrict__ band_hz,
                                        const float* __restrict__ n_, const float* __restrict__ window,
                                        float* __restrict__ filt, int C, int Cp, int k, float min_low, float min_band,
                                        float nyq) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  float* f = filt + (long long)c * k;
  if (c >= C) {
    for (int j = 0; j < k; ++j) f[j] = 0.f;
    return;
  }
  const int half = k / 2;
  const float low = min_low + fabsf(low_hz[c]);                                    // sinc.py:184
  const float high = fminf(fmaxf(low + min_band + fabsf(band_hz[c]), min_low), nyq);  // :187-191
  const float band = high - low;                                                   // :192
  const float inv = 1.f / (2.f * band);
  for (int j = 0; j < half; ++j) {
    const float n = n_[j];
    const float v = (sinf(high * n) - sinf(low * n)) / n * 2.f * window[j];        // :204
    f[j] = v * inv;                                                                // :218
    f[k - 1 - j] = v * inv;                                                        // :210 flip
  }
  f[half] = 2.f * band * inv;                                                      // :207
}

// dlow[c], dband[c] are ACCUMULATED (+=) from dfilt[c][j]
__global__ void sinc_filters_bwd_kernel(const float* __restrict__ low_hz, const float* __restrict__ band_hz,
                                        const float* __restrict__ n_, const float* __restrict__ window,
                                        const float* __restrict__ dfilt, float* __restrict__ dlow,
                                        float* __restrict__ dband, int C, int k, float min_low, float min_band,
                                        float nyq) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int half = k / 2;
  const float lo_p = low_hz[c], ba_p = band_hz[c];
  const float low = min_low + fabsf(lo_p);
  const float pre = low + min_band + fabsf(ba_p);
  const float high = fminf(fmaxf(pre, min_low), nyq);
  const bool inrange = pre >= min_low && pre <= nyq;
  const float band = high - low;
  const float* df = dfilt + (long long)c * k;
  float g_high = 0.f, g_low = 0.f;
  for (int j = 0; j < half; ++j) {
    const float n = n_[j], w = window[j];
    const float G = df[j] + df[k - 1 - j];
    const float F = (sinf(high * n) - sinf(low * n)) * w / (n * band);
    g_high += G * (w * cosf(high * n) / band - F / band);
    g_low += G * (-w * cosf(low * n) / band + F / band);
  }
  const float gh = inrange ? g_high : 0.f;
  const float sl = lo_p > 0.f ? 1.f : (lo_p < 0.f ? -1.f : 0.f);
  const float sb = ba_p > 0.f ? 1.f : (ba_p < 0.f ? -1.f : 0.f);
  dlow[c] += sl * (g_low + gh);
  dband[c] += sb * gh;
}

__device__ __forceinline__ int reflect_idx(int i, int L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// ---------------------------------------------------------------------------------- fused forward
// Workgroup = 4 waves; each wave owns 16 consecutive samples per tile, the block walks `tiles_per_blk`
// tiles of 64 samples.  Orientation: D[row = channel][col = time] so a time step's 128 channels sit in
// ONE lane column -> LayerNorm statistics are in-lane sums + 2 shuffles.
template <typename T>
__global__ __launch_bounds__(256) void sinc_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ filt,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                             const float* __restrict__ ps_a, const float* __restrict__ ps_b,
                                                             T* __restrict__ y, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, int B, int L, int C, int k,
                                                             float eps, int tiles_per_blk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sPar = (float*)smem;                       // [4][128]: ln_w, ln_b, ps_a, ps_b
  float* sX = sPar + 4 * SINC_CP;                   // 2 x [64 + 64] padded signal of the current / next tile
  T* sOut = (T*)(sX + 256);                         // [4 waves][16][SINC_CP + 8]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int half = k / 2;

  // A band-pass row is symmetric (sinc.py:204-218: [left, centre, flip(left)]; a2v_sinc_filters_fwd writes f[j] and
  // f[k-1-j] from one value), so  sum_j W[c][j] x[t+j] = sum_{j<half} W[c][j] (x[t+j] + x[t+k-1-j]) + W[c][half] x[t+half]:
  // the contraction runs over half + 1 <= 32 folded taps instead of 64 -- half the fp32 MFMA work, which is what bounds
  // this kernel (exact fp32 matrix cores run at 1/16 of the bf16 rate) -- and the 8 x 8 folded filter values a lane needs
  // fit in registers for the block's whole life (the LDS image of the filters cost one ds_read per MFMA).
  // bf16 output mode: the (folded, K = 32) contraction runs on the bf16 matrix cores in split precision -- x = xh + xl,
  // w = wh + wl with bf16 halves (16 mantissa bits each way), acc += wl xh + wh xl + wh xh in fp32: products exact to
  // 2^-17, i.e. ~1e-5 on the conv output against 4e-3 of the bf16 store, at 3 x 16 cycles per 16 x 16 x 32 tile instead of
  // 8 x 32 for the exact fp32 MFMA (which the fp32 parity mode keeps).
  constexpr bool SPLIT = std::is_same<T, bf16>::value;
  float wa[SPLIT ? 1 : 8][8];
  bf16x8 wah[SPLIT ? 8 : 1], wal[SPLIT ? 8 : 1];
  if constexpr (SPLIT) {
#pragma unroll
    for (int jc = 0; jc < 8; ++jc) {
      const int c = 16 * jc + fr;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 8 * g + e;
        // (clamped loads used unconditionally: behind `c < C && j <= half` each of the 64 was its own exec region closed by
        // s_waitcnt vmcnt(0) -- 64 global round trips in a row at the head of every workgroup, longer than its 20 tiles of work)
        // s_waitcnt vmcnt(0) -- 64 global round trips in a row at the head of every workgroup, longer than its 20 tiles of work;
        // and a select is sunk back into a branch: the mask is a FACTOR, which needs the loaded value whatever it is)
        const float wv = filt[(long long)min(c, C - 1) * k + min(j, half)];
        const float w = wv * ((c < C && j <= half) ? (j == half ? 0.5f : 1.f) : 0.f);   // (centre: see xf)
        const bf16 h = (bf16)w;
        wah[jc][e] = h;
        wal[jc][e] = (bf16)(w - (float)h);
      }
    }
  } else {
#pragma unroll
    for (int jc = 0; jc < 8; ++jc)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int c = 16 * jc + fr, j = 4 * ks + g;
        const float wv = filt[(long long)min(c, C - 1) * k + min(j, half)];
        wa[jc][ks] = wv * ((c < C && j <= half) ? (j == half ? 0.5f : 1.f) : 0.f);
      }
  }
  for (int i = tid; i < SINC_CP; i += 256) {
    const bool ok = i < C;
    sPar[i] = ok ? ln_w[i] : 0.f;
    sPar[SINC_CP + i] = ok ? ln_b[i] : 0.f;
    sPar[2 * SINC_CP + i] = ok ? ps_a[i] : 0.f;
    sPar[3 * SINC_CP + i] = ok ? ps_b[i] : 0.f;
  }
  const int tiles_per_clip = (L + 63) / 64;
  const long long ntiles = (long long)B * tiles_per_clip;
  T* myOut = sOut + wid * 16 * (SINC_CP + 8);

  // the padded signal of tile it + 1 is requested before tile it is multiplied and lands in the other half of sX after it:
  // one barrier per tile and no exposed global round trip (was: load -> barrier -> compute, twice a barrier per tile)
  auto xsrc = [&](int b_, int t0_) { return x[(long long)b_ * L + reflect_idx(t0_ - half + (tid & 127), L)]; };
  const long long tile_first = (long long)blockIdx.x * tiles_per_blk;
  // (clip, first step) of the current tile, advanced by hand (a 64-bit division per tile is ~150 scalar instructions, and there were two)
  int b = (int)(tile_first / tiles_per_clip);
  int t0 = (int)(tile_first - (long long)b * tiles_per_clip) * 64;
  if (tile_first < ntiles && tid < 128) sX[tid] = xsrc(b, t0);
  __syncthreads();
  for (int it = 0; it < tiles_per_blk; ++it, t0 += 64) {
    const long long tile = tile_first + it;
    if (tile >= ntiles) break;
    if (t0 >= tiles_per_clip * 64) { t0 = 0; ++b; }
    const float* sXc = sX + (it & 1) * 128;
    const bool more = it + 1 < tiles_per_blk && tile + 1 < ntiles;
    // (every thread, every tile: no control flow around the load; the last tile re-reads itself)
    const bool wrap = t0 + 64 >= tiles_per_clip * 64;
    const float xnext = xsrc(more ? (wrap ? b + 1 : b) : b, more ? (wrap ? 0 : t0 + 64) : t0);

    f32x4 acc[8];
#pragma unroll
    for (int jc = 0; jc < 8; ++jc) acc[jc] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tl = wid * 16 + fr;                     // local time of this lane
    if constexpr (SPLIT) {
      // B[k = 8g + e][col = t]: folded taps, split.  xf[t][j] = x[t + j] + x[t + k-1-j] for EVERY j: at the centre tap that is 2 x,
      // against a filter value stored halved (exact); taps j > half meet zero weights
      bf16x8 xh, xl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 8 * g + e;
        const float xf = sXc[tl + j] + sXc[tl + max(k - 1 - j, 0)];
        const bf16 h = (bf16)xf;
        xh[e] = h;
        xl[e] = (bf16)(xf - (float)h);
      }
#pragma unroll
      for (int jc = 0; jc < 8; ++jc) {
        acc[jc] = A2V_MFMA_16x16x32(wal[jc], xh, acc[jc], 0, 0, 0);
        acc[jc] = A2V_MFMA_16x16x32(wah[jc], xl, acc[jc], 0, 0, 0);
        acc[jc] = A2V_MFMA_16x16x32(wah[jc], xh, acc[jc], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int j = ks * 4 + g;                        // folded tap: B[k = j][col = t] = x[t+j] + x[t+k-1-j] (centre: 2 x, weight halved)
        const float xb = sXc[tl + j] + sXc[tl + max(k - 1 - j, 0)];
#pragma unroll
        for (int jc = 0; jc < 8; ++jc) acc[jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[jc][ks], xb, acc[jc], 0, 0, 0);
      }
    }
    // acc[jc][r] = conv[c = 16jc+4g+r][t = t0+tl]
    float s = 0.f;
#pragma unroll
    for (int jc = 0; jc < 8; ++jc)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += (16 * jc + 4 * g + r < C) ? acc[jc][r] : 0.f;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int jc = 0; jc < 8; ++jc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[jc][r] - mean;
        q += (16 * jc + 4 * g + r < C) ? d * d : 0.f;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    const int t = t0 + tl;
    if (g == 0 && t < L && mean_out) {
      mean_out[(long long)b * L + t] = mean;
      rstd_out[(long long)b * L + t] = rstd;
    }
#pragma unroll
    for (int jc = 0; jc < 8; ++jc) {
      const int c0 = 16 * jc + 4 * g;
      const f32x4 w = *(const f32x4*)(sPar + c0), bb = *(const f32x4*)(sPar + SINC_CP + c0);
      const f32x4 pa = *(const f32x4*)(sPar + 2 * SINC_CP + c0), pb = *(const f32x4*)(sPar + 3 * SINC_CP + c0);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float yl = (acc[jc][r] - mean) * rstd * w[r] + bb[r];
        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-pb[r] * yl));   // v_rcp_f32 (1 ulp), not the 10-instruction IEEE division
        o[r] = (c0 + r < C) ? yl * pa[r] * sg : 0.f;                    // utils.py:1431
      }
      Vec4<T>::store(myOut + fr * (SINC_CP + 8) + c0, o);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // coalesced store of this wave's 16 rows x 128 channels (8 elements per lane per pass)
    constexpr int LPR = SINC_CP / 8;                 // lanes per row = 16
#pragma unroll
    for (int pass = 0; pass < 16 / (64 / LPR); ++pass) {
      const int rrow = pass * (64 / LPR) + lane / LPR;
      const int cc = (lane % LPR) * 8;
      const int tt = t0 + wid * 16 + rrow;
      if (tt < L) {
        float v[8];
        Vec8<T>::load(myOut + rrow * (SINC_CP + 8) + cc, v);
        Vec8<T>::store(y + ((long long)b * L + tt) * SINC_CP + cc, v);
      }
    }
    if (more && tid < 128) sX[((it + 1) & 1) * 128 + tid] = xnext;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------- fused backward
// Orientation: D[row = time][col = channel]: a lane owns channels {fr + 16 jc}, so per-channel
// parameter gradients accumulate in registers over all tiles of the block; the dconv accumulator
// layout is directly the A operand (k = time, order permuted) of dW = dconv^T . Xpad.
// grads: dfilt [Cp][k] fp32 (atomic), dpar [4][128] fp32 (atomic): d ln_w, d ln_b, d ps_a, d ps_b
// EIGHT waves: wave (wt, wc) owns time steps 16 wt + [0, 16) of the 64-step tile and channels 64 wc + [0, 64).  As four waves x 128
// channels the kernel needed 512 registers per lane (dW 64, acc / dh / h 96, eight values of each of four parameters and four
// parameter gradients 64, ...): one wave per SIMD, a third of its VALU instructions were v_accvgpr moves and nothing ran under the
// exp / rcp chains, the cross-lane sums (ds_bpermute round trips) or the fp32 MFMAs -- 0.06 of its HBM roofline.  Half the channels
// per wave fit in 256 registers, two waves share a SIMD, and the only thing the halves exchange is the LayerNorm backward's two
// row sums (through LDS, one more barrier per tile); the 16-lane sums are DPP adds.
#define SB_NJ 4      // 16-channel blocks per wave
template <typename T>
__global__ __launch_bounds__(512) void sinc_fused_bwd_kernel(const float* __restrict__ x, const float* __restrict__ filt,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                             const float* __restrict__ ps_a, const float* __restrict__ ps_b,
                                                             const T* __restrict__ dy, const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, float* __restrict__ dfilt,
                                                             float* __restrict__ dpar, int B, int L, int C, int k,
                                                             int tiles_per_blk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sF = (float*)smem;                         // [128][SINC_FP]
  float* sX = sF + SINC_CP * SINC_FP;               // 2 x [128]
  float* sMean = sX + 256;                          // 2 x [64]
  float* sRstd = sMean + 128;                       // 2 x [64]
  float* sRed = sRstd + 128;                        // [channel half][64 t][2]: partial row sums of the LayerNorm backward
  T* sDy = (T*)(sRed + 256);                        // 2 x [64][SINC_CP + 8]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = wid & 3, wc = wid >> 2, cb0 = 64 * wc;
  const int fr = lane & 15, g = lane >> 4;
  const int half = k / 2;

  // bf16 mode: the conv recompute runs on the bf16 matrix cores in split precision exactly as the forward does (x = xh + xl,
  // w = wh + wl, three products per tile: 3 x 16 cycles instead of 8 x 32 of the exact fp32 MFMA, same sums as the forward's);
  // the folded filter rows sit in LDS already split, [c][32 taps] bf16 each.
  constexpr bool SPLIT = std::is_same<T, bf16>::value;
  bf16* const sFh = (bf16*)sF;
  bf16* const sFl = sFh + SINC_CP * 32;
  if constexpr (SPLIT) {
    for (int i = tid; i < SINC_CP * 32; i += 512) {
      const int c = i >> 5, j = i & 31;
      const float wv = filt[(long long)min(c, C - 1) * k + min(j, half)];
      const float w = wv * ((c < C && j <= half) ? (j == half ? 0.5f : 1.f) : 0.f);   // (centre: see xf below; the mask is a factor)
      const bf16 h = (bf16)w;
      sFh[i] = h;
      sFl[i] = (bf16)(w - (float)h);
    }
  } else {
    for (int i = tid; i < SINC_CP * SINC_MAXK; i += 512) {
      const int c = i >> 6, j = i & 63;
      const float wv = filt[(long long)min(c, C - 1) * k + min(j, half)];
      sF[c * SINC_FP + j] = wv * ((c < C && j <= half) ? (j == half ? 0.5f : 1.f) : 0.f);
    }
  }
  // per-lane channel parameters: channels c = cb0 + fr + 16 jc
  float pw[SB_NJ], pb_[SB_NJ], pa[SB_NJ], pbs[SB_NJ];
#pragma unroll
  for (int jc = 0; jc < SB_NJ; ++jc) {
    const int c = cb0 + fr + 16 * jc;
    const bool ok = c < C;
    const int cc_ = min(c, C - 1);                  // (clamped, unconditional loads)
    const float l0 = ln_w[cc_], l1 = ln_b[cc_], l2 = ps_a[cc_], l3 = ps_b[cc_];
    const float okf = ok ? 1.f : 0.f;               // (a factor, not a select: selects are sunk back into exec regions)
    pw[jc] = l0 * okf; pb_[jc] = l1 * okf;
    pa[jc] = l2 * okf; pbs[jc] = l3 * okf;
  }
  float pbl[SB_NJ];                                 // -ps_b log2(e): exp(-ps_b y) = exp2(pbl y), one multiply less per element
#pragma unroll
  for (int jc = 0; jc < SB_NJ; ++jc) pbl[jc] = -1.44269504088896f * pbs[jc];
  float gw[SB_NJ], gb[SB_NJ], ga[SB_NJ], gbs[SB_NJ];
#pragma unroll
  for (int jc = 0; jc < SB_NJ; ++jc) { gw[jc] = 0.f; gb[jc] = 0.f; ga[jc] = 0.f; gbs[jc] = 0.f; }
  // symmetric band-pass rows (see the forward): taps are folded, j <-> k-1-j, so both the conv recompute and the filter
  // gradient run over half + 1 <= 32 folded taps; dfilt[c][j] (j < half) receives d/dW[c][j] + d/dW[c][k-1-j] -- the one
  // number a2v_sinc_filters_bwd reads (it sums the two mirrored entries) -- and the mirrored entries stay untouched
  f32x4 dW[SB_NJ][2];                               // [jc][jt]: rows c = cb0+16jc+4g+r, cols folded tap = 16jt+fr
#pragma unroll
  for (int jc = 0; jc < SB_NJ; ++jc)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) dW[jc][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tiles_per_clip = (L + 63) / 64;
  const long long ntiles = (long long)B * tiles_per_clip;
  const float invC = 1.f / (float)C;

  // Staging of tile it + 1 (signal, LayerNorm statistics, the 64 x 128 dy tile) is requested before tile it is worked on
  // and written to the other LDS buffers after it: one barrier per tile, no exposed global round trip (this kernel runs one
  // workgroup per CU, so nothing else hides it).
  constexpr int DYCH = 64 * (SINC_CP / 8) / 512;      // dy chunks (8 elements) per thread: 2
  struct Stage { float xv, mv, rv; Raw8<T> dyv[DYCH]; };
  auto stage_fetch = [&](int tid, int b_, int t0_, Stage& st) {
    // every thread loads (clamped indices; stage_store keeps what it needs): no control flow around the loads -- inside exec-masked
    // regions the compiler closed each one with s_waitcnt vmcnt(0), a global round trip exposed in every tile.  Rows past L get the
    // last row's statistics: their dy is zeroed below, so nothing of them reaches a sum.
    st.xv = x[(long long)b_ * L + reflect_idx(t0_ - half + (tid & 127), L)];
    const int ts_ = min(t0_ + (tid & 63), L - 1);
    st.mv = mean_in[(long long)b_ * L + ts_];
    st.rv = rstd_in[(long long)b_ * L + ts_];
#pragma unroll
    for (int u = 0; u < DYCH; ++u) {
      const int i = tid + 512 * u;
      const int r = i / (SINC_CP / 8), cc = (i % (SINC_CP / 8)) * 8;
      const int tt = t0_ + r < L ? t0_ + r : L - 1;                       // (rows past L are zeroed by the t0 + tl >= L test below)
      st.dyv[u].load(dy + ((long long)b_ * L + tt) * SINC_CP + cc);
    }
  };
  auto stage_store = [&](int tid, int buf, const Stage& st) {
    if (tid < 128) sX[buf * 128 + tid] = st.xv;
    if (tid < 64) { sMean[buf * 64 + tid] = st.mv; sRstd[buf * 64 + tid] = st.rv; }
#pragma unroll
    for (int u = 0; u < DYCH; ++u) {
      const int i = tid + 512 * u;
      const int r = i / (SINC_CP / 8), cc = (i % (SINC_CP / 8)) * 8;
      float v[8];
      st.dyv[u].unpack(v);
      Vec8<T>::store(sDy + (size_t)buf * 64 * (SINC_CP + 8) + r * (SINC_CP + 8) + cc, v);
    }
  };
  const long long tile_first = (long long)blockIdx.x * tiles_per_blk;
  // (clip, first step) of the current tile, advanced by hand: the 64-bit division per tile was ~300 scalar instructions
  int b = (int)(tile_first / tiles_per_clip);
  int t0 = (int)(tile_first - (long long)b * tiles_per_clip) * 64;
  {
    Stage st0;
    if (tile_first < ntiles) { stage_fetch(tid, b, t0, st0); stage_store(tid, 0, st0); }
  }
  __syncthreads();
  for (int it = 0; it < tiles_per_blk; ++it, t0 += 64) {
    const long long tile = tile_first + it;
    if (tile >= ntiles) break;
    if (t0 >= tiles_per_clip * 64) { t0 = 0; ++b; }
    const int cb_ = it & 1;
    const bool more = it + 1 < tiles_per_blk && tile + 1 < ntiles;
    // lane-derived indices are re-formed from opaque copies in every tile: hoisted out of the loop (dozens of LDS addresses and
    // masks) they were spilled to scratch and reloaded, with a full vmcnt wait, in front of their uses
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int tid_o = tid;                            // (the staging addresses may stay hoisted: a dozen registers)
    const int fr = lane_o & 15, g = lane_o >> 4;
    // always fetched (the last tile of a block re-fetches itself and drops it): behind `if (more)` the loaded registers met the
    // untaken path's in copies placed right after the loads -- a wait for the global round trip at the top of every tile
    Stage stn;
    {
      const bool wrap = t0 + 64 >= tiles_per_clip * 64;
      stage_fetch(tid_o, more ? (wrap ? b + 1 : b) : b, more ? (wrap ? 0 : t0 + 64) : t0, stn);
    }
    const float* const sXc = sX + cb_ * 128;
    const float* const sMc = sMean + cb_ * 64;
    const float* const sRc = sRstd + cb_ * 64;
    const T* const sDc = sDy + (size_t)cb_ * 64 * (SINC_CP + 8);

    // conv recompute: D[row t = 4g+r][col c = fr+16jc]; A = X[t=fr][tap], B = filt[c][tap]
    f32x4 acc[SB_NJ];
#pragma unroll
    for (int jc = 0; jc < SB_NJ; ++jc) acc[jc] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (SPLIT) {
      // A[row = t = fr][k = 8g + e]: folded taps, split.  xf[t][j] = x[t + j] + x[t + k-1-j] for EVERY j: at the centre tap
      // (j = half = k-1-j) that is 2 x, against a filter column stored halved (exact); columns j > half meet zero weights
      bf16x8 xh, xl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 8 * g + e;
        const float xf = sXc[wt * 16 + fr + j] + sXc[wt * 16 + fr + max(k - 1 - j, 0)];
        const bf16 h = (bf16)xf;
        xh[e] = h;
        xl[e] = (bf16)(xf - (float)h);
      }
#pragma unroll
      for (int jc = 0; jc < SB_NJ; ++jc) {
        const bf16x8 wh = *(const bf16x8*)(sFh + (cb0 + 16 * jc + fr) * 32 + 8 * g);
        const bf16x8 wl = *(const bf16x8*)(sFl + (cb0 + 16 * jc + fr) * 32 + 8 * g);
        acc[jc] = A2V_MFMA_16x16x32(xh, wl, acc[jc], 0, 0, 0);
        acc[jc] = A2V_MFMA_16x16x32(xl, wh, acc[jc], 0, 0, 0);
        acc[jc] = A2V_MFMA_16x16x32(xh, wh, acc[jc], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < 8; ++ks) {
        const int j = ks * 4 + g;
        const float xa = sXc[wt * 16 + fr + j] + sXc[wt * 16 + fr + max(k - 1 - j, 0)];
#pragma unroll
        for (int jc = 0; jc < SB_NJ; ++jc) {
          const float wb = sF[(cb0 + 16 * jc + fr) * SINC_FP + j];                 // (zero for j > half, halved at j = half)
          acc[jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, wb, acc[jc], 0, 0, 0);
        }
      }
    }
    // back-prop through PSwish and LN.  values indexed [jc][r]: t = 16wt+4g+r, c = cb0+fr+16jc
    f32x4 dh[SB_NJ], hh[SB_NJ];
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float mr[4], rr[4];                                // this lane's four rows' statistics
#pragma unroll
    for (int r = 0; r < 4; ++r) { mr[r] = sMc[wt * 16 + 4 * g + r]; rr[r] = sRc[wt * 16 + 4 * g + r]; }
#pragma unroll
    for (int jc = 0; jc < SB_NJ; ++jc) {
      const bool cok = cb0 + fr + 16 * jc < C;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tl = wt * 16 + 4 * g + r;
        const float h = (acc[jc][r] - mr[r]) * rr[r];
        const float yl = h * pw[jc] + pb_[jc];
        const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(pbl[jc] * yl));    // sigmoid(ps_b yl)
        const float dzv = Cvt<T>::to(sDc[tl * (SINC_CP + 8) + cb0 + fr + 16 * jc]);      // (unconditional: a select, not a branch)
        const float dz = (cok && t0 + tl < L) ? dzv : 0.f;
        // z = ps_a yl sg:  dz/d ps_a = yl sg,  dz/d ps_b = ps_a yl^2 sg (1 - sg),  dz/d yl = ps_a (sg + ps_b yl sg (1 - sg))
        const float u = yl * fmaf(-sg, sg, sg);
        const float A = dz * pa[jc];
        ga[jc] = fmaf(dz, yl * sg, ga[jc]);
        gbs[jc] = fmaf(A * u, yl, gbs[jc]);
        const float dyl = A * fmaf(pbs[jc], u, sg);
        gw[jc] = fmaf(dyl, h, gw[jc]);
        gb[jc] += dyl;
        const float d = dyl * pw[jc];
        dh[jc][r] = d;
        hh[jc][r] = h;                                  // (padded channels: d = 0 here and the result is masked below)
        s1[r] += d;
        s2[r] = fmaf(d, h, s2[r]);
      }
      // the block's parameter-gradient sums are formed HERE (opaque uses): left to the scheduler they sink to the end of the tile
      // and the 64 products feeding them stay live across the filter-gradient MFMAs (spilled accumulators, reloaded with vmcnt(0))
      asm volatile("" : "+v"(ga[jc]), "+v"(gbs[jc]), "+v"(gw[jc]), "+v"(gb[jc]));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[r] = row16_sum(s1[r]);
      s2[r] = row16_sum(s2[r]);
      if (fr == 0) *(float2*)(sRed + (wc * 64 + wt * 16 + 4 * g + r) * 2) = make_float2(s1[r], s2[r]);
    }
    __syncthreads();                                  // the other channel half's sums (this tile's, written above)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 o = *(const float2*)(sRed + ((wc ^ 1) * 64 + wt * 16 + 4 * g + r) * 2);
      s1[r] = (s1[r] + o.x) * invC;
      s2[r] = (s2[r] + o.y) * invC;
    }
    // dconv[t][c] = rstd * (dh - mean(dh) - h * mean(dh*h)), zero for padded channel / time
#pragma unroll
    for (int jc = 0; jc < SB_NJ; ++jc) {
      const bool cok = cb0 + fr + 16 * jc < C;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tl = wt * 16 + 4 * g + r;
        const float v = rr[r] * (dh[jc][r] - s1[r] - hh[jc][r] * s2[r]);
        dh[jc][r] = (cok && t0 + tl < L) ? v : 0.f;
      }
    }
    // dW[c][tap] += sum_t dconv[t][c] * xpad[t + tap]: k-step r has k-slot g <-> t = 16wt+4g+r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tl = wt * 16 + 4 * g + r;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int j = 16 * jt + fr;                      // B[k=t][col = folded tap j]; column half sums 2 x (halved at the flush),
        const float xb = sXc[tl + j] + sXc[tl + max(k - 1 - j, 0)];      // columns j > half are never flushed
#pragma unroll
        for (int jc = 0; jc < SB_NJ; ++jc)
          dW[jc][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[jc][r], xb, dW[jc][jt], 0, 0, 0);
      }
    }
    if (more) stage_store(tid_o, cb_ ^ 1, stn);
    __syncthreads();
  }
  // flush: filter gradient rows c = cb0+16jc+4g+r, cols tap = 16jt+fr
#pragma unroll
  for (int jc = 0; jc < SB_NJ; ++jc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cb0 + 16 * jc + 4 * g + r;
      if (c >= C) continue;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int tap = 16 * jt + fr;
        if (tap <= half) atomicAdd(dfilt + (long long)c * k + tap, tap == half ? 0.5f * dW[jc][jt][r] : dW[jc][jt][r]);
      }
    }
  // per-channel parameter grads: reduce over the 4 lane groups (same fr), then atomics
#pragma unroll
  for (int jc = 0; jc < SB_NJ; ++jc) {
    float a0 = gw[jc], a1 = gb[jc], a2 = ga[jc], a3 = gbs[jc];
    a0 += __shfl_xor(a0, 16, 64); a0 += __shfl_xor(a0, 32, 64);
    a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
    a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
    a3 += __shfl_xor(a3, 16, 64); a3 += __shfl_xor(a3, 32, 64);
    const int c = cb0 + fr + 16 * jc;
    if (g == 0 && c < C) {
      atomicAdd(dpar + c, a0);
      atomicAdd(dpar + SINC_CP + c, a1);
      atomicAdd(dpar + 2 * SINC_CP + c, a2);
      atomicAdd(dpar + 3 * SINC_CP + c, a3);
    }
  }
}

// ---------------------------------------------------------------------------------- general shapes
// More than 128 filters or more than 63 taps (BASELINE config 5: 296 filters x 149 taps at 44.1 kHz): the
// same two kernels, blocked.  Forward: all 16 NJC channels of a time step stay in one lane column (LayerNorm statistics
// remain in-lane sums), the taps are walked in chunks of 64 with the filter chunk re-staged in LDS per chunk, and the
// result goes out straight from registers (8 / 16 B per lane).  Backward: a workgroup recomputes the whole conv tile
// (all channels, all taps -- the LayerNorm backward needs every channel) but reduces the filter gradient only for
// its (channel group, tap chunk) = blockIdx.y, so dW fits in registers; the per-channel parameter gradients are
// accumulated in LDS.  [superseded for the backward: see sinc_dconv_gen_kernel / sinc_dw_gen_kernel below]
#define SINC_GEN_NJC 20                 // channels padded to 320
#define SINC_GEN_CP (16 * SINC_GEN_NJC)
#define SINC_GEN_MAXCH 4                // tap chunks of 64: k <= 255
#define SINC_GEN_CGB 5                  // channel blocks (of 16) per dW workgroup

template <typename T, int NJC>
__global__ __launch_bounds__(256) void sinc_fused_fwd_gen_kernel(const float* __restrict__ x, const float* __restrict__ filt,
                                                                 const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                                 const float* __restrict__ ps_a, const float* __restrict__ ps_b,
                                                                 T* __restrict__ y, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out, int B, int L, int C, int k,
                                                                 float eps, int tiles_per_blk, int nchunks) {
  constexpr int CP = 16 * NJC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sF = (float*)smem;                         // [CP][SINC_FP]: the current chunk of 64 taps
  float* sPar = sF + CP * SINC_FP;                  // [4][CP]
  float* sX = sPar + 4 * CP;                        // [64 + 64 nchunks]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int half = k / 2;
  for (int i = tid; i < CP; i += 256) {
    const bool ok = i < C;
    sPar[i] = ok ? ln_w[i] : 0.f;
    sPar[CP + i] = ok ? ln_b[i] : 0.f;
    sPar[2 * CP + i] = ok ? ps_a[i] : 0.f;
    sPar[3 * CP + i] = ok ? ps_b[i] : 0.f;
  }
  const int tiles_per_clip = (L + 63) / 64;
  const long long ntiles = (long long)B * tiles_per_clip;
  const int win = 64 + 64 * nchunks;
  for (int it = 0; it < tiles_per_blk; ++it) {
    const long long tile = (long long)blockIdx.x * tiles_per_blk + it;
    if (tile >= ntiles) break;
    const int b = (int)(tile / tiles_per_clip);
    const int t0 = (int)(tile - (long long)b * tiles_per_clip) * 64;
    __syncthreads();
    for (int i = tid; i < win; i += 256) sX[i] = x[(long long)b * L + reflect_idx(t0 - half + i, L)];
    f32x4 acc[NJC];
#pragma unroll
    for (int jc = 0; jc < NJC; ++jc) acc[jc] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tl = wid * 16 + fr;
    for (int ch = 0; ch < nchunks; ++ch) {
      __syncthreads();                                // the previous chunk's readers are done (first pass: sX / sPar ordering)
      for (int i = tid; i < CP * 64; i += 256) {
        const int c = i >> 6, j = ch * 64 + (i & 63);
        sF[c * SINC_FP + (i & 63)] = (c < C && j < k) ? filt[(long long)c * k + j] : 0.f;
      }
      __syncthreads();
#pragma unroll 2
      for (int ks = 0; ks < 16; ++ks) {
        const float xb = sX[tl + ch * 64 + ks * 4 + g];
#pragma unroll
        for (int jc = 0; jc < NJC; ++jc) {
          const float wa = sF[(16 * jc + fr) * SINC_FP + ks * 4 + g];
          acc[jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xb, acc[jc], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int jc = 0; jc < NJC; ++jc)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += (16 * jc + 4 * g + r < C) ? acc[jc][r] : 0.f;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int jc = 0; jc < NJC; ++jc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[jc][r] - mean;
        q += (16 * jc + 4 * g + r < C) ? d * d : 0.f;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    const int t = t0 + tl;
    if (g == 0 && t < L && mean_out) {
      mean_out[(long long)b * L + t] = mean;
      rstd_out[(long long)b * L + t] = rstd;
    }
    if (t < L) {
#pragma unroll
      for (int jc = 0; jc < NJC; ++jc) {
        const int c0 = 16 * jc + 4 * g;
        const f32x4 w = *(const f32x4*)(sPar + c0), bb = *(const f32x4*)(sPar + CP + c0);
        const f32x4 pa = *(const f32x4*)(sPar + 2 * CP + c0), pb = *(const f32x4*)(sPar + 3 * CP + c0);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yl = (acc[jc][r] - mean) * rstd * w[r] + bb[r];
          const float sg = 1.f / (1.f + __expf(-pb[r] * yl));
          o[r] = (c0 + r < C) ? yl * pa[r] * sg : 0.f;
        }
        Vec4<T>::store(y + ((long long)b * L + t) * CP + c0, o);
      }
    }
  }
}

// K1: dconv[t][c] (fp32, channel pitch CP) for every channel of every tile -> scratch, plus the per-channel parameter
// gradients (LDS atomics, flushed once per workgroup).
template <typename T, int NJC>
__global__ __launch_bounds__(256) void sinc_dconv_gen_kernel(const float* __restrict__ x, const float* __restrict__ filt,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                             const float* __restrict__ ps_a, const float* __restrict__ ps_b,
                                                             const T* __restrict__ dy, const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, float* __restrict__ dconv,
                                                             float* __restrict__ dpar, int B, int L, int C, int k,
                                                             int tiles_per_blk, int nchunks) {
  constexpr int CP = 16 * NJC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sF = (float*)smem;                         // [CP][SINC_FP]
  float* sPar = sF + CP * SINC_FP;                  // [4][CP] parameters
  float* sG = sPar + 4 * CP;                        // [4][CP] parameter gradients
  float* sX = sG + 4 * CP;                          // [64 + 64 nchunks]
  float* sMean = sX + 64 + 64 * SINC_GEN_MAXCH;     // [64]
  float* sRstd = sMean + 64;                        // [64]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int half = k / 2;
  for (int i = tid; i < CP; i += 256) {
    const bool ok = i < C;
    sPar[i] = ok ? ln_w[i] : 0.f;
    sPar[CP + i] = ok ? ln_b[i] : 0.f;
    sPar[2 * CP + i] = ok ? ps_a[i] : 0.f;
    sPar[3 * CP + i] = ok ? ps_b[i] : 0.f;
    sG[i] = 0.f; sG[CP + i] = 0.f; sG[2 * CP + i] = 0.f; sG[3 * CP + i] = 0.f;
  }
  const int tiles_per_clip = (L + 63) / 64;
  const long long ntiles = (long long)B * tiles_per_clip;
  const float invC = 1.f / (float)C;
  const int win = 64 + 64 * nchunks;

  for (int it = 0; it < tiles_per_blk; ++it) {
    const long long tile = (long long)blockIdx.x * tiles_per_blk + it;
    if (tile >= ntiles) break;
    const int b = (int)(tile / tiles_per_clip);
    const int t0 = (int)(tile - (long long)b * tiles_per_clip) * 64;
    __syncthreads();
    for (int i = tid; i < win; i += 256) sX[i] = x[(long long)b * L + reflect_idx(t0 - half + i, L)];
    if (tid < 64) {
      const int t = t0 + tid;
      sMean[tid] = t < L ? mean_in[(long long)b * L + t] : 0.f;
      sRstd[tid] = t < L ? rstd_in[(long long)b * L + t] : 0.f;
    }
    // conv recompute: D[row t = 4g+r][col c = fr + 16 jc]
    f32x4 acc[NJC];
#pragma unroll
    for (int jc = 0; jc < NJC; ++jc) acc[jc] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nchunks; ++ch) {
      __syncthreads();
      for (int i = tid; i < CP * 64; i += 256) {
        const int c = i >> 6, j = ch * 64 + (i & 63);
        sF[c * SINC_FP + (i & 63)] = (c < C && j < k) ? filt[(long long)c * k + j] : 0.f;
      }
      __syncthreads();
#pragma unroll 2
      for (int ks = 0; ks < 16; ++ks) {
        const float xa = sX[wid * 16 + fr + ch * 64 + ks * 4 + g];
#pragma unroll
        for (int jc = 0; jc < NJC; ++jc) {
          const float wb = sF[(16 * jc + fr) * SINC_FP + ks * 4 + g];
          acc[jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, wb, acc[jc], 0, 0, 0);
        }
      }
    }
    // pass 1: d = dL/dh per element (kept in acc's place: h is recomputable from the statistics only with the conv
    // value, so keep both: h in acc, d in dh), LayerNorm-backward row sums, parameter gradients
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float mrow[4], rrow[4];
    bool tok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tl = wid * 16 + 4 * g + r;
      mrow[r] = sMean[tl]; rrow[r] = sRstd[tl]; tok[r] = t0 + tl < L;
    }
    f32x4 dh[NJC];
#pragma unroll
    for (int jc = 0; jc < NJC; ++jc) {
      const int c = fr + 16 * jc;
      const bool cok = c < C;
      const float pw = sPar[c], pb_ = sPar[CP + c], pa = sPar[2 * CP + c], pbs = sPar[3 * CP + c];
      float gw = 0.f, gb = 0.f, ga = 0.f, gbs = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tl = wid * 16 + 4 * g + r;
        const float h = (acc[jc][r] - mrow[r]) * rrow[r];
        const float yl = h * pw + pb_;
        const float sg = 1.f / (1.f + __expf(-pbs * yl));
        const float dz = (cok && tok[r]) ? Cvt<T>::to(dy[((long long)b * L + t0 + tl) * CP + c]) : 0.f;
        const float dyl = dz * pa * (sg + yl * pbs * sg * (1.f - sg));
        ga += dz * yl * sg;
        gbs += dz * yl * pa * sg * (1.f - sg) * yl;
        gw += dyl * h;
        gb += dyl;
        const float d = dyl * pw;
        acc[jc][r] = cok ? h : 0.f;
        dh[jc][r] = d;
        s1[r] += d;
        s2[r] += cok ? d * h : 0.f;
      }
      gw += __shfl_xor(gw, 16, 64); gw += __shfl_xor(gw, 32, 64);
      gb += __shfl_xor(gb, 16, 64); gb += __shfl_xor(gb, 32, 64);
      ga += __shfl_xor(ga, 16, 64); ga += __shfl_xor(ga, 32, 64);
      gbs += __shfl_xor(gbs, 16, 64); gbs += __shfl_xor(gbs, 32, 64);
      if (g == 0 && cok) {
        atomicAdd(sG + c, gw); atomicAdd(sG + CP + c, gb); atomicAdd(sG + 2 * CP + c, ga); atomicAdd(sG + 3 * CP + c, gbs);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
      s1[r] *= invC; s2[r] *= invC;
    }
    // pass 2: dconv[t][c] = rstd * (d - mean(d) - h * mean(d h)) -> scratch (zero for padded channels / time)
#pragma unroll
    for (int jc = 0; jc < NJC; ++jc) {
      const int c = fr + 16 * jc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tl = wid * 16 + 4 * g + r;
        if (tok[r]) {
          const float v = rrow[r] * (dh[jc][r] - s1[r] - acc[jc][r] * s2[r]);
          dconv[((long long)b * L + t0 + tl) * CP + c] = c < C ? v : 0.f;
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 4 * CP; i += 256) {
    const int c = i % CP;
    if (c < C && sG[i] != 0.f) atomicAdd(dpar + i, sG[i]);
  }
}

// K2: dW[c][tap] += sum_t dconv[t][c] * xpad[t + tap] for this workgroup's (channel group, tap chunk) = blockIdx.y
template <int NJC, int CGB>
__global__ __launch_bounds__(256) void sinc_dw_gen_kernel(const float* __restrict__ x, const float* __restrict__ dconv,
                                                          float* __restrict__ dfilt, int B, int L, int C, int k,
                                                          int tiles_per_blk, int nchunks) {
  constexpr int CP = 16 * NJC;
  __shared__ float sX[64 + 64 + 16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int half = k / 2;
  const int tg = blockIdx.y % nchunks, cg = blockIdx.y / nchunks;
  f32x4 dW[CGB][4];                                 // rows c = 16 (cg CGB + jl) + 4g + r, cols tap = 64 tg + 16 jt + fr
#pragma unroll
  for (int jl = 0; jl < CGB; ++jl)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) dW[jl][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int tiles_per_clip = (L + 63) / 64;
  const long long ntiles = (long long)B * tiles_per_clip;
  for (int it = 0; it < tiles_per_blk; ++it) {
    const long long tile = (long long)blockIdx.x * tiles_per_blk + it;
    if (tile >= ntiles) break;
    const int b = (int)(tile / tiles_per_clip);
    const int t0 = (int)(tile - (long long)b * tiles_per_clip) * 64;
    __syncthreads();
    if (tid < 128) sX[tid] = x[(long long)b * L + reflect_idx(t0 - half + tg * 64 + tid, L)];     // samples t0 + tl + tap - half
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tl = wid * 16 + 4 * g + r;
      const bool tok = t0 + tl < L;
      float dh[CGB];
#pragma unroll
      for (int jl = 0; jl < CGB; ++jl) {
        const int c = 16 * (cg * CGB + jl) + fr;
        dh[jl] = (tok && c < C) ? dconv[((long long)b * L + t0 + tl) * CP + c] : 0.f;
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const float xb = sX[tl + 16 * jt + fr];
#pragma unroll
        for (int jl = 0; jl < CGB; ++jl)
          dW[jl][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[jl], xb, dW[jl][jt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int jl = 0; jl < CGB; ++jl)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * (cg * CGB + jl) + 4 * g + r;
      if (c >= C) continue;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int tap = tg * 64 + 16 * jt + fr;
        if (tap < k) atomicAdd(dfilt + (long long)c * k + tap, dW[jl][jt][r]);
      }
    }
}

// =============================================================================================
extern "C" int a2v_sinc_filters_fwd(const float* low_hz, const float* band_hz, const float* n_, const float* window,
                                    float* filt, int C, int Cp, int k, float min_low_hz, float min_band_hz,
                                    float sample_rate, hipStream_t st) {
  if (C <= 0 || Cp < C || k < 3 || !(k & 1)) return A2V_ERR_ARG;
  hipLaunchKernelGGL(sinc_filters_fwd_kernel, dim3((Cp + 63) / 64), dim3(64), 0, st, low_hz, band_hz, n_, window, filt, C,
                     Cp, k, min_low_hz, min_band_hz, sample_rate * 0.5f);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_sinc_filters_bwd(const float* low_hz, const float* band_hz, const float* n_, const float* window,
                                    const float* dfilt, float* dlow, float* dband, int C, int k, float min_low_hz,
                                    float min_band_hz, float sample_rate, hipStream_t st) {
  if (C <= 0 || k < 3 || !(k & 1)) return A2V_ERR_ARG;
  hipLaunchKernelGGL(sinc_filters_bwd_kernel, dim3((C + 63) / 64), dim3(64), 0, st, low_hz, band_hz, n_, window, dfilt, dlow,
                     dband, C, k, min_low_hz, min_band_hz, sample_rate * 0.5f);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

static int sinc_tiles_per_block(long long ntiles) {
  int tpb = (int)((ntiles + 2047) / 2048);
  if (tpb < 1) tpb = 1;
  return tpb;
}

// y: (B, L, 128) channels-last, channels >= C zero.  filt: (>=C rows, k) fp32 from a2v_sinc_filters_fwd.
extern "C" int a2v_sinc_fused_fwd(int dtype, const float* x, const float* filt, const float* ln_w, const float* ln_b,
                                  const float* ps_alpha, const float* ps_beta, void* y, float* mean, float* rstd, int B,
                                  int L, int C, int k, float eps, hipStream_t st) {
  if (B <= 0 || L < 128 || C <= 0 || !(k & 1)) return A2V_ERR_ARG;
  if (C > SINC_CP || k > 63) {                       // general shapes: y is (B, L, 320)
    const int nch = (k + 63) / 64;
    if (C > SINC_GEN_CP || nch > SINC_GEN_MAXCH || L < 64 + 64 * nch) return A2V_ERR_ARG;
    const long long nt = (long long)B * ((L + 63) / 64);
    const int tpbg = sinc_tiles_per_block(nt);
    const int gridg = (int)((nt + tpbg - 1) / tpbg);
    const int ldsg = (SINC_GEN_CP * SINC_FP + 4 * SINC_GEN_CP + 64 + 64 * SINC_GEN_MAXCH) * 4;
    if (dtype == A2V_BF16) {
      (void)hipFuncSetAttribute((const void*)sinc_fused_fwd_gen_kernel<bf16, SINC_GEN_NJC>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsg);
      hipLaunchKernelGGL((sinc_fused_fwd_gen_kernel<bf16, SINC_GEN_NJC>), dim3(gridg), dim3(256), ldsg, st, x, filt, ln_w, ln_b,
                         ps_alpha, ps_beta, (bf16*)y, mean, rstd, B, L, C, k, eps, tpbg, nch);
    } else if (dtype == A2V_F32) {
      (void)hipFuncSetAttribute((const void*)sinc_fused_fwd_gen_kernel<float, SINC_GEN_NJC>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsg);
      hipLaunchKernelGGL((sinc_fused_fwd_gen_kernel<float, SINC_GEN_NJC>), dim3(gridg), dim3(256), ldsg, st, x, filt, ln_w, ln_b,
                         ps_alpha, ps_beta, (float*)y, mean, rstd, B, L, C, k, eps, tpbg, nch);
    } else
      return A2V_ERR_DTYPE;
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  const long long ntiles = (long long)B * ((L + 63) / 64);
  const int tpb = sinc_tiles_per_block(ntiles);
  const int grid = (int)((ntiles + tpb - 1) / tpb);
  const int esz = dtype == A2V_BF16 ? 2 : 4;
  const int lds = (4 * SINC_CP + 256) * 4 + 4 * 16 * (SINC_CP + 8) * esz;
  if (dtype == A2V_BF16) {
    (void)hipFuncSetAttribute((const void*)sinc_fused_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(sinc_fused_fwd_kernel<bf16>, dim3(grid), dim3(256), lds, st, x, filt, ln_w, ln_b, ps_alpha, ps_beta,
                       (bf16*)y, mean, rstd, B, L, C, k, eps, tpb);
  } else if (dtype == A2V_F32) {
    (void)hipFuncSetAttribute((const void*)sinc_fused_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(sinc_fused_fwd_kernel<float>, dim3(grid), dim3(256), lds, st, x, filt, ln_w, ln_b, ps_alpha, ps_beta,
                       (float*)y, mean, rstd, B, L, C, k, eps, tpb);
  } else
    return A2V_ERR_DTYPE;
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// dfilt (C rows used, k) and dpar (4 x 128) are ACCUMULATED (atomics): zero them first.
extern "C" int a2v_sinc_fused_bwd(int dtype, const float* x, const float* filt, const float* ln_w, const float* ln_b,
                                  const float* ps_alpha, const float* ps_beta, const void* dy, const float* mean,
                                  const float* rstd, float* dfilt, float* dpar, float* scratch, int B, int L, int C, int k,
                                  hipStream_t st) {
  if (B <= 0 || L < 128 || C <= 0 || !(k & 1)) return A2V_ERR_ARG;
  if (C > SINC_CP || k > 63) {                       // general shapes: dy is (B, L, 320), dpar is (4, 320)
    const int nch = (k + 63) / 64;
    if (C > SINC_GEN_CP || nch > SINC_GEN_MAXCH || L < 64 + 64 * nch || !scratch) return A2V_ERR_ARG;
    const long long nt = (long long)B * ((L + 63) / 64);
    const int tpbg = sinc_tiles_per_block(nt);
    const int gridg = (int)((nt + tpbg - 1) / tpbg);
    const int ldsg = (SINC_GEN_CP * SINC_FP + 8 * SINC_GEN_CP + 64 + 64 * SINC_GEN_MAXCH + 128) * 4;
    if (dtype == A2V_BF16) {
      (void)hipFuncSetAttribute((const void*)sinc_dconv_gen_kernel<bf16, SINC_GEN_NJC>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsg);
      hipLaunchKernelGGL((sinc_dconv_gen_kernel<bf16, SINC_GEN_NJC>), dim3(gridg), dim3(256), ldsg, st, x, filt, ln_w, ln_b,
                         ps_alpha, ps_beta, (const bf16*)dy, mean, rstd, scratch, dpar, B, L, C, k, tpbg, nch);
    } else if (dtype == A2V_F32) {
      (void)hipFuncSetAttribute((const void*)sinc_dconv_gen_kernel<float, SINC_GEN_NJC>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsg);
      hipLaunchKernelGGL((sinc_dconv_gen_kernel<float, SINC_GEN_NJC>), dim3(gridg), dim3(256), ldsg, st, x, filt, ln_w, ln_b,
                         ps_alpha, ps_beta, (const float*)dy, mean, rstd, scratch, dpar, B, L, C, k, tpbg, nch);
    } else
      return A2V_ERR_DTYPE;
    int tpb2 = (int)((nt + 255) / 256);
    if (tpb2 < 1) tpb2 = 1;
    const int ncg = (SINC_GEN_NJC + SINC_GEN_CGB - 1) / SINC_GEN_CGB;
    hipLaunchKernelGGL((sinc_dw_gen_kernel<SINC_GEN_NJC, SINC_GEN_CGB>), dim3((unsigned)((nt + tpb2 - 1) / tpb2), (unsigned)(ncg * nch)),
                       dim3(256), 0, st, x, scratch, dfilt, B, L, C, k, tpb2, nch);
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  const long long ntiles = (long long)B * ((L + 63) / 64);
  int tpb = (int)((ntiles + 255) / 256);        // one 8-wave workgroup per CU (168 registers: a second one does not fit), one round:
                                                // every workgroup flushes its dW / parameter sums once, 16 K atomics onto 8 K addresses
  if (tpb < 1) tpb = 1;
  const int grid = (int)((ntiles + tpb - 1) / tpb);
  const int esz = dtype == A2V_BF16 ? 2 : 4;
  const int lds = (SINC_CP * SINC_FP + 256 + 256 + 256) * 4 + 2 * 64 * (SINC_CP + 8) * esz;
  if (dtype == A2V_BF16) {
    (void)hipFuncSetAttribute((const void*)sinc_fused_bwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(sinc_fused_bwd_kernel<bf16>, dim3(grid), dim3(512), lds, st, x, filt, ln_w, ln_b, ps_alpha, ps_beta,
                       (const bf16*)dy, mean, rstd, dfilt, dpar, B, L, C, k, tpb);
  } else if (dtype == A2V_F32) {
    (void)hipFuncSetAttribute((const void*)sinc_fused_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(sinc_fused_bwd_kernel<float>, dim3(grid), dim3(512), lds, st, x, filt, ln_w, ln_b, ps_alpha, ps_beta,
                       (const float*)dy, mean, rstd, dfilt, dpar, B, L, C, k, tpb);
  } else
    return A2V_ERR_DTYPE;
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// channel pitch of the fused front end's output / dy / dpar rows for a (filters, taps) spec: 128 for the 8 kHz specs
// (C <= 128, k <= 63), 320 for the general kernels (C <= 320, k <= 255), -1 if unsupported
extern "C" int a2v_sinc_out_channels(int C, int k) {
  if (C <= 0 || k < 3 || !(k & 1)) return -1;
  if (C <= SINC_CP && k <= 63) return SINC_CP;
  if (C <= SINC_GEN_CP && (k + 63) / 64 <= SINC_GEN_MAXCH) return SINC_GEN_CP;
  return -1;
}
