// BC-learning source mixup (reference nn/data2vec2.py:453-498 compute_gain_torch, :536-598 mixup block):
//   gain_db[b,f] = 10 log10 max( sum_k |rfft(hann * frame_f)|^2[k] * A_weight[k], 1e-8 )
//   p = 1 / (1 + 10^((G1-G2)/20) * (1-r)/r),  G = max_f gain_db ;  out = (p s1 + (1-p) s2) / sqrt(p^2+(1-p)^2)
// One workgroup per (clip, frame): the windowed frame and a cos/sin twiddle table live in LDS, thread k owns
// frequency bin k (direct DFT, twiddles by a rotation recurrence re-seeded from the table every 16 taps).
#include "a2v_common.h"

__global__ __launch_bounds__(256) void mixup_gain_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                                         const float* __restrict__ aweight, float* __restrict__ gain_db,
                                                         int L, int n_fft, int hop, int nframes, float floor_) {
  extern __shared__ float sm[];
  float* fr = sm;                 // [n_fft] windowed frame
  float* tc = sm + n_fft;         // [n_fft] cos(2 pi j / n_fft)
  float* ts = tc + n_fft;         // [n_fft] sin
  __shared__ float red[4];
  const int f = blockIdx.x, b = blockIdx.y;
  const float* xb = x + (long long)b * L + (long long)f * hop;
  for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
    fr[i] = xb[i] * window[i];
    float s, c;
    sincospif(2.f * (float)i / (float)n_fft, &s, &c);
    tc[i] = c; ts[i] = s;
  }
  __syncthreads();
  const int nbins = n_fft / 2 + 1;
  float part = 0.f;
  // The twiddle of tap n + 1 is the twiddle of tap n turned by the bin's own angle: one table read per 16 taps (re-seeding keeps
  // the recurrence's drift at ~1e-6) instead of two gathered LDS reads per tap -- the table lookups, not the FMAs, bounded this
  // kernel (three LDS reads per two FMAs); fr[n] is a broadcast read.
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
    float re = 0.f, im = 0.f;
    const float cd = tc[k], sd = ts[k];                   // k < n_fft / 2 + 1 <= n_fft
    for (int n0 = 0; n0 < n_fft; n0 += 16) {
      const int j = (int)(((long long)k * n0) % n_fft);
      float c = tc[j], sn = ts[j];
      const int n1 = min(n0 + 16, n_fft);
      for (int n = n0; n < n1; ++n) {
        const float x = fr[n];
        re = fmaf(x, c, re);
        im = fmaf(x, sn, im);
        const float c2 = fmaf(c, cd, -sn * sd);
        sn = fmaf(sn, cd, c * sd);
        c = c2;
      }
    }
    part += (re * re + im * im) * aweight[k];
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float g = fmaxf(red[0] + red[1] + red[2] + red[3], floor_);
    gain_db[(long long)b * nframes + f] = 10.f * log10f(g);
  }
}

__global__ __launch_bounds__(256) void mixup_apply_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                          const float* __restrict__ gain_db, float* __restrict__ out,
                                                          int B, int L, int nframes, float r,
                                                          const unsigned char* __restrict__ mix_mask) {
  __shared__ float sp[2];
  const int b = blockIdx.y;
  if (mix_mask && !mix_mask[b]) {          // mixup_prob < 1: this clip stays as it is (data2vec2.py:540-547,584-589)
    const float* s1 = x + (long long)b * L;
    float* o = out + (long long)b * L;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i < L; i += gridDim.x * blockDim.x * 4) {
      if (i + 3 < L) *(f32x4*)(o + i) = *(const f32x4*)(s1 + i);
      else for (int j = i; j < L; ++j) o[j] = s1[j];
    }
    return;
  }
  const int b2 = perm[b];
  if (threadIdx.x < 64) {           // wave 0: max over frames of both clips
    float g1 = -INFINITY, g2 = -INFINITY;
    for (int f = threadIdx.x; f < nframes; f += 64) {
      g1 = fmaxf(g1, gain_db[(long long)b * nframes + f]);
      g2 = fmaxf(g2, gain_db[(long long)b2 * nframes + f]);
    }
    g1 = wave_max(g1); g2 = wave_max(g2);
    if (threadIdx.x == 0) {
      const float p = 1.f / (1.f + powf(10.f, (g1 - g2) / 20.f) * (1.f - r) / r);
      sp[0] = p;
      sp[1] = rsqrtf(p * p + (1.f - p) * (1.f - p));
    }
  }
  __syncthreads();
  const float p = sp[0], inv = sp[1];
  const float* s1 = x + (long long)b * L;
  const float* s2 = x + (long long)b2 * L;
  float* o = out + (long long)b * L;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i < L; i += gridDim.x * blockDim.x * 4) {
    if (i + 3 < L) {
      const f32x4 a = *(const f32x4*)(s1 + i), c = *(const f32x4*)(s2 + i);
      *(f32x4*)(o + i) = (a * p + c * (1.f - p)) * inv;
    } else {
      for (int j = i; j < L; ++j) o[j] = (s1[j] * p + s2[j] * (1.f - p)) * inv;
    }
  }
}

// gain_db: (B, nframes) fp32, nframes = (L - n_fft) / hop + 1; window (n_fft) and aweight (n_fft/2+1) are
// the reference's hann window and linear-power A-weighting table.
extern "C" int a2v_mixup_gain(const float* x, const float* window, const float* aweight, float* gain_db, int B, int L,
                              int n_fft, int hop, float floor_, hipStream_t st) {
  if (B <= 0 || n_fft <= 0 || n_fft > 4096 || hop <= 0 || L < n_fft) return A2V_ERR_ARG;      // 3 n_fft floats of LDS (<= 48 KB)
  const int nframes = (L - n_fft) / hop + 1;
  hipLaunchKernelGGL(mixup_gain_kernel, dim3(nframes, B), dim3(256), 3 * n_fft * sizeof(float), st, x, window, aweight,
                     gain_db, L, n_fft, hop, nframes, floor_);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// out may not alias x (clips read their mixing partner).  L % 4 == 0 and 16-B aligned rows give vector access.
extern "C" int a2v_mixup_apply(const float* x, const int* perm, const float* gain_db, float* out, int B, int L, int nframes,
                               float r, const unsigned char* mix_mask, hipStream_t st) {
  if (B <= 0 || L <= 0 || nframes <= 0 || !(r > 0.f && r <= 1.f) || x == out || (L & 3)) return A2V_ERR_ARG;
  int gx = (L / 4 + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(mixup_apply_kernel, dim3(gx, B), dim3(256), 0, st, x, perm, gain_db, out, B, L, nframes, r, mix_mask);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}
