// HBM-bound index / elementwise kernels of the animal2vec pretraining path: token gather under
// clone-batch masking (never materialising the clone_batch-fold copy), decoder input assembly,
// fused masked MSE loss + gradient + variance statistics, EMA, Adam, weight packing, col2im.
// All rows are channels-last; every thread moves 16 bytes where the layout allows it.
#include "a2v_common.h"

struct PtrList32 { const void* p[32]; };

// ---------------------------------------------------------------------------------- gather tokens
// out[n,k,:] = a[n/cb, ids[n,k], :] (+ b[n, ids[n,k], :])        reference base.py:244,447,277-280
template <typename T>
__global__ __launch_bounds__(256) void gather_tokens_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                            const int* __restrict__ ids, T* __restrict__ out, int N,
                                                            int K, int D, int Ta, int cb, Drop drop) {
  const int vecs = D / 8;
  const long long total = (long long)N * K * vecs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vecs);
    const long long nk = i / vecs;
    const int n = (int)(nk / K);
    const int t = ids[nk];
    float x[8];
    Vec8<T>::load(a + ((long long)(n / cb) * Ta + t) * D + v * 8, x);
    if (b) {
      float y[8];
      Vec8<T>::load(b + ((long long)n * Ta + t) * D + v * 8, y);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += y[j];
    }
    if (drop.thresh) {
      drop_apply_n<8>(drop, (unsigned long long)nk * D + v * 8, x);
    }
    Vec8<T>::store(out + nk * D + v * 8, x);
  }
}

// dpos[n,t,:] = restore[n,t] < K ? dtok[n, restore[n,t], :] : 0        (scatter of kept-token grads)
template <typename T>
__global__ __launch_bounds__(256) void scatter_tokens_kernel(const T* __restrict__ dtok, const int* __restrict__ restore,
                                                             T* __restrict__ dpos, int N, int K, int D, int Tn) {
  const int vecs = D / 8;
  const long long total = (long long)N * Tn * vecs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vecs);
    const long long nt = i / vecs;
    const int n = (int)(nt / Tn);
    const int r = restore[nt];
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < K) Vec8<T>::load(dtok + ((long long)n * K + r) * D + v * 8, x);
    Vec8<T>::store(dpos + nt * D + v * 8, x);
  }
}

// dfeat[b,t,:] = sum_{c<cb, n=b*cb+c, kept(n,t)} ( dxm[n,t,:] + dtok[n, restore[n,t], :] )
// gradient of: x.repeat_interleave(cb) -> {x*(1-mask) -> pos-conv input ; gather kept tokens}
template <typename T>
__global__ __launch_bounds__(256) void clone_reduce_kernel(const T* __restrict__ dxm, const T* __restrict__ dtok,
                                                           const int* __restrict__ restore, T* __restrict__ dfeat,
                                                           int B, int cb, int K, int D, int Tn) {
  const int vecs = D / 8;
  const long long total = (long long)B * Tn * vecs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vecs);
    const long long bt = i / vecs;
    const int b = (int)(bt / Tn), t = (int)(bt - (long long)b * Tn);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < cb; ++c) {
      const int n = b * cb + c;
      const int r = restore[(long long)n * Tn + t];
      if (r < K) {
        float y[8];
        Vec8<T>::load(dtok + ((long long)n * K + r) * D + v * 8, y);
        if (dxm) {       // NULL: the pos-conv input gradient was formed at the kept rows only and is already inside dtok
          float x[8];
          Vec8<T>::load(dxm + ((long long)n * Tn + t) * D + v * 8, x);
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] += x[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += y[j];
      }
    }
    Vec8<T>::store(dfeat + bt * D + v * 8, acc);
  }
}

// Two N(0, std) values per 32-bit word (Box-Muller, both outputs).  The radius takes the hashed word, the angle one more
// multiply-xorshift round of it (a mixing bijection of that word: as good as a second hash for two coordinates that enter
// through log and sin/cos, at a third of its cost -- this generator is VALU-bound on its quarter-rate integer multiplies).
// v_log / v_sqrt / v_sin / v_cos directly: log2 of (0, 1], the angle in revolutions.
__device__ __forceinline__ void normal_pair(unsigned h1, float std, float& z0, float& z1) {
  unsigned h2 = (h1 ^ 0x68E31DA4u) * 0x2C1B3C6Du;
  h2 ^= h2 >> 15;
  const float u1 = ((float)h1 + 1.f) * 2.3283064365386963e-10f;            // (0, 1]
  const float u2 = (float)(h2 >> 8) * 5.9604644775390625e-08f;               // [0, 1) in 2^-24 steps
  const float rad = std * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
  z0 = rad * __builtin_amdgcn_cosf(u2);
  z1 = rad * __builtin_amdgcn_sinf(u2);
}

// decoder input (base.py:162-192): out[n,t,:] = r<K ? x[n,r,:] : noise[n,r-K,:], r = restore[n,t]
// noise == NULL: the masked positions draw N(0, noise_std) on the spot -- normal_pair on the 2-D counter hash (a2v_common.h:
// row term of (seed, output row) + pair-of-columns term); the distribution of the reference's normal_(), not torch's Philox
// stream -- instead of reading a materialised (N, Tn-K, D) fp32 tensor.
// One output row per wave and round (the row's source and the division by Tn are wave-uniform; the element-per-thread form
// spent more on two 64-bit divisions per 16 bytes than on the copy).
template <typename T>
__global__ __launch_bounds__(256) void decoder_input_kernel(const T* __restrict__ x, const float* __restrict__ noise,
                                                            const int* __restrict__ restore, T* __restrict__ out, int N,
                                                            int K, int D, int Tn, Drop drop, float noise_std,
                                                            unsigned noise_seed) {
  const int vecs = D / 8;
  const int lane = threadIdx.x & 63;
  const int rows = N * Tn;
  const int nw = gridDim.x * 4;
  for (int nt = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); nt < rows; nt += nw) {
    const int n = nt / Tn;
    const int r = restore[nt];
    T* const orow = out + (long long)nt * D;
    if (r < K) {
      const long long src = ((long long)n * K + r) * D;
      for (int v = lane; v < vecs; v += 64) {
        float o[8];
        Vec8<T>::load(x + src + v * 8, o);
        if (drop.thresh) drop_apply_n<8>(drop, (unsigned long long)src + v * 8, o);      // decoder input_dropout (base.py:163-165)
        Vec8<T>::store(orow + v * 8, o);
      }
    } else if (noise) {
      const float* const nrow = noise + ((long long)n * (Tn - K) + (r - K)) * D;
      for (int v = lane; v < vecs; v += 64) {
        float o[8];
        Vec8<float>::load(nrow + v * 8, o);
        Vec8<T>::store(orow + v * 8, o);
      }
    } else {
      const unsigned rt = drop_row_term(noise_seed, (unsigned)nt);
      for (int v = lane; v < vecs; v += 64) {
        float o[8];
        const unsigned c0 = (unsigned)v * 4u * 0x85EBCA6Bu;
#pragma unroll
        for (int j = 0; j < 4; ++j) normal_pair(drop_finalize(rt + c0 + (unsigned)j * 0x85EBCA6Bu), noise_std, o[2 * j], o[2 * j + 1]);
        Vec8<T>::store(orow + v * 8, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------- d2v loss
// data2vec2.py:850-864,904-909,1005-1021 + compute_var statistics (:1095-1110) in one pass.
//   masked rows:  l = (x - y[n/cb])^2 * scale ; dx = 2*(x-y)*scale*gscale ; kept rows: dx = 0
// stats (double): [0..D) sum x, [D..2D) sum x^2, [2D..3D) sum y, [3D..4D) sum y^2, [4D] loss sum,
// [4D+1] masked-row count.
// One wave per (clip, frame): the target row is read once and stays in registers for the cb clones of the clip, whose rows
// are fetched four at a time before any of them is used (a row-after-row loop with the mask test in front of each load was
// latency-bound at 0.1 of the HBM rate).  Lane-owned columns accumulate over the wave's frames; the D2V_WAVES waves of a
// workgroup combine in a fixed order through LDS, then one double atomic per column and statistic.
#define D2V_WAVES 8
template <typename T, bool ROWS>
__global__ __launch_bounds__(64 * D2V_WAVES, 4) void d2v_loss_kernel(const T* __restrict__ x, const float* __restrict__ y,
                                                                     const unsigned char* __restrict__ mask, T* __restrict__ dx,
                                                                     float* __restrict__ loss_rows, double* __restrict__ stats,
                                                                     int N, int Tn, int D, int cb, float scale, float gscale) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: row addresses stay in SGPRs)
  const int groups = (N + cb - 1) / cb;
  const long long pairs = (long long)groups * Tn;
  const int nch = (D + 511) / 512;
  // The 512-column chunks of a row go to neighbouring waves of the workgroup when their number divides the wave count (1, 2,
  // 4 chunks), so that whole rows stream at any time; one chunk after the other (every wave on the same one) read 1 KB out
  // of every 2 KB and reached half the rate.
  const bool par = nch <= 4 && D2V_WAVES % nch == 0;
  const int wpc = par ? D2V_WAVES / nch : D2V_WAVES;         // waves per chunk = (clip, frame) pairs per workgroup and round
  const int slot = par ? wid / nch : wid;
  const long long nw = (long long)gridDim.x * wpc;
  const float g2 = 2.f * scale * gscale;
  float ls_tot = 0.f, cnt_tot = 0.f;
  __shared__ float red[4][4][8][64];
  for (int cc = 0; cc < (par ? 1 : nch); ++cc) {
    const int c = par ? wid % nch : cc;
    const int rc = par ? c : 0;
    const int col_ = (c * 64 + lane) * 8;
    const bool colok = col_ < D;
    const int col = colok ? col_ : D - 8;      // lanes past the last column read a valid one and contribute / store nothing
    float sx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sxx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float sy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, syy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ls = 0.f, cnt = 0.f;
    for (long long p = (long long)blockIdx.x * wpc + slot; p < pairs; p += nw) {
      const int gi = (int)(p / Tn);
      const int t = (int)(p - (long long)gi * Tn);
      float yv[8];
      Vec8<float>::load(y + ((long long)gi * Tn + t) * D + col, yv);
      const int n0 = gi * cb;
      const int ncl = min(cb, N - n0);                         // clones of this clip that exist (>= 1)
      const long long stride = (long long)Tn * D;              // one clone to the next, same frame
      for (int c0 = 0; c0 < ncl; c0 += 4) {
        const long long base = ((long long)(n0 + c0) * Tn + t) * D + col;
        const int nv = min(4, ncl - c0);
        bool m[4];
        float xv[4][8];
        // (branch-free on the mask: the four mask bytes and the four rows are in flight together; a kept row goes through
        //  the same arithmetic with its values selected to zero)
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = mask[(long long)(n0 + c0 + min(u, nv - 1)) * Tn + t] != 0 && u < nv;
#pragma unroll
        for (int u = 0; u < 4; ++u) Vec8<T>::load(x + base + min(u, nv - 1) * stride, xv[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float o[8], l8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xm = m[u] ? xv[u][j] : 0.f, ym = m[u] ? yv[j] : 0.f;
            const float d = xm - ym;
            l8[j] = d * d * scale;
            o[j] = d * g2;
            sx[j] += xm; sxx[j] += xm * xm;
            sy[j] += ym; syy[j] += ym * ym;
          }
          ls += ((l8[0] + l8[1]) + (l8[2] + l8[3])) + ((l8[4] + l8[5]) + (l8[6] + l8[7]));   // (a tree: one chain through all
          if (c == 0 && lane == 0 && m[u]) cnt += 1.f;                                      //  32 values kept them all live)
          // (pins the order: without it the compiler interleaves the four rows' updates of the 32 column sums and keeps all
          //  64 selected values live -- 166 registers instead of 102)
#pragma unroll
          for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(sx[j]), "+v"(sxx[j]), "+v"(sy[j]), "+v"(syy[j]));
          if (u < nv && colok) {
            if (ROWS) Vec8<float>::store(loss_rows + base + u * stride, l8);
            if (dx) Vec8<T>::store(dx + base + u * stride, o);
          }
        }
      }
    }
    for (int wv = 0; wv < D2V_WAVES; ++wv) {
      __syncthreads();
      if (wid == wv && colok) {
        const bool first = par ? wv < nch : wv == 0;         // the first wave of each chunk writes, the others add, in order
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (first) {
            red[rc][0][j][lane] = sx[j]; red[rc][1][j][lane] = sxx[j]; red[rc][2][j][lane] = sy[j]; red[rc][3][j][lane] = syy[j];
          } else {
            red[rc][0][j][lane] += sx[j]; red[rc][1][j][lane] += sxx[j]; red[rc][2][j][lane] += sy[j]; red[rc][3][j][lane] += syy[j];
          }
        }
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (par ? nch : 1) * 2048; e += 64 * D2V_WAVES) {
      const int ch = e >> 11, which = (e >> 9) & 3, j = (e >> 6) & 7, ln = e & 63;
      const int gc = (par ? ch : cc) * 512 + ln * 8 + j;
      if (gc < D) atomicAdd(stats + which * D + gc, (double)red[ch][which][j][ln]);
    }
    ls_tot += colok ? ls : 0.f;
    cnt_tot += cnt;
  }
  // loss and count: ONE double atomic per workgroup each (same-address atomics are serialised by the L2 at tens of
  // nanoseconds apiece)
  __shared__ float red2[2][D2V_WAVES];
  ls_tot = wave_sum(ls_tot);
  cnt_tot = wave_sum(cnt_tot);
  if (lane == 0) { red2[0][wid] = ls_tot; red2[1][wid] = cnt_tot; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double l = 0.0;
    float cn = 0.f;
    for (int w = 0; w < D2V_WAVES; ++w) { l += (double)red2[0][w]; cn += red2[1][w]; }
    atomicAdd(stats + 4 * D, l);
    if (cn != 0.f) atomicAdd(stats + 4 * D + 1, (double)cn);
  }
}

// ---------------------------------------------------------------------------------- finetune path
// sigmoid focal loss on logits, summed (nn/utils.py:971-1010 as called by criterions.py:243-248), with its gradient:
//   p = sigmoid(x), ce = -[t log p + (1-t) log(1-p)], p_t = p t + (1-p)(1-t), L = alpha_t ce (1-p_t)^gamma
//   dL/dx = alpha_t [ (p - t)(1-p_t)^gamma - gamma ce (1-p_t)^(gamma-1) p (1-p) (2t - 1) ]
// targets are floats (target mixup makes them soft).  logits / dlogits are (M, ld) with C valid columns; padded
// columns of dlogits are written as zero.
template <typename T>
__global__ __launch_bounds__(256) void focal_loss_kernel(const T* __restrict__ logits, const float* __restrict__ targets,
                                                         T* __restrict__ dlogits, double* __restrict__ loss_sum, long long M,
                                                         int C, int ld, float alpha, float gamma, float gscale) {
  const long long total = M * ld;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld);
    const long long m = i / ld;
    float d = 0.f;
    if (c < C) {
      const float x = Cvt<T>::to(logits[i]);
      const float t = targets[m * C + c];
      const float pr = 1.f / (1.f + __expf(-x));
      // numerically stable BCE-with-logits: max(x,0) - x t + log(1 + exp(-|x|))
      const float ce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
      const float p_t = pr * t + (1.f - pr) * (1.f - t);
      const float om = 1.f - p_t;
      const float f = powf(om, gamma);
      const float a_t = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
      acc += a_t * ce * f;
      const float fm1 = gamma == 2.f ? om : powf(om, gamma - 1.f);
      d = a_t * ((pr - t) * f - gamma * ce * fm1 * pr * (1.f - pr) * (2.f * t - 1.f)) * gscale;
    }
    if (dlogits) dlogits[i] = Cvt<T>::from(d);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss_sum, (double)acc);
}

// out = scale * sum_l x_l   (the finetune head's input: mean of the last average_top_k_layers layer results,
// wav2vec2.py:448-466); elementwise over n values, n % 8 == 0
template <typename T>
__global__ __launch_bounds__(256) void mean_layers_kernel(PtrList32 layers, T* __restrict__ out, int L, long long n, float scale) {
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      float v[8];
      Vec8<T>::load((const T*)layers.p[l] + i * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= scale;
    Vec8<T>::store(out + i * 8, acc);
  }
}

// out = a + drop(b + c): a, c optional.  Forward: x1 + post_mlp_dropout(t) when t must stay a result of its own
// (layer results, modules.py:328-337); backward: the gradient arriving at t through the Mlp's own output dropout.
// The dropout index is the element's linear index (same convention as the GEMM / LayerNorm epilogues).
template <typename T>
__global__ __launch_bounds__(256) void add_drop_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                       const T* __restrict__ c, T* __restrict__ out, long long n, Drop d) {
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float vb[8], va[8];
    Vec8<T>::load(b + i * 8, vb);
    if (c) {
      float vc[8];
      Vec8<T>::load(c + i * 8, vc);
#pragma unroll
      for (int j = 0; j < 8; ++j) vb[j] += vc[j];
    }
    if (d.thresh) {
      drop_apply_n<8>(d, (unsigned long long)i * 8, vb);
    }
    if (a) {
      Vec8<T>::load(a + i * 8, va);
#pragma unroll
      for (int j = 0; j < 8; ++j) vb[j] += va[j];
    }
    Vec8<T>::store(out + i * 8, vb);
  }
}

// Finetune train-mode input masking (base.py:457-485, apply_mask): frames with mask_t[b,t] != 0 are replaced by
// N(0, noise_std) (or zero: encoder_zero_mask), then channels with mask_c[b,c] != 0 are zeroed for every frame.
// mode 1 = the backward of the same map (masked positions pass no gradient, nothing is replaced).
// Noise: Box-Muller on two counter hashes of (seed, element index) -- same distribution as the reference's
// normal_(), not torch's Philox stream.
template <typename T>
__global__ __launch_bounds__(256) void mask_frames_kernel(const T* __restrict__ x, const unsigned char* __restrict__ mask_t,
                                                          const unsigned char* __restrict__ mask_c, T* __restrict__ out,
                                                          int B, int Tn, int D, float noise_std, unsigned seed, int mode) {
  const long long nv = (long long)B * Tn * D / 8;
  const int dv = D / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / dv;
    const int col = (int)(i - row * dv) * 8;
    const int b = (int)(row / Tn);
    float v[8];
    const bool mt = mask_t && mask_t[row];
    if (mt) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = 0.f;
        if (mode == 0 && noise_std > 0.f) {
          const unsigned long long e = (unsigned long long)(i * 8 + j);
          const float u1 = ((float)drop_hash(seed, 2 * e) + 1.f) * 2.3283064365386963e-10f;
          const float u2 = (float)drop_hash(seed, 2 * e + 1) * 2.3283064365386963e-10f;
          z = noise_std * sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
        }
        v[j] = z;
      }
    } else {
      Vec8<T>::load(x + i * 8, v);
    }
    if (mask_c) {
      const unsigned long long mc = *(const unsigned long long*)(mask_c + (long long)b * D + col);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if ((mc >> (8 * j)) & 0xFFull) v[j] = 0.f;
    }
    Vec8<T>::store(out + i * 8, v);
  }
}

// ---------------------------------------------------------------------------------- optimizer-side
// EMA teacher (fairseq EMAModule.step): e = e*d + (1-d)*p  (fp32), 4 floats per thread
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ p, long long n,
                                                  float d) {
  const long long nv = n / 4;
  const float om = 1.f - d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    f32x4 a = *(const f32x4*)(e + i * 4), b = *(const f32x4*)(p + i * 4);
    a = a * d + b * om;
    *(f32x4*)(e + i * 4) = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = nv * 4 + threadIdx.x;
    e[i] = e[i] * d + p[i] * om;
  }
}

// fairseq Adam (decoupled weight decay applied before the step), flat fp32 buffers
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                   float b1, float b2, float eps, float wd_, float step_size,
                                                   float gscale_, const unsigned char* __restrict__ decay_mask,
                                                   const float* __restrict__ gscale_dev,
                                                   const float* __restrict__ lr_scale) {
  const float gscale = gscale_ * (gscale_dev ? *gscale_dev : 1.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float wd = (!decay_mask || decay_mask[i >> 6]) ? wd_ : 0.f;
    const float ls = lr_scale ? lr_scale[i >> 6] : 1.f;      // per-parameter-group lr_scale (wav2vec2.py:214-236)
    const float gr = g[i] * gscale;
    const float mi = m[i] * b1 + gr * (1.f - b1);
    const float vi = v[i] * b2 + gr * gr * (1.f - b2);
    float pi = p[i];
    pi -= wd * (lr * ls) * pi;
    pi -= (step_size * ls) * mi / (sqrtf(vi) + eps);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

// sum of squares of a flat fp32 buffer (gradient norm), double accumulator
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, double* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (double)(red[0] + red[1] + red[2] + red[3]));
}

// ---------------------------------------------------------------------------------- weight packing
// Source: reference parameter layout w[(g*N + n)][ci][j]  (C_out, Cg, k), fp32.
// mode 0: out[g][n][j*Cgp + ci]            forward operand (K contiguous), Cgp >= Cg zero padded
// mode 1: out[g][ci][j*Np + n] = w[g,n,ci,k-1-j]   stride-1 dgrad-as-conv operand (flipped taps)
// mode 2: out[j*Cgp + ci][n]               (G==1) dA = dY * Wmat operand, rows kk, N contiguous
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int G, int N,
                                                          int Cg, int k, int Cgp, int Np, int mode, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float val = 0.f;
    if (mode == 0) {
      const int ci = (int)(i % Cgp);
      const int j = (int)((i / Cgp) % k);
      const long long gn = i / ((long long)Cgp * k);
      if (ci < Cg) val = w[(gn * Cg + ci) * k + j];
    } else if (mode == 1) {
      const int n = (int)(i % Np);
      const int j = (int)((i / Np) % k);
      const int ci = (int)((i / ((long long)Np * k)) % Cg);
      const int g = (int)(i / ((long long)Np * k * Cg));
      if (n < N) val = w[(((long long)g * N + n) * Cg + ci) * k + (k - 1 - j)];
    } else {
      const int n = (int)(i % Np);
      const int ci = (int)((i / Np) % Cgp);
      const int j = (int)(i / ((long long)Np * Cgp));
      if (n < N && ci < Cg) val = w[((long long)n * Cg + ci) * k + j];
    }
    out[i] = Cvt<T>::from(val);
  }
}

// All operand images of one parameter set in ONE launch: desc holds 10 int64 per weight (src, dst, G, N, Cg, k, Cgp,
// Np, mode, total); blockIdx.y picks the weight.  (658 separate ~12 us launches per optimizer step were launch bound.)
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const long long* __restrict__ desc) {
  const long long* d = desc + (long long)blockIdx.y * 10;
  const float* __restrict__ w = (const float*)d[0];
  T* __restrict__ out = (T*)d[1];
  const int G = (int)d[2], N = (int)d[3], Cg = (int)d[4], k = (int)d[5], Cgp = (int)d[6], Np = (int)d[7], mode = (int)d[8];
  const long long total = d[9];
  (void)G;
  if (mode == 0 && k == 1 && Cgp == Cg && total % 8 == 0) {
    // Linear forward operand: the image is the parameter itself, cast -- 8 elements (32 B in, 16 B out) per thread
    const long long nv = total / 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
      const f32x4 a = *(const f32x4*)(w + i * 8), b = *(const f32x4*)(w + i * 8 + 4);
      const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      Vec8<T>::store(out + i * 8, v);
    }
    return;
  }
  if (mode == 1 && k == 1 && Cg % 64 == 0 && N % 64 == 0 && Np == N && d[2] == 1) {
    // Linear transpose out[ci][n] = w[n][ci]: 64 x 64 tiles through LDS, 16-B loads and 8 values along n per store --
    // both sides move whole 128-B+ segments
    __shared__ float tile64[64][65];
    const int tiles_c = Cg / 64, tiles_n = N / 64;
    for (int t = blockIdx.x; t < tiles_c * tiles_n; t += gridDim.x) {
      const int tn = t / tiles_c, tc = t - tn * tiles_c;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {                       // 64 rows (n) x 16 float4 (ci)
        const int c = threadIdx.x + 256 * u, r = c >> 4, q = c & 15;
        const f32x4 a = *(const f32x4*)(w + (long long)(tn * 64 + r) * Cg + tc * 64 + q * 4);
        tile64[r][q * 4 + 0] = a[0]; tile64[r][q * 4 + 1] = a[1]; tile64[r][q * 4 + 2] = a[2]; tile64[r][q * 4 + 3] = a[3];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {                       // 64 rows (ci) x 8 chunks of 8 n
        const int c = threadIdx.x + 256 * u, r = c >> 3, q = c & 7;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile64[q * 8 + j][r];
        Vec8<T>::store(out + (long long)(tc * 64 + r) * N + tn * 64 + q * 8, v);
      }
    }
    return;
  }
  if (mode == 1 && k == 1 && Cg % 32 == 0 && N % 32 == 0 && Np == N && d[2] == 1) {
    // Linear transpose out[ci][n] = w[n][ci]: 32 x 32 tiles through LDS, coalesced on both sides
    __shared__ float tile[32][33];
    const int tiles_c = Cg / 32, tiles_n = N / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int t = blockIdx.x; t < tiles_c * tiles_n; t += gridDim.x) {
      const int tn = t / tiles_c, tc = t - tn * tiles_c;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 32; r += 8) tile[ty + r][tx] = w[(long long)(tn * 32 + ty + r) * Cg + tc * 32 + tx];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 32; r += 8) out[(long long)(tc * 32 + ty + r) * N + tn * 32 + tx] = Cvt<T>::from(tile[tx][ty + r]);
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float val = 0.f;
    if (mode == 0) {
      const int ci = (int)(i % Cgp);
      const int j = (int)((i / Cgp) % k);
      const long long gn = i / ((long long)Cgp * k);
      if (ci < Cg) val = w[(gn * Cg + ci) * k + j];
    } else if (mode == 1) {
      const int n = (int)(i % Np);
      const int j = (int)((i / Np) % k);
      const int ci = (int)((i / ((long long)Np * k)) % Cg);
      const int g = (int)(i / ((long long)Np * k * Cg));
      if (n < N) val = w[(((long long)g * N + n) * Cg + ci) * k + (k - 1 - j)];
    } else {
      const int n = (int)(i % Np);
      const int ci = (int)((i / Np) % Cgp);
      const int j = (int)(i / ((long long)Np * Cgp));
      if (n < N && ci < Cg) val = w[((long long)n * Cg + ci) * k + j];
    }
    out[i] = Cvt<T>::from(val);
  }
}

// gw[(g*N+n)][ci][j] += packed[g][n][j*Cgp + ci]     (fp32 both)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(float* __restrict__ packed, float* __restrict__ gw,
                                                           int Cg, int k, int Cgp, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % k);
    const int ci = (int)((i / k) % Cg);
    const long long gn = i / ((long long)k * Cg);
    const long long src = (gn * k + j) * Cgp + ci;
    gw[i] += packed[src];
    packed[src] = 0.f;          // leave the packed accumulator clear for the next micro-batch (saves a fill launch per use;
                                // the tap-padding entries ci >= Cg never receive anything: their A operand is zero)
  }
}

// col2im for the strided dense convs' input gradient:
//   dX[b,u,ci] = sum_{t,j : t*s - p + j = u} dA[(b,t)][j*C + ci]
// One input frame (b, u) per group of C/8 lanes (16 bytes of channels each; a whole wave at C = 512, four frames per wave
// at C = 128): which taps reach the frame (j = j0, j0 + s, ... with t = t0, t0 - 1, ...) is three 32-bit divisions per
// frame, two taps in flight.  (The element-per-thread form spent two 64-bit and 2k 32-bit divisions per 16 bytes.)
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dA, T* __restrict__ dX, int B, int T_in,
                                                     int T_out, int C, int k, int s, int p, long long lda,
                                                     long long ldo) {
  const int vecs = C / 8;
  const int lanes = vecs >= 64 ? 64 : (vecs > 32 ? 64 : vecs > 16 ? 32 : vecs > 8 ? 16 : 8);   // lanes per frame
  const int fpw = 64 / lanes;                                                                    // frames per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / lanes, l0 = lane - sub * lanes;
  const int rows = B * T_in;
  const int nw = gridDim.x * 4 * fpw;
  for (int bu = (blockIdx.x * 4 + (threadIdx.x >> 6)) * fpw + sub; bu < rows; bu += nw) {
    const int b = bu / T_in, u = bu - b * T_in;
    const int j0 = (u + p) % s, t0 = (u + p - j0) / s;        // u + p >= 0
    const T* const arow = dA + (long long)b * T_out * lda;
    for (int v = l0; v < vecs; v += lanes) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int j = j0, t = t0; j < k && t >= 0; j += 2 * s, t -= 2) {
        const bool ok0 = t < T_out, ok1 = j + s < k && t - 1 >= 0 && t - 1 < T_out;
        const int ta = ok0 ? t : 0, tb = ok1 ? t - 1 : 0;
        const int ja = ok0 ? j : 0, jb = ok1 ? j + s : 0;
        float x0[8], x1[8];
        Vec8<T>::load(arow + (long long)ta * lda + (long long)ja * C + v * 8, x0);
        Vec8<T>::load(arow + (long long)tb * lda + (long long)jb * C + v * 8, x1);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += (ok0 ? x0[q] : 0.f) + (ok1 ? x1[q] : 0.f);
      }
      Vec8<T>::store(dX + (long long)bu * ldo + v * 8, acc);
    }
  }
}

// 2-D cast/copy with strides: out[m][0..C) (ldo) = in[m][0..C) (ldi)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast2d_kernel(const TI* __restrict__ in, TO* __restrict__ out, long long M, int C,
                                                     long long ldi, long long ldo) {
  const int vecs = C / 4;
  const long long total = M * vecs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vecs);
    const long long m = i / vecs;
    Vec4<TO>::store(out + m * ldo + v * 4, Vec4<TI>::load(in + m * ldi + v * 4));
  }
}

// =============================================================================================
// (round 5: the in-order-front launch form that pays for the LayerNorms does not for these kernels -- the cap at 65 536 workgroups changes
// nothing on the step, at 1 M it costs 1.5 %)
static inline int grid_for(long long total, int cap = 8192) {
  long long g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
  if (dtype == A2V_BF16) { CALL_BF16; } else if (dtype == A2V_F32) { CALL_F32; } else return A2V_ERR_DTYPE;

extern "C" int a2v_gather_tokens(int dtype, const void* a, const void* b, const int* ids, void* out, int N, int K, int D,
                                 int Ta, int cb, float drop_p, unsigned drop_seed, hipStream_t st) {
  if (N <= 0 || K <= 0 || D % 8 || cb <= 0 || drop_p < 0.f || drop_p >= 1.f) return A2V_ERR_ARG;
  const Drop drop = make_drop(drop_p, drop_seed);
  const int g = grid_for((long long)N * K * (D / 8));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gather_tokens_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, ids, (bf16*)out, N, K, D, Ta, cb, drop),
             hipLaunchKernelGGL(gather_tokens_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, ids, (float*)out, N, K, D, Ta, cb, drop));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_scatter_tokens(int dtype, const void* dtok, const int* restore, void* dpos, int N, int K, int D, int Tn,
                                  hipStream_t st) {
  if (N <= 0 || K <= 0 || D % 8) return A2V_ERR_ARG;
  const int g = grid_for((long long)N * Tn * (D / 8));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(scatter_tokens_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)dtok, restore, (bf16*)dpos, N, K, D, Tn),
             hipLaunchKernelGGL(scatter_tokens_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dtok, restore, (float*)dpos, N, K, D, Tn));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_clone_reduce(int dtype, const void* dxm, const void* dtok, const int* restore, void* dfeat, int B, int cb,
                                int K, int D, int Tn, hipStream_t st) {
  if (B <= 0 || K <= 0 || D % 8 || cb <= 0) return A2V_ERR_ARG;
  const int g = grid_for((long long)B * Tn * (D / 8));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(clone_reduce_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)dxm, (const bf16*)dtok, restore, (bf16*)dfeat, B, cb, K, D, Tn),
             hipLaunchKernelGGL(clone_reduce_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dxm, (const float*)dtok, restore, (float*)dfeat, B, cb, K, D, Tn));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

static int decoder_input_launch(int dtype, const void* x, const float* noise, const int* restore, void* out, int N, int K,
                                int D, int Tn, float drop_p, unsigned drop_seed, float noise_std, unsigned noise_seed,
                                hipStream_t st) {
  const Drop drop = make_drop(drop_p, drop_seed);
  if ((long long)N * Tn > 0x7FFFFFFFLL) return A2V_ERR_ARG;
  const int g = grid_for((long long)N * Tn * 64);                 // one wave per output row
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(decoder_input_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)x, noise, restore, (bf16*)out, N, K, D, Tn, drop, noise_std, noise_seed),
             hipLaunchKernelGGL(decoder_input_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, noise, restore, (float*)out, N, K, D, Tn, drop, noise_std, noise_seed));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_decoder_input(int dtype, const void* x, const float* noise, const int* restore, void* out, int N, int K,
                                 int D, int Tn, float drop_p, unsigned drop_seed, hipStream_t st) {
  if (N <= 0 || K <= 0 || K > Tn || D % 8 || drop_p < 0.f || drop_p >= 1.f || !x || !restore || !out || (!noise && K < Tn))
    return A2V_ERR_ARG;
  return decoder_input_launch(dtype, x, noise, restore, out, N, K, D, Tn, drop_p, drop_seed, 0.f, 0u, st);
}

extern "C" int a2v_decoder_input_rng(int dtype, const void* x, const int* restore, void* out, int N, int K, int D, int Tn,
                                     float noise_std, unsigned noise_seed, float drop_p, unsigned drop_seed, hipStream_t st) {
  if (N <= 0 || K <= 0 || K > Tn || D % 8 || drop_p < 0.f || drop_p >= 1.f || !x || !restore || !out || !(noise_std >= 0.f))
    return A2V_ERR_ARG;
  return decoder_input_launch(dtype, x, nullptr, restore, out, N, K, D, Tn, drop_p, drop_seed, noise_std, noise_seed, st);
}

extern "C" int a2v_d2v_loss(int dtype, const void* x, const float* y, const unsigned char* mask, void* dx, float* loss_rows,
                            double* stats, int N, int Tn, int D, int cb, float scale, float gscale, hipStream_t st) {
  if (N <= 0 || Tn <= 0 || D % 8 || cb <= 0 || !stats) return A2V_ERR_ARG;
  // one (clip, frame) pair per wave and round; 2 workgroups of 8 waves per CU (103 registers)
  const long long pairs = (long long)((N + cb - 1) / cb) * Tn;
  const int nch = (D + 511) / 512;
  const int wpc = (nch <= 4 && D2V_WAVES % nch == 0) ? D2V_WAVES / nch : D2V_WAVES;
  long long gl = (pairs + wpc - 1) / wpc;
  const int g = (int)(gl < 1 ? 1 : gl > 512 ? 512 : gl);
#define D2V_LAUNCH(T_, R_) hipLaunchKernelGGL((d2v_loss_kernel<T_, R_>), dim3(g), dim3(64 * D2V_WAVES), 0, st, (const T_*)x, y, mask, (T_*)dx, loss_rows, stats, N, Tn, D, cb, scale, gscale)
  if (loss_rows) { DISPATCH_T(dtype, D2V_LAUNCH(bf16, true), D2V_LAUNCH(float, true)); }
  else { DISPATCH_T(dtype, D2V_LAUNCH(bf16, false), D2V_LAUNCH(float, false)); }
#undef D2V_LAUNCH
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_focal_loss(int dtype, const void* logits, const float* targets, void* dlogits, double* loss_sum, long long M,
                              int C, int ld, float alpha, float gamma, float gscale, hipStream_t st) {
  if (M <= 0 || C <= 0 || ld < C || !logits || !targets || !loss_sum) return A2V_ERR_ARG;
  const int g = grid_for(M * ld);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(focal_loss_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)logits, targets, (bf16*)dlogits, loss_sum, M, C, ld, alpha, gamma, gscale),
             hipLaunchKernelGGL(focal_loss_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)logits, targets, (float*)dlogits, loss_sum, M, C, ld, alpha, gamma, gscale));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_mean_layers(int dtype, const void* const* layers, int L, void* out, long long n, float scale, hipStream_t st) {
  if (L <= 0 || L > 32 || n <= 0 || n % 8 || !out) return A2V_ERR_ARG;
  PtrList32 pl;
  for (int i = 0; i < 32; ++i) pl.p[i] = i < L ? layers[i] : nullptr;
  const int g = grid_for(n / 8);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(mean_layers_kernel<bf16>, dim3(g), dim3(256), 0, st, pl, (bf16*)out, L, n, scale),
             hipLaunchKernelGGL(mean_layers_kernel<float>, dim3(g), dim3(256), 0, st, pl, (float*)out, L, n, scale));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_add_drop(int dtype, const void* a, const void* b, const void* c, void* out, long long n, float drop_p,
                            unsigned drop_seed, hipStream_t st) {
  if (n <= 0 || n % 8 || !b || !out || drop_p < 0.f || drop_p >= 1.f) return A2V_ERR_ARG;
  const Drop d = make_drop(drop_p, drop_seed);
  const int g = grid_for(n / 8);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(add_drop_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)c, (bf16*)out, n, d),
             hipLaunchKernelGGL(add_drop_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)c, (float*)out, n, d));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_mask_frames(int dtype, const void* x, const unsigned char* mask_t, const unsigned char* mask_c, void* out,
                               int B, int T, int D, float noise_std, unsigned seed, int mode, hipStream_t st) {
  if (B <= 0 || T <= 0 || D <= 0 || D % 8 || !x || !out || mode < 0 || mode > 1 || noise_std < 0.f) return A2V_ERR_ARG;
  const long long n = (long long)B * T * D;
  const int g = grid_for(n / 8);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(mask_frames_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)x, mask_t, mask_c, (bf16*)out, B, T, D, noise_std, seed, mode),
             hipLaunchKernelGGL(mask_frames_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, mask_t, mask_c, (float*)out, B, T, D, noise_std, seed, mode));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_ema_update(float* ema, const float* p, long long n, float decay, hipStream_t st) {
  if (n <= 0 || (((uintptr_t)ema | (uintptr_t)p) & 15)) return A2V_ERR_ARG;
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st, ema, p, n, decay);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// decay_mask: one byte per 64 parameters (0 -> weight decay off for that block) or NULL; lr_scale: one float per 64
// parameters (the composite optimizer's per-group lr_scale, e.g. layer_decay in finetuning) or NULL;
// grad_scale_dev: optional device scalar multiplied into grad_scale (clip / sample-size factor).
extern "C" int a2v_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale,
                             const unsigned char* decay_mask, const float* grad_scale_dev, const float* lr_scale,
                             hipStream_t st) {
  if (n <= 0 || step <= 0) return A2V_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     step_size, grad_scale, decay_mask, grad_scale_dev, lr_scale);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_sumsq(const float* x, long long n, double* out, hipStream_t st) {
  if (n <= 0) return A2V_ERR_ARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, st, x, n, out);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_pack_weight(int dtype, const float* w, void* out, int G, int N, int Cg, int k, int Cgp, int Np, int mode,
                               hipStream_t st) {
  if (G <= 0 || N <= 0 || Cg <= 0 || k <= 0 || Cgp < Cg || Np < N || mode < 0 || mode > 2) return A2V_ERR_ARG;
  if (mode == 2 && G != 1) return A2V_ERR_ARG;
  long long total;
  if (mode == 0) total = (long long)G * N * k * Cgp;
  else if (mode == 1) total = (long long)G * Cg * k * Np;
  else total = (long long)k * Cgp * Np;
  const int g = grid_for(total);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(g), dim3(256), 0, st, w, (bf16*)out, G, N, Cg, k, Cgp, Np, mode, total),
             hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(g), dim3(256), 0, st, w, (float*)out, G, N, Cg, k, Cgp, Np, mode, total));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_pack_weight_batch(int dtype, const long long* desc_dev, int n, hipStream_t st) {
  if (!desc_dev || n <= 0) return A2V_ERR_ARG;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_weight_batch_kernel<bf16>, dim3(128, n), dim3(256), 0, st, desc_dev),
             hipLaunchKernelGGL(pack_weight_batch_kernel<float>, dim3(128, n), dim3(256), 0, st, desc_dev));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_unpack_wgrad(float* packed, float* gw, int GN, int Cg, int k, int Cgp, hipStream_t st) {
  if (GN <= 0 || Cg <= 0 || k <= 0 || Cgp < Cg) return A2V_ERR_ARG;
  const long long total = (long long)GN * Cg * k;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for(total)), dim3(256), 0, st, packed, gw, Cg, k, Cgp, total);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_col2im(int dtype, const void* dA, void* dX, int B, int T_in, int T_out, int C, int k, int s, int p,
                          long long lda, long long ldo, hipStream_t st) {
  if (B <= 0 || C % 8 || lda % 8 || ldo % 8 || ldo < C || s <= 0 || p < 0 || T_in <= 0 || T_out <= 0 ||
      (long long)B * T_in > 0x7FFFFFFFLL)
    return A2V_ERR_ARG;
  const int lanes_ = C / 8 > 32 ? 64 : C / 8 > 16 ? 32 : C / 8 > 8 ? 16 : 8;
  const int g = grid_for((long long)B * T_in * lanes_);        // C/8 lanes (rounded up to a power of two) per input frame
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(col2im_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)dA, (bf16*)dX, B, T_in, T_out, C, k, s, p, lda, ldo),
             hipLaunchKernelGGL(col2im_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dA, (float*)dX, B, T_in, T_out, C, k, s, p, lda, ldo));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// in_dtype / out_dtype: A2V_F32 or A2V_BF16
extern "C" int a2v_cast2d(int in_dtype, int out_dtype, const void* in, void* out, long long M, int C, long long ldi,
                          long long ldo, hipStream_t st) {
  if (M <= 0 || C <= 0 || C % 4 || ldi % 4 || ldo % 4) return A2V_ERR_ARG;
  const int g = grid_for(M * (C / 4));
  if (in_dtype == A2V_F32 && out_dtype == A2V_BF16)
    hipLaunchKernelGGL((cast2d_kernel<float, bf16>), dim3(g), dim3(256), 0, st, (const float*)in, (bf16*)out, M, C, ldi, ldo);
  else if (in_dtype == A2V_F32 && out_dtype == A2V_F32)
    hipLaunchKernelGGL((cast2d_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)in, (float*)out, M, C, ldi, ldo);
  else if (in_dtype == A2V_BF16 && out_dtype == A2V_F32)
    hipLaunchKernelGGL((cast2d_kernel<bf16, float>), dim3(g), dim3(256), 0, st, (const bf16*)in, (float*)out, M, C, ldi, ldo);
  else if (in_dtype == A2V_BF16 && out_dtype == A2V_BF16)
    hipLaunchKernelGGL((cast2d_kernel<bf16, bf16>), dim3(g), dim3(256), 0, st, (const bf16*)in, (bf16*)out, M, C, ldi, ldo);
  else
    return A2V_ERR_DTYPE;
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// materialise the dropout keep-mask of linear indices [0, n) for a (p, seed) site: test / debugging aid
__global__ void dropout_mask_kernel(unsigned char* out, long long n, Drop d) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = drop_keep(d, (unsigned long long)i) ? 1 : 0;
}
extern "C" int a2v_dropout_mask(unsigned char* keep, long long n, float drop_p, unsigned drop_seed, hipStream_t st) {
  if (n <= 0 || !(drop_p > 0.f && drop_p < 1.f)) return A2V_ERR_ARG;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, st, keep, n, make_drop(drop_p, drop_seed));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// keep-mask of the attention-probability dropout (2-D pair hash, a2v_common.h): rows = N*H*S query rows, cols = S keys
__global__ void dropout_mask2d_kernel(unsigned char* out, long long rows, int cols, Drop d) {
  const long long n = rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    out[i] = drop_keep2(d, drop_row_term(d.seed, (unsigned)r), (unsigned)(i - r * cols)) ? 1 : 0;
  }
}
extern "C" int a2v_dropout_mask2d(unsigned char* keep, long long rows, int cols, float drop_p, unsigned drop_seed, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || !(drop_p > 0.f && drop_p < 1.f)) return A2V_ERR_ARG;
  hipLaunchKernelGGL(dropout_mask2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, st, keep, rows, cols, make_drop(drop_p, drop_seed));
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_abi_version(void) { return 14; }
