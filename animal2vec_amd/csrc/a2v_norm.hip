// HBM-bound normalisation kernels (channels-last rows): LayerNorm fwd/bwd (optional residual add
// in, optional exact-GELU out, optional column sums out), the instance-norm teacher-target builder
// (reference data2vec2.py:1023-1066) and column sums.  One wave (64 lanes) owns one row; each lane
// keeps its 8-element chunks in registers (16-B loads for bf16, 2x16-B for fp32); statistics in fp32.
#include "a2v_common.h"

#define LN_MAXCHUNK 4     // C <= 4*512 = 2048
#define LN_ROWS_PER_WAVE 8

// R rows per wave are kept in flight together: short tensors (a transformer block's 13-16 k rows) are latency
// bound with one row per wave (60 us -> 25 us with R = 2); long ones prefer R = 1 (occupancy).
template <typename T, int NCH, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            T* __restrict__ sum_out, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, const T* __restrict__ post,
                                                            long long M, int C, long long ldx, long long ldy, float eps,
                                                            int act, Drop drop) {
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  float v[R][NCH][8];
  // issue every load of all R rows first
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long long row = row0 + r < M ? row0 + r : M - 1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) {
        Vec8<T>::load(x + row * ldx + col, v[r][c]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[r][c][i] = 0.f;
      }
    }
  }
  if (res) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = row0 + r < M ? row0 + r : M - 1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < C) {
          float q[8];
          Vec8<T>::load(res + row * ldx + col, q);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[r][c][i] += q[i];
        }
      }
    }
  }
  float gm[NCH][8], bt[NCH][8];
  if (gamma) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) { Vec8<float>::load(gamma + col, gm[c]); Vec8<float>::load(beta + col, bt[c]); }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long long row = row0 + r;
    if (row >= M) break;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) {
        if (sum_out) Vec8<T>::store(sum_out + row * ldx + col, v[r][c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[r][c][i];
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[r][c][i] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float t = (v[r][c][i] - mean) * rstd;
          if (gamma) t = t * gm[c][i] + bt[c][i];
          if (act == 1) t = gelu_f(t);
          if (drop.thresh) t = drop_apply(drop, (unsigned long long)row * C + col + i, t);
          o[i] = t;
        }
        if (post) {
          float pr[8];
          Vec8<T>::load(post + row * ldy + col, pr);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += pr[i];
        }
        Vec8<T>::store(y + row * ldy + col, o);
      }
    }
  }
}

// Backward of y = act(LN(x)*gamma+beta).  dx; dgamma/dbeta and (optionally) column sums of dx are
// accumulated with one atomic per column per workgroup (each lane owns fixed columns).
// ----------------------------------------------------------------------------------------------
// Column sums (dgamma / dbeta / bias gradients) leave every workgroup as atomics onto the SAME few KB: thousands of
// workgroups adding into one row run ~14x below the chip's atomic rate (MI355X_MICROARCH.md, Global float
// atomics).  For large grids workgroup b therefore adds into copy b % SPREAD of a library-owned, self-cleaning
// scratch ([SPREAD][3][SPREAD_C] fp32) and a small second kernel folds the copies into the caller's destination
// (one uncontended atomic per column) and zeroes them again.
#define SPREAD 64
#define SPREAD_C 4096
// one self-cleaning copy set per stream: A2V_WS_SPREAD of the stream's workspace (zero when handed over, zeroed again by the fold)
static_assert(SPREAD * 3 * SPREAD_C * 4 == A2V_WS_SPREAD_BYTES, "workspace layout");
static float* spread_ws(hipStream_t st) {
  char* const ws = a2v_stream_ws(st);
  return ws ? (float*)(ws + A2V_WS_SPREAD) : nullptr;          // no workspace left for this stream: direct atomics
}
__global__ __launch_bounds__(256) void spread_fold_kernel(float* __restrict__ ws, float* __restrict__ d0,
                                                          float* __restrict__ d1, float* __restrict__ d2, int C) {
  const int which = blockIdx.y;
  float* const dst = which == 0 ? d0 : (which == 1 ? d1 : d2);
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (!dst || c >= C) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < SPREAD; ++k) {
    float* p = ws + ((long long)k * 3 + which) * SPREAD_C + c;
    s += *p;
    *p = 0.f;
  }
  atomicAdd(dst + c, s);
}
static void spread_fold(float* ws, float* d0, float* d1, float* d2, int C, hipStream_t st) {
  hipLaunchKernelGGL(spread_fold_kernel, dim3((C + 255) / 256, 3), dim3(256), 0, st, ws, d0, d1, d2, C);
}

// ACT / DM are template parameters: with run-time flags the GELU' and dropout paths kept ~60 VGPRs alive in every
// instance (190 VGPRs at C = 1024: 2 waves per SIMD for a kernel that only hides latency with occupancy).
template <typename T, int NCH, int RPW, int ACT, int DM>
__global__ __launch_bounds__(256, (ACT == 0 && NCH <= 2 && sizeof(T) == 2) ? 3 : 1) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, T* __restrict__ dx,
                                                            const T* __restrict__ dres, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dxsum,
                                                            long long M, int C, long long ldx, long long lddy,
                                                            Drop drop, T* __restrict__ dx_drop,
                                                            float* __restrict__ spread) {
  constexpr int rows_per_wave = RPW;
  // drop_mode 1: the forward applied dropout AFTER this LN (y = drop(act(LN x))): mask the incoming dy.
  // drop_mode 2: also emit dx_drop = dropout-masked dx (gradient of a branch x -> drop(f(x)) summed into this
  //              LN's input); dxsum then sums dx_drop (the branch's bias gradient).
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float ag[NCH][8], ab[NCH][8], as_[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { ag[c][i] = 0.f; ab[c][i] = 0.f; as_[c][i] = 0.f; }

  const long long row0 = ((long long)blockIdx.x * 4 + wid) * rows_per_wave;
  // bf16: TWO rows' x / dy are requested together (raw, 16 B per chunk) and processed one after the other.  The kernel hides
  // memory latency with nothing but bytes in flight: at C = 1024 it holds 3 waves per SIMD (144-160 VGPRs), one row each =
  // 12 MB on the chip, which is 4 TB/s at ~2-3 us of loaded latency; the second row doubles that.  Its 16 registers are paid by
  // keeping gamma in LDS (read per row, two ds_read_b128 per chunk) instead of 8 NCH registers per lane.  Measured at
  // 53760 x 1024: 92.0 -> 82.2 us (plain), 109.7 -> 98.8 us (dropout mode 2).
  constexpr bool PAIR = sizeof(T) == 2 && ACT == 0 && NCH >= 2;   // (C = 512 with GELU' is VALU-bound: pairing cost it 6 %)
  __shared__ float sgam[PAIR ? LN_MAXCHUNK * 512 : 1];
  float gmr_reg[PAIR ? 1 : NCH][8];
  if constexpr (PAIR) {
    for (int k = threadIdx.x; k < NCH * 512; k += 256) sgam[k] = (gamma && k < C) ? gamma[k] : 1.f;
    __syncthreads();
  } else {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) gmr_reg[c][i] = 1.f;
      if (gamma && col < C) Vec8<float>::load(gamma + col, gmr_reg[c]);
    }
  }
  auto one_row = [&](long long row, const Raw8<T>* xr, const Raw8<T>* dr) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float xh[NCH][8], g[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) {
        float xv[8], dv[8], ga8[8];
        xr[c].unpack(xv);
        dr[c].unpack(dv);
        if constexpr (PAIR) {
          const f32x4 g0 = *(const f32x4*)(sgam + col), g1 = *(const f32x4*)(sgam + col + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { ga8[i] = g0[i]; ga8[4 + i] = g1[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) ga8[i] = gmr_reg[c][i];
        }
        float bt8[8];
        if (ACT == 1 && gamma) Vec8<float>::load(beta + col, bt8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float h = (xv[i] - mean) * rstd;
          const float ga = ga8[i];
          float d = dv[i];
          if (DM == 1) d = drop_apply(drop, (unsigned long long)row * C + col + i, d);
          if (ACT == 1) d *= gelu_grad_f(gamma ? h * ga + bt8[i] : h);
          ag[c][i] += d * h;
          ab[c][i] += d;
          const float dg = d * ga;
          xh[c][i] = h;
          g[c][i] = dg;
          s1 += dg;
          s2 += dg * h;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xh[c][i] = 0.f; g[c][i] = 0.f; }
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rstd * (g[c][i] - s1 - xh[c][i] * s2);
        if (dres) {
          float r[8];
          Vec8<T>::load(dres + row * ldx + col, r);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += r[i];
        }
        Vec8<T>::store(dx + row * ldx + col, o);
        if (DM == 2) {
          drop_apply_n<8>(drop, (unsigned long long)row * C + col, o);
          Vec8<T>::store(dx_drop + row * ldx + col, o);
        }
        if (dxsum) {
#pragma unroll
          for (int i = 0; i < 8; ++i) as_[c][i] += Cvt<T>::to(Cvt<T>::from(o[i]));
        }
      }
    }
  };
  constexpr int STEP = PAIR ? 2 : 1;
  for (int rr = 0; rr < rows_per_wave; rr += STEP) {
    const long long row = row0 + rr;
    if (row >= M) break;
    Raw8<T> xa[NCH], da[NCH], xb[PAIR ? NCH : 1], db[PAIR ? NCH : 1];
    const bool two = PAIR && row + 1 < M;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < C) {
        xa[c].load(x + row * ldx + col);
        da[c].load(dy + row * lddy + col);
        if constexpr (PAIR) {
          const long long rb = two ? row + 1 : row;           // (a last odd row: the second request repeats the first, unused)
          xb[c].load(x + rb * ldx + col);
          db[c].load(dy + rb * lddy + col);
        }
      }
    }
    one_row(row, xa, da);
    __builtin_amdgcn_sched_barrier(0);             // (interleaved, the two rows' temporaries cost a wave of occupancy)
    if constexpr (PAIR) {
      if (two) one_row(row + 1, xb, db);
    }
  }
  // cross-wave reduction of the per-column partials, then one atomic per column
  const int nvec = (dgamma ? 2 : 0) + (dxsum ? 1 : 0);
  for (int which = 0; which < 3; ++which) {
    if (which < 2 && !dgamma) continue;
    if (which == 2 && !dxsum) continue;
    float* dst = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
    if (spread) dst = spread + ((long long)(blockIdx.x % SPREAD) * 3 + which) * SPREAD_C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wid][lane * 8 + i] = which == 0 ? ag[c][i] : (which == 1 ? ab[c][i] : as_[c][i]);
      __syncthreads();
      // 512 columns of this chunk, 256 threads -> 2 each
      for (int k = threadIdx.x; k < 512; k += 256) {
        const int col = c * 512 + k;
        if (col < C) atomicAdd(dst + col, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
      }
    }
  }
  (void)nvec;
}

// ---------------------------------------------------------------------------------- instance norm
// stats over T per (layer, batch, channel): mean and rstd (biased variance, eps) -- two passes over
// the (L2-resident) column strip for accuracy.
struct PtrList { const void* p[32]; };

// One pass with 16-B loads: a workgroup owns 512 channels of one (layer, batch item), its four waves take
// interleaved rows.  The moments are accumulated about the first row's value c (s1 = sum(x - c), s2 = sum((x - c)^2)):
// the same shift in every wave makes the partials additive and keeps var = s2/n - (s1/n)^2 free of the cancellation
// a raw second moment would have.  (The two-pass, 128-B-per-row version read the 524 MB twice at 1 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void instnorm_stats_kernel(PtrList layers, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int B, int Tn, int D,
                                                             float eps) {
  __shared__ float red[2][4][512];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * 8;
  const int b = blockIdx.y, l = blockIdx.z;
  const T* base = (const T*)layers.p[l] + (long long)b * Tn * D;
  const bool ok = c0 < D;
  float c[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { c[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
  if (ok) {
    Vec8<T>::load(base + c0, c);
    for (int t = wid; t < Tn; t += 4) {
      float v[8];
      Vec8<T>::load(base + (long long)t * D + c0, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[i] - c[i]; s1[i] += d; s2[i] += d * d; }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[0][wid][lane * 8 + i] = s1[i]; red[1][wid][lane * 8 + i] = s2[i]; }
  __syncthreads();
  if (wid == 0 && ok) {
    const float inv = 1.f / (float)Tn;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane * 8 + i;
      const float m1 = (red[0][0][k] + red[0][1][k] + red[0][2][k] + red[0][3][k]) * inv;
      const float m2 = (red[1][0][k] + red[1][1][k] + red[1][2][k] + red[1][3][k]) * inv;
      const long long o = ((long long)l * B + b) * D + c0 + i;
      mean[o] = c[i] + m1;
      rstd[o] = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
    }
  }
}

// y[b,t,c] = (1/L) * sum_l (x_l[b,t,c] - mean[l,b,c]) * rstd[l,b,c]   (fp32 out)
// One frame (b, t) per wave and round, 16 bytes of every layer per lane, the layers' loads issued four at a time (the
// element-per-thread form: two 64-bit divisions and one 8-byte load per layer in flight -- 3.1 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void instnorm_apply_kernel(PtrList layers, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ y,
                                                             int L, int B, int Tn, int D) {
  const int lane = threadIdx.x & 63;
  const int rows = B * Tn, vecs = D / 8;
  const int nw = gridDim.x * 4;
  const float inv = 1.f / (float)L;
  for (int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); row < rows; row += nw) {
    const int b = row / Tn;
    for (int v = lane; v < vecs; v += 64) {
      const long long e = (long long)row * D + v * 8;
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int l0 = 0; l0 < L; l0 += 4) {
        float xv[4][8], mu[4][8], rs[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int l = l0 + u < L ? l0 + u : L - 1;
          Vec8<T>::load((const T*)layers.p[l] + e, xv[u]);
          const long long o = ((long long)l * B + b) * D + v * 8;
          Vec8<float>::load(mean + o, mu[u]);
          Vec8<float>::load(rstd + o, rs[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (l0 + u < L) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += (xv[u][i] - mu[u][i]) * rs[u][i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= inv;
      Vec8<float>::store(y + e, acc);
    }
  }
}

// The same average with the statistics out of the inner loop (round 3): y = (1/L) sum_l x_l rstd_l - (1/L) sum_l mean_l rstd_l, and both
// rstd_l[b, c] and the second term are constants of a (clip, column): a wave owns 8 columns x 64 lanes = 512 columns of one clip and
// a range of frames, loads the L x 8 reciprocal deviations of its columns ONCE into registers (128 for 16 layers) and then streams
// frames -- 16 B of every layer per lane, four layers in flight, one FMA per element.  The form above pulled 64 B of mean / rstd
// through L2 for every 16 B of activation (3.4 TB/s).  Needs D % 512 == 0 and L <= 16.
template <typename T>
__global__ __launch_bounds__(256) void instnorm_apply_cols_kernel(PtrList layers, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ y,
                                                                  int L, int B, int Tn, int D, int tchunk) {
  constexpr int LMAX = 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nchunk = D / 512;
  const int ntc = (Tn + 4 * tchunk - 1) / (4 * tchunk);
  int id = blockIdx.x;
  const int tcg = id % ntc; id /= ntc;
  const int cc = id % nchunk;
  const int b = id / nchunk;
  const int t0 = (tcg * 4 + wid) * tchunk;
  const int t1 = t0 + tchunk < Tn ? t0 + tchunk : Tn;
  if (t0 >= Tn) return;
  const int c = cc * 512 + lane * 8;
  const float inv = 1.f / (float)L;
  float rs[LMAX][8], c0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < LMAX; ++l) {
    float mu[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { rs[l][i] = 0.f; mu[i] = 0.f; }
    if (l < L) {
      const long long o = ((long long)l * B + b) * D + c;
      Vec8<float>::load(rstd + o, rs[l]);
      Vec8<float>::load(mean + o, mu);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { rs[l][i] *= inv; c0[i] = fmaf(mu[i], rs[l][i], c0[i]); }
  }
  for (int t = t0; t < t1; ++t) {
    const long long e = ((long long)b * Tn + t) * D + c;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = -c0[i];
#pragma unroll
    for (int l0 = 0; l0 < LMAX; l0 += 4) {
      if (l0 < L) {                                   // uniform
        Raw8<T> xr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xr[u].load((const T*)layers.p[l0 + u < L ? l0 + u : L - 1] + e);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float xv[8];
          xr[u].unpack(xv);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = fmaf(xv[i], rs[l0 + u][i], acc[i]);     // (layers past L carry rs = 0)
        }
      }
    }
    Vec8<float>::store(y + e, acc);
  }
}

// column sums: out[c] += sum_m x[m][c].  One workgroup = 512 columns x rows_per_block rows: every wave reads whole
// 1-KiB row segments (16 B per lane), the four waves take interleaved rows; partial sums meet in LDS and leave as
// one atomic per column into the workgroup's spread copy.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long long M,
                                                     int C, long long ld, int rows_per_block, float* __restrict__ spread) {
  constexpr int V = 16 / sizeof(T);                  // elements per lane
  __shared__ float red[4][64 * V];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * V;
  const long long m0 = (long long)blockIdx.y * rows_per_block;
  const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  float s[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s[i] = 0.f;
  if (c0 < C) {
    for (long long m = m0 + wid; m < m1; m += 4) {
      float v[V];
      if constexpr (V == 8) Vec8<T>::load(x + m * ld + c0, v);
      else { const f32x4 q = *(const f32x4*)(x + m * ld + c0); v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; }
#pragma unroll
      for (int i = 0; i < V; ++i) s[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) red[wid][lane * V + i] = s[i];
  __syncthreads();
  float* dst = spread ? spread + ((long long)((blockIdx.y * gridDim.x + blockIdx.x) % SPREAD) * 3) * SPREAD_C : out;
  for (int k = threadIdx.x; k < 64 * V; k += 256) {
    const int c = blockIdx.x * 64 * V + k;
    if (c < C) atomicAdd(dst + c, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
  }
}

// =============================================================================================
template <typename T>
static int ln_fwd_launch(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* sum_out,
                         float* mean, float* rstd, const void* post, long long M, int C, long long ldx, long long ldy, float eps, int act,
                         Drop drop, hipStream_t st) {
  const int nch = (C + 511) / 512;
  const bool two = M <= 65536;
  dim3 grid((unsigned)((M + (two ? 8 : 4) - 1) / (two ? 8 : 4))), blk(256);
#define LN_F(N)                                                                                                        \
  if (two)                                                                                                             \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, N, 2>), grid, blk, 0, st, (const T*)x, (const T*)res, gamma, beta, (T*)y, \
                       (T*)sum_out, mean, rstd, (const T*)post, M, C, ldx, ldy, eps, act, drop);                       \
  else                                                                                                                 \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, N, 1>), grid, blk, 0, st, (const T*)x, (const T*)res, gamma, beta, (T*)y, \
                       (T*)sum_out, mean, rstd, (const T*)post, M, C, ldx, ldy, eps, act, drop)
  switch (nch) {
    case 1: LN_F(1); break;
    case 2: LN_F(2); break;
    case 3: LN_F(3); break;
    case 4: LN_F(4); break;
    default: return A2V_ERR_ARG;
  }
#undef LN_F
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta,
                                 void* y, void* sum_out, float* mean, float* rstd, const void* post, long long M, int C,
                                 long long ldx, long long ldy, float eps, int act, float drop_p, unsigned drop_seed,
                                 hipStream_t stream) {
  if (M <= 0 || C <= 0 || C % 8 || ldx % 8 || ldy % 8 || C > 512 * LN_MAXCHUNK) return A2V_ERR_ARG;
  if ((gamma == nullptr) != (beta == nullptr)) return A2V_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && ldy != C)) return A2V_ERR_ARG;
  const Drop drop = make_drop(drop_p, drop_seed);
  if (dtype == A2V_BF16) return ln_fwd_launch<bf16>(x, res, gamma, beta, y, sum_out, mean, rstd, post, M, C, ldx, ldy, eps, act, drop, stream);
  if (dtype == A2V_F32) return ln_fwd_launch<float>(x, res, gamma, beta, y, sum_out, mean, rstd, post, M, C, ldx, ldy, eps, act, drop, stream);
  return A2V_ERR_DTYPE;
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, void* dx, const void* dres, float* dgamma, float* dbeta, float* dxsum,
                         long long M, int C, long long ldx, long long lddy, int act, Drop drop, int drop_mode,
                         void* dx_drop, hipStream_t st) {
  const int nch = (C + 511) / 512;
  const int rpw = LN_ROWS_PER_WAVE;     // 4 was measured slower for short tensors (dgamma atomics dominate)
  dim3 grid((unsigned)((M + 4 * rpw - 1) / (4 * rpw))), blk(256);
  float* spread = nullptr;
  if (grid.x >= 16 * SPREAD && (dgamma || dxsum)) {     // short tensors: the fold launch costs more than the contention
    spread = spread_ws(st);
    if (!spread) return A2V_ERR_LAUNCH;
  }
#define LN_B2(N, A, D)                                                                                           \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, N, LN_ROWS_PER_WAVE, A, D>), grid, blk, 0, st, (const T*)dy, (const T*)x, gamma, \
                     beta, mean, rstd, (T*)dx, (const T*)dres, dgamma, dbeta, dxsum, M, C, ldx, lddy, drop, (T*)dx_drop, spread)
#define LN_B1(N, A)                                                                                              \
  { if (drop_mode == 0) LN_B2(N, A, 0); else if (drop_mode == 1) LN_B2(N, A, 1); else LN_B2(N, A, 2); }
#define LN_B(N) { if (act == 1) LN_B1(N, 1) else LN_B1(N, 0) }
  switch (nch) {
    case 1: LN_B(1); break;
    case 2: LN_B(2); break;
    case 3: LN_B(3); break;
    case 4: LN_B(4); break;
    default: return A2V_ERR_ARG;
  }
#undef LN_B1
#undef LN_B2
#undef LN_B
  if (spread) spread_fold(spread, dgamma, dbeta, dxsum, C, st);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, void* dx, const void* dres, float* dgamma,
                                 float* dbeta, float* dxsum, long long M, int C, long long ldx, long long lddy, int act,
                                 float drop_p, unsigned drop_seed, int drop_mode, void* dx_drop, hipStream_t stream) {
  if (M <= 0 || C <= 0 || C % 8 || ldx % 8 || lddy % 8 || C > 512 * LN_MAXCHUNK) return A2V_ERR_ARG;
  if ((gamma == nullptr) != (beta == nullptr)) return A2V_ERR_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return A2V_ERR_ARG;
  if (drop_mode < 0 || drop_mode > 2 || (drop_mode && !(drop_p > 0.f && drop_p < 1.f)) || (drop_mode == 2 && !dx_drop) ||
      (drop_mode && (ldx != C || lddy != C)))
    return A2V_ERR_ARG;
  const Drop drop = make_drop(drop_mode ? drop_p : 0.f, drop_seed);
  if (dtype == A2V_BF16) return ln_bwd_launch<bf16>(dy, x, gamma, beta, mean, rstd, dx, dres, dgamma, dbeta, dxsum, M, C, ldx, lddy, act, drop, drop_mode, dx_drop, stream);
  if (dtype == A2V_F32) return ln_bwd_launch<float>(dy, x, gamma, beta, mean, rstd, dx, dres, dgamma, dbeta, dxsum, M, C, ldx, lddy, act, drop, drop_mode, dx_drop, stream);
  return A2V_ERR_DTYPE;
}

extern "C" int a2v_instnorm_targets(int dtype, const void* const* layers, int L, float* mean_ws, float* rstd_ws,
                                    float* y, int B, int Tn, int D, float eps, hipStream_t stream) {
  if (L <= 0 || L > 32 || B <= 0 || Tn <= 0 || D <= 0 || D % 8) return A2V_ERR_ARG;
  PtrList pl;
  for (int i = 0; i < 32; ++i) pl.p[i] = i < L ? layers[i] : nullptr;
  dim3 g1((D + 511) / 512, B, L), blk(256);
  if ((long long)B * Tn > 0x7FFFFFFFLL) return A2V_ERR_ARG;
  int g2 = (int)(((long long)B * Tn + 3) / 4);                 // one wave per frame
  if (g2 > 8192) g2 = 8192;
  // statistics-in-registers form: 512-column chunks of a clip x frame ranges (~8 workgroups per CU).  bf16 inputs only: it forms
  // sum_l x rstd_l - sum_l mean_l rstd_l, two terms of size |mean| / std that cancel -- |mean| / std x 2^-24 of absolute error in
  // normalised units, nothing beside the 2^-9-relative rounding a bf16 x carries into the same subtraction, but in the fp32 /
  // bf16x3 parity modes a column with |mean| >> std (bias-dominated channels of trained checkpoints) would spend the 1e-3 budget
  // on it; those modes keep (x - mean_l) rstd_l per element (instnorm_apply_kernel).
  const bool cols = dtype == A2V_BF16 && D % 512 == 0 && L <= 16 && Tn >= 64;
  int tchunk = 0, g3 = 0;
  if (cols) {
    const int nchunk = D / 512;
    int per_clip = cdiv(2048, B * nchunk);               // groups of 4 frame ranges per (clip, chunk)
    if (per_clip < 1) per_clip = 1;
    tchunk = cdiv(Tn, 4 * per_clip);
    if (tchunk < 16) tchunk = 16;
    g3 = B * nchunk * cdiv(Tn, 4 * tchunk);
  }
  if (dtype == A2V_BF16) {
    hipLaunchKernelGGL(instnorm_stats_kernel<bf16>, g1, blk, 0, stream, pl, mean_ws, rstd_ws, B, Tn, D, eps);
    if (cols) hipLaunchKernelGGL(instnorm_apply_cols_kernel<bf16>, dim3(g3), blk, 0, stream, pl, mean_ws, rstd_ws, y, L, B, Tn, D, tchunk);
    else hipLaunchKernelGGL(instnorm_apply_kernel<bf16>, dim3(g2), blk, 0, stream, pl, mean_ws, rstd_ws, y, L, B, Tn, D);
  } else if (dtype == A2V_F32) {
    hipLaunchKernelGGL(instnorm_stats_kernel<float>, g1, blk, 0, stream, pl, mean_ws, rstd_ws, B, Tn, D, eps);
    hipLaunchKernelGGL(instnorm_apply_kernel<float>, dim3(g2), blk, 0, stream, pl, mean_ws, rstd_ws, y, L, B, Tn, D);
  } else
    return A2V_ERR_DTYPE;
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_colsum(int dtype, const void* x, float* out, long long M, int C, long long ld, hipStream_t stream) {
  if (dtype != A2V_BF16 && dtype != A2V_F32) return A2V_ERR_DTYPE;
  const int V = dtype == A2V_BF16 ? 8 : 4;
  if (M <= 0 || C <= 0 || C > SPREAD_C || C % V || ld % V || (((uintptr_t)x) & 15)) return A2V_ERR_ARG;
  const int rows = 256;
  dim3 grid((C + 64 * V - 1) / (64 * V), (unsigned)((M + rows - 1) / rows)), blk(256);
  float* spread = nullptr;
  if ((long long)grid.x * grid.y >= 2 * SPREAD) {
    spread = spread_ws(stream);
    if (!spread) return A2V_ERR_LAUNCH;
  }
  if (dtype == A2V_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16>, grid, blk, 0, stream, (const bf16*)x, out, M, C, ld, rows, spread);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, blk, 0, stream, (const float*)x, out, M, C, ld, rows, spread);
  if (spread) spread_fold(spread, out, nullptr, nullptr, C, stream);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}
