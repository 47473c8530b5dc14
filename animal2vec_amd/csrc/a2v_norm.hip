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

// =============================================================================================
//                      streaming LayerNorm (bf16, round 5): raw staging, in-order fronts
// =============================================================================================
// What bounds these kernels was measured first (tools/native/copy_bench.hip, profiles/r05_copy_bench.txt, 3 GiB buffers): a copy in
// the LayerNorm's access shape (a wave moves whole 2-KB rows) runs at 5.9 TB/s when every workgroup moves ONE 8-KB piece and
// retires -- workgroups are dispatched in order, so the chip walks the tensor as one narrow front -- and at 4.8-5.2 TB/s as a
// persistent grid-stride loop, whatever its depth (1-8 rows in flight per wave, 2-32 workgroups per CU, hand-counted waits that
// never wait for a store: all within 5 %); reads alone 6.4-6.8, writes alone 5.2 (persistent) / 6.8 (in order).  The row-per-wave
// kernels above have the right launch shape but hold 64-160 registers per row in flight and spend ~1 us per row between the
// last load and the first store (two dependent wave sums, GELU): 4.1-5.2 TB/s.  The kernels here keep the R rows of a wave's batch
// PACKED (4 registers per 16-B chunk) until a row is worked on, pack the result back into the registers the row came in and store all
// R rows at the end of the batch (every wait inside a batch is for loads only: hipcc turns a wait for a load into vmcnt(0) as soon
// as a store is pending next to it); gamma / beta live in LDS.  Launch forms (`bpw`):
//   front (bpw > 0)  a workgroup owns 4 bpw consecutive batches and retires: forward 5.5-5.6 TB/s (was 4.2-5.2); the backward of
//                    the conv stacks (GELU', bias-gradient column sums kept in registers across the workgroup's rows) 5.1-5.4
//   persistent (0)   batches wave, wave + nwaves, ...: the backward flavours with 32-48 column-sum registers per lane (blocks:
//                    dgamma, dbeta, bias gradient), where one reduction per workgroup outweighs the launch order
// profiles/r05_ln_front.txt has the sweep (rows per batch x bpw x flavour) the choices below come from.
int g_a2v_ln_stream = 3;     // a2v_debug_set(30, v): bit 0 forward, bit 1 backward through the streaming kernels
int g_a2v_ln_rows = 0;       // a2v_debug_set(31, v): rows per batch (0 = by shape)
int g_a2v_ln_wpc = 0;        // a2v_debug_set(32, v): workgroups per CU of the persistent form (0 = by flavour)
int g_a2v_ln_bpw = 0;        // a2v_debug_set(33, v): forward launch form: 0 by shape, v > 0 front with bpw = v, -1 persistent
int g_a2v_ln_bpw_bwd = 0;    // a2v_debug_set(34, v): the same for the backward

static int ln_cus() {
  static int cus[A2V_MAX_DEVICES];
  const int d = a2v_device();
  if (!cus[d]) {
    hipDeviceProp_t p;
    cus[d] = (hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return cus[d];
}
template <typename K>
static int ln_occupancy(K kernel, int* cache, int want) {
  const int d = a2v_device();
  if (!cache[d]) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess || n < 1) n = 4;
    cache[d] = n > 8 ? 8 : n;
  }
  const int w = g_a2v_ln_wpc > 0 ? g_a2v_ln_wpc : (want > 0 ? want : cache[d]);
  return w < cache[d] ? w : cache[d];
}
// rows per batch: the shape's choice (want) or the forced one, never more than the flavour's registers allow (rmax)
static int ln_pick_rows(int want, int rmax) {
  int r = g_a2v_ln_rows > 0 ? g_a2v_ln_rows : want;
  if (r > rmax) r = rmax;
  return r == 3 ? 2 : (r > 4 && r < 8 ? 4 : r);
}
// (the empty asm pins the packed value where it is formed: the stores sit under "row < M" at the end of the batch and LLVM otherwise
//  sinks the whole normalisation of every row down to them, keeping all R rows unpacked -- 220 registers instead of 80)
__device__ __forceinline__ uint4 ln_pack8(const float* o) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
  uint4 u = __builtin_bit_cast(uint4, v);
  asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w));
  return u;
}

template <int NCH, int R, int ACT, bool AFFINE, bool POST, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_stream_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            const bf16* __restrict__ post, long long M, int C, float eps, int bpw) {
  __shared__ __attribute__((aligned(16))) float sgb[AFFINE ? 2 * NCH * 512 : 4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (AFFINE) {
    for (int k = threadIdx.x; k < NCH * 512; k += 256) {
      sgb[k] = k < C ? gamma[k] : 1.f;
      sgb[NCH * 512 + k] = k < C ? beta[k] : 0.f;
    }
    __syncthreads();
  }
  int col[NCH];
  bool ok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int cc = (c * 64 + lane) * 8;
    ok[c] = FULL || cc < C;
    col[c] = ok[c] ? cc : 0;                      // (idle lanes of a ragged last chunk re-read column 0: no load under a lane mask)
  }
  const long long nb = (M + R - 1) / R;
  // bpw = 0: persistent, batches wave, wave + nwaves, ...; bpw > 0: the workgroup owns 4 bpw CONSECUTIVE batches and retires
  // (workgroups are dispatched in order: the chip then walks the tensor as one narrow front -- tools/native/copy_bench.hip)
  const long long nw = bpw ? 4 : (long long)gridDim.x * 4;
  const long long bfirst = (bpw ? (long long)blockIdx.x * bpw * 4 : (long long)blockIdx.x * 4) + wid;
  const long long bend = bpw ? ((long long)(blockIdx.x + 1) * bpw * 4 < nb ? (long long)(blockIdx.x + 1) * bpw * 4 : nb) : nb;
  for (long long b = bfirst; b < bend; b += nw) {
    const long long row0 = b * R;
    Raw8<bf16> xr[R][NCH], pr[POST ? R : 1][POST ? NCH : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = row0 + r < M ? row0 + r : M - 1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        xr[r][c].load(x + row * C + col[c]);
        if constexpr (POST) pr[r][c].load(post + row * C + col[c]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    float mymean = 0.f, myrstd = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {               // (rows past M are copies of the last row: computed, never stored -- no branch in here)
      float v[NCH][8];
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        xr[r][c].unpack(v[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (!FULL && !ok[c]) v[c][i] = 0.f;
          s += v[c][i];
        }
      }
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = (!FULL && !ok[c]) ? 0.f : v[c][i] - mean;
          q += d * d;
        }
      const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
      if (lane == r) { mymean = mean; myrstd = rstd; }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float o[8], ga[8], be[8], pv[8];
        if constexpr (AFFINE) {
          const f32x4 g0 = *(const f32x4*)(sgb + col[c]), g1 = *(const f32x4*)(sgb + col[c] + 4);
          const f32x4 b0 = *(const f32x4*)(sgb + NCH * 512 + col[c]), b1 = *(const f32x4*)(sgb + NCH * 512 + col[c] + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { ga[i] = g0[i]; ga[4 + i] = g1[i]; be[i] = b0[i]; be[4 + i] = b1[i]; }
        }
        if constexpr (POST) pr[r][c].unpack(pv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float t = (v[c][i] - mean) * rstd;
          if constexpr (AFFINE) t = t * ga[i] + be[i];
          if constexpr (ACT == 1) t = gelu_f(t);
          if constexpr (POST) t += pv[i];
          o[i] = t;
        }
        xr[r][c].v = ln_pack8(o);
      }
      __builtin_amdgcn_sched_barrier(0);          // (rows one after the other: interleaved, their temporaries cost waves of occupancy)
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row0 + r >= M) break;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (FULL || ok[c]) *(uint4*)(y + (row0 + r) * C + col[c]) = xr[r][c].v;
    }
    if (mean_out && lane < R && row0 + lane < M) { mean_out[row0 + lane] = mymean; rstd_out[row0 + lane] = myrstd; }
  }
}

// waves per SIMD the backward is built for: 4 (<= 128 registers); the affine flavours at C > 512 carry 32-48 column-sum registers
// per lane besides the row (3 waves, one row per batch)
#define LN_BWD_WAVES ((AFFINE && NCH == 2) ? 3 : 4)
// Backward of y = act(LN(x) gamma + beta) in the same form.  dx is packed into the registers dy came in, dropout(dx) (mode 2) into
// x's; the column sums (dgamma / dbeta / dxsum) stay in registers over all batches of the wave and leave once per workgroup.
template <int NCH, int R, int ACT, bool AFFINE, int DM, bool XS, bool FULL>
__global__ __launch_bounds__(256, LN_BWD_WAVES) void ln_bwd_stream_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            bf16* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dxsum, long long M,
                                                            int C, Drop drop, bf16* __restrict__ dx_drop,
                                                            float* __restrict__ spread, int bpw) {
  __shared__ float red[4][64 * 8];
  __shared__ __attribute__((aligned(16))) float sgb[AFFINE ? (ACT ? 2 : 1) * NCH * 512 : 4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (AFFINE) {
    for (int k = threadIdx.x; k < NCH * 512; k += 256) {
      sgb[k] = k < C ? gamma[k] : 1.f;
      if constexpr (ACT == 1) sgb[NCH * 512 + k] = k < C ? beta[k] : 0.f;
    }
    __syncthreads();
  }
  int col[NCH];
  bool ok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int cc = (c * 64 + lane) * 8;
    ok[c] = FULL || cc < C;
    col[c] = ok[c] ? cc : 0;
  }
  float ag[AFFINE ? NCH : 1][8], ab[AFFINE ? NCH : 1][8], as_[XS ? NCH : 1][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (AFFINE) { ag[c][i] = 0.f; ab[c][i] = 0.f; }
      if constexpr (XS) as_[c][i] = 0.f;
    }
  const long long nb = (M + R - 1) / R;
  // bpw = 0: persistent, batches wave, wave + nwaves, ...; bpw > 0: the workgroup owns 4 bpw CONSECUTIVE batches and retires
  // (workgroups are dispatched in order: the chip then walks the tensor as one narrow front -- tools/native/copy_bench.hip)
  const long long nw = bpw ? 4 : (long long)gridDim.x * 4;
  const long long bfirst = (bpw ? (long long)blockIdx.x * bpw * 4 : (long long)blockIdx.x * 4) + wid;
  const long long bend = bpw ? ((long long)(blockIdx.x + 1) * bpw * 4 < nb ? (long long)(blockIdx.x + 1) * bpw * 4 : nb) : nb;
  for (long long b = bfirst; b < bend; b += nw) {
    const long long row0 = b * R;
    Raw8<bf16> xr[R][NCH], dr[R][NCH];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = row0 + r < M ? row0 + r : M - 1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        xr[r][c].load(x + row * C + col[c]);
        dr[r][c].load(dy + row * C + col[c]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {               // (rows past M are copies of the last row with dy = 0: no branch in here)
      const bool live = row0 + r < M;
      const long long row = live ? row0 + r : M - 1;
      const float mean = mean_in[row], rstd = rstd_in[row];
      // phase 1: the row's two sums (and the column sums of dgamma / dbeta).  Nothing per element survives it except, with an
      // activation, d = dy gelu'(.) (16 registers): phase 2 unpacks x and dy again instead of keeping xhat and dy gamma (32 more)
      float dk[ACT == 1 ? NCH : 1][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float xv[8], dv[8], ga8[8], bt8[8];
        xr[r][c].unpack(xv);
        dr[r][c].unpack(dv);
        if constexpr (AFFINE) {
          const f32x4 g0 = *(const f32x4*)(sgb + col[c]), g1 = *(const f32x4*)(sgb + col[c] + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { ga8[i] = g0[i]; ga8[4 + i] = g1[i]; }
          if constexpr (ACT == 1) {
            const f32x4 b0 = *(const f32x4*)(sgb + NCH * 512 + col[c]), b1 = *(const f32x4*)(sgb + NCH * 512 + col[c] + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { bt8[i] = b0[i]; bt8[4 + i] = b1[i]; }
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float h = (xv[i] - mean) * rstd;
          float d = ((!FULL && !ok[c]) || !live) ? 0.f : dv[i];
          if constexpr (ACT == 1) {
            d *= gelu_grad_f(AFFINE ? h * ga8[i] + bt8[i] : h);
            dk[c][i] = d;
          }
          float dg = d;
          if constexpr (AFFINE) {
            ag[c][i] += d * h;
            ab[c][i] += d;
            dg = d * ga8[i];
          }
          s1 += dg;
          s2 += dg * h;
        }
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {       // (opaque: otherwise the unpacked values of phase 1 are kept for phase 2 -- the point was not to)
        asm volatile("" : "+v"(xr[r][c].v.x), "+v"(xr[r][c].v.y), "+v"(xr[r][c].v.z), "+v"(xr[r][c].v.w));
        if constexpr (ACT == 0) asm volatile("" : "+v"(dr[r][c].v.x), "+v"(dr[r][c].v.y), "+v"(dr[r][c].v.z), "+v"(dr[r][c].v.w));
      }
      // phase 2: dx = rstd (dg - s1 - xhat s2) = d (rstd gamma) + (x - mean) (-rstd^2 s2) + (-rstd s1)
      const float kb = -rstd * rstd * s2, kc = -rstd * s1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float xv[8], dv[8], ga8[8], o[8];
        xr[r][c].unpack(xv);
        if constexpr (ACT == 0) dr[r][c].unpack(dv);
        if constexpr (AFFINE) {
          const f32x4 g0 = *(const f32x4*)(sgb + col[c]), g1 = *(const f32x4*)(sgb + col[c] + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { ga8[i] = g0[i] * rstd; ga8[4 + i] = g1[i] * rstd; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = ACT == 1 ? dk[c][i] : dv[i];
          o[i] = fmaf(d, AFFINE ? ga8[i] : rstd, fmaf(xv[i] - mean, kb, kc));
        }
        dr[r][c].v = ln_pack8(o);
        if constexpr (DM == 2) {
          drop_apply_n<8>(drop, (unsigned long long)row * C + col[c], o);
          xr[r][c].v = ln_pack8(o);
        }
        if constexpr (XS) {
          float q[8];
          (DM == 2 ? xr[r][c] : dr[r][c]).unpack(q);               // the sum of what is stored (rounded to bf16)
#pragma unroll
          for (int i = 0; i < 8; ++i) as_[c][i] += ((!FULL && !ok[c]) || !live) ? 0.f : q[i];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row0 + r >= M) break;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (FULL || ok[c]) {
          *(uint4*)(dx + (row0 + r) * C + col[c]) = dr[r][c].v;
          if constexpr (DM == 2) *(uint4*)(dx_drop + (row0 + r) * C + col[c]) = xr[r][c].v;
        }
    }
  }
  // cross-wave reduction of the per-column partials, then one atomic per column and workgroup
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    if (which < 2 && !(AFFINE && dgamma)) continue;
    if (which == 2 && !(XS && dxsum)) continue;
    float* dst = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
    if (spread) dst = spread + ((long long)(blockIdx.x % SPREAD) * 3 + which) * SPREAD_C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float vv = 0.f;
        if constexpr (AFFINE) { if (which == 0) vv = ag[c][i]; if (which == 1) vv = ab[c][i]; }
        if constexpr (XS) { if (which == 2) vv = as_[c][i]; }
        red[wid][lane * 8 + i] = vv;
      }
      __syncthreads();
      for (int k = threadIdx.x; k < 512; k += 256) {
        const int cc = c * 512 + k;
        if (cc < C) atomicAdd(dst + cc, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
      }
    }
  }
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
// ---- streaming forms: host side
template <int NCH, int R, int ACT, bool AFFINE, bool POST, bool FULL>
static void ln_fwd_stream_go(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                             const void* post, long long M, int C, float eps, hipStream_t st) {
  static int occ[A2V_MAX_DEVICES];
  auto kern = ln_fwd_stream_kernel<NCH, R, ACT, AFFINE, POST, FULL>;
  const long long need = ((M + R - 1) / R + 3) / 4;
  const long long cap = (long long)ln_occupancy(kern, occ, 0) * ln_cus();
  const int bpw = g_a2v_ln_bpw == 0 ? 1 : (g_a2v_ln_bpw < 0 ? 0 : g_a2v_ln_bpw);       // forward: one batch per wave, in order
  const long long grid = bpw > 0 ? (need + bpw - 1) / bpw : (need < cap ? need : cap);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, st, (const bf16*)x, gamma, beta, (bf16*)y, mean,
                     rstd, (const bf16*)post, M, C, eps, bpw);
}
template <int NCH, int ACT, bool AFFINE, bool POST>
static void ln_fwd_stream_rows(int cpb, bool full, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                               float* rstd, const void* post, long long M, int C, float eps, hipStream_t st) {
#define LN_GO(R_)                                                                                                   \
  { if (full) ln_fwd_stream_go<NCH, R_, ACT, AFFINE, POST, true>(x, gamma, beta, y, mean, rstd, post, M, C, eps, st); \
    else ln_fwd_stream_go<NCH, R_, ACT, AFFINE, POST, false>(x, gamma, beta, y, mean, rstd, post, M, C, eps, st); }
  const int r = cpb / NCH;            // cpb: 16-B chunks of one tensor per lane and batch (2, 4 or 8)
  if (r <= 1 * (2 / NCH)) LN_GO(1 * (2 / NCH))
  else if (r <= 2 * (2 / NCH)) LN_GO(2 * (2 / NCH))
  else LN_GO(4 * (2 / NCH))
#undef LN_GO
}
// true when the call was taken
static bool ln_fwd_stream(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* sum_out,
                          float* mean, float* rstd, const void* post, long long M, int C, long long ldx, long long ldy, float eps,
                          int act, Drop drop, hipStream_t st) {
  if (!(g_a2v_ln_stream & 1) || res || sum_out || drop.thresh || ldx != C || ldy != C || C > 1024 || M < 2048) return false;
  const bool aff = gamma != nullptr;
  if (!act && (!aff || post)) return false;
  if (act && aff && post) return false;
  const int nch = (C + 511) / 512;
  const bool full = C == nch * 512;
  const int rmax = (post ? 4 : 8) / nch;                       // rows per batch the flavour's registers allow
  // 4 chunks of 16 B per lane and batch (C = 1024: 2 rows, C = 512: 4 rows); one row for the longest C = 1024 tensors
  const int cpb = ln_pick_rows((nch == 2 && full && M > 262144) ? 1 : 4 / nch, rmax) * nch;
#define LN_FL(N)                                                                                                            \
  { if (!act) ln_fwd_stream_rows<N, 0, true, false>(cpb, full, x, gamma, beta, y, mean, rstd, post, M, C, eps, st);          \
    else if (aff) ln_fwd_stream_rows<N, 1, true, false>(cpb, full, x, gamma, beta, y, mean, rstd, post, M, C, eps, st);     \
    else if (post) ln_fwd_stream_rows<N, 1, false, true>(cpb, full, x, gamma, beta, y, mean, rstd, post, M, C, eps, st);    \
    else ln_fwd_stream_rows<N, 1, false, false>(cpb, full, x, gamma, beta, y, mean, rstd, post, M, C, eps, st); }
  if (nch == 1) LN_FL(1) else LN_FL(2)
#undef LN_FL
  return true;
}

template <typename T>
static int ln_fwd_launch(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* sum_out,
                         float* mean, float* rstd, const void* post, long long M, int C, long long ldx, long long ldy, float eps, int act,
                         Drop drop, hipStream_t st) {
  if (sizeof(T) == 2 && ln_fwd_stream(x, res, gamma, beta, y, sum_out, mean, rstd, post, M, C, ldx, ldy, eps, act, drop, st)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
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

template <int NCH, int R, int ACT, bool AFFINE, int DM, bool XS, bool FULL>
static int ln_bwd_stream_go(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, void* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C, Drop drop,
                            void* dx_drop, hipStream_t st) {
  static int occ[A2V_MAX_DEVICES];
  auto kern = ln_bwd_stream_kernel<NCH, R, ACT, AFFINE, DM, XS, FULL>;
  const long long need = ((M + R - 1) / R + 3) / 4;
  // with an activation (conv stacks: GELU', at most 16 column-sum registers): front form, one batch per wave where the rows are
  // full 512-column chunks, 8 where a ragged chunk leaves half a wave idle (C = 768) or the rows are short (C = 512);
  // without (blocks, project_features: dgamma / dbeta / bias gradient): persistent, two workgroups per CU at C > 512
  const int auto_bpw = ACT == 1 ? ((NCH == 2 && FULL) ? 1 : 8) : 0;
  const int bpw = g_a2v_ln_bpw_bwd == 0 ? auto_bpw : (g_a2v_ln_bpw_bwd < 0 ? 0 : g_a2v_ln_bpw_bwd);
  const long long cap = (long long)ln_occupancy(kern, occ, (ACT == 0 && NCH == 2) ? 2 : 0) * ln_cus();
  const unsigned grid = (unsigned)(bpw > 0 ? (need + bpw - 1) / bpw : (need < cap ? need : cap));
  float* spread = nullptr;
  if (grid >= 4 * SPREAD && (dgamma || dxsum)) {
    spread = spread_ws(st);
    if (!spread) return A2V_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, gamma, beta, mean, rstd, (bf16*)dx, dgamma,
                     dbeta, dxsum, M, C, drop, (bf16*)dx_drop, spread, bpw);
  if (spread) spread_fold(spread, dgamma, dbeta, dxsum, C, st);
  return A2V_OK;
}
template <int NCH, int ACT, bool AFFINE, int DM, bool XS>
static int ln_bwd_stream_rows(int cpb, bool full, const void* dy, const void* x, const float* gamma, const float* beta,
                              const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, float* dxsum,
                              long long M, int C, Drop drop, void* dx_drop, hipStream_t st) {
#define LN_GO(R_)                                                                                                                \
  return full ? ln_bwd_stream_go<NCH, R_, ACT, AFFINE, DM, XS, true>(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, dxsum, M, C, \
                                                                     drop, dx_drop, st)                                            \
              : ln_bwd_stream_go<NCH, R_, ACT, AFFINE, DM, XS, false>(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, dxsum, M, C, \
                                                                      drop, dx_drop, st);
  if (AFFINE || cpb / NCH <= 1 * (2 / NCH)) { LN_GO(1 * (2 / NCH)) }
  if constexpr (!AFFINE) { LN_GO(2 * (2 / NCH)) }
#undef LN_GO
}
// 1: not taken; otherwise the call's return code
static int ln_bwd_stream(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, void* dx, const void* dres, float* dgamma, float* dbeta, float* dxsum, long long M,
                         int C, long long ldx, long long lddy, int act, Drop drop, int drop_mode, void* dx_drop, hipStream_t st) {
  if (!(g_a2v_ln_stream & 2) || dres || ldx != C || lddy != C || C > 1024 || M < 2048) return 1;
  const bool aff = gamma != nullptr, xs = dxsum != nullptr;
  int fl = -1;                                    // (ACT, AFFINE, DM, XS)
  if (!act && aff && drop_mode == 2 && xs) fl = 0;
  else if (!act && aff && drop_mode == 0) fl = xs ? 1 : 2;
  else if (act && !aff && drop_mode == 0 && xs) fl = 3;
  else if (act && aff && drop_mode == 0 && !xs) fl = 4;
  if (fl < 0) return 1;
  const int nch = (C + 511) / 512;
  const bool full = C == nch * 512;
  const int cpb = ln_pick_rows(4 / nch, 4 / nch) * nch;
#define LN_A dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, dxsum, M, C, drop, dx_drop, st
#define LN_FL(N)                                                                    \
  switch (fl) {                                                                     \
    case 0: return ln_bwd_stream_rows<N, 0, true, 2, true>(cpb, full, LN_A);        \
    case 1: return ln_bwd_stream_rows<N, 0, true, 0, true>(cpb, full, LN_A);        \
    case 2: return ln_bwd_stream_rows<N, 0, true, 0, false>(cpb, full, LN_A);       \
    case 3: return ln_bwd_stream_rows<N, 1, false, 0, true>(cpb, full, LN_A);       \
    default: return ln_bwd_stream_rows<N, 1, true, 0, false>(cpb, full, LN_A);      \
  }
  if (nch == 1) LN_FL(1)
  LN_FL(2)
#undef LN_FL
#undef LN_A
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, void* dx, const void* dres, float* dgamma, float* dbeta, float* dxsum,
                         long long M, int C, long long ldx, long long lddy, int act, Drop drop, int drop_mode,
                         void* dx_drop, hipStream_t st) {
  if (sizeof(T) == 2) {
    const int rc = ln_bwd_stream(dy, x, gamma, beta, mean, rstd, dx, dres, dgamma, dbeta, dxsum, M, C, ldx, lddy, act, drop,
                                 drop_mode, dx_drop, st);
    if (rc != 1) {
      if (rc == A2V_OK) A2V_CHECK_LAUNCH();
      return rc;
    }
  }
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
