// Implicit-GEMM family for gfx950: every dense contraction of the animal2vec hot path.
//
//   NT:  C[m][g*N+n] = epi( sum_kk A_g[m][kk] * B_g[n][kk] )          (Linear fwd/dgrad, conv fwd/dgrad)
//   TN:  W[g][n][kk] += sum_m  P[m][g*N+n] * A_g[m][kk]               (Linear / conv weight gradient)
//
// A is an *implicit im2col view* of a channels-last activation tensor (B, T_in, pitch):
//   A_g[m=(b,t)][kk=(j,ci)] = X[b][t*stride - pad + j][g*goff + ci]   (0 outside [0,T_in))
// so a Linear is taps=1, a dense strided Conv1d is G=1, a grouped Conv1d walks `goff` per group.
// Weights are pre-packed as B[g][n][kk] with kk = j*Cg + ci (K-contiguous).
//
// bf16 mode : v_mfma_f32_16x16x32_bf16, fp32 accumulate, 128x128x64 tiles, 4 waves (2x2), double-
//             buffered LDS fed through registers (the implicit gather needs per-lane predication),
//             XOR-swizzled 16-B chunks for conflict-free ds_read_b128, XCD-aware tile order.
// fp32 mode : v_mfma_f32_16x16x4_f32 (bitwise fmaf chain) -- the parity mode.
// TN bf16 reads both operands with ds_read_b64_tr_b16 (hardware transpose) from row-major tiles.
#include "a2v_gemm.h"

int g_a2v_wgrad_slab = 1;   // a2v_debug_set(2, v): grouped-conv weight grads through the slab kernel
int g_a2v_conv_slab = 1;    // a2v_debug_set(3, v): grouped-conv forward / input grads through the slab kernel
int g_a2v_tn_p8 = 1;        // a2v_debug_set(4, v): Linear weight gradients through the 256x256 TN kernel
int g_a2v_gemm_p8 = 1;   // a2v_debug_set(1, v): route eligible Linear layers to the 256x256 8-phase kernel

// =============================================================================================
//                                   bf16  NT   (128 x 128 x 64)
// =============================================================================================
#define NT_BM 128
#define NT_BN 128
#define NT_BK 64


__global__ __launch_bounds__(NT_THREADS) void gemm_nt_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#define sA_(b) (smem + (b) * 16384)
#define sB_(b) (smem + 32768 + (b) * 16384)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (g.N + NT_BN - 1) / NT_BN;
  const int tiles_m = (g.M + NT_BM - 1) / NT_BM;
  const int ntile = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, ntile);
  const int tm = id / tiles_n, tn = id - tm * tiles_n;
  const int grp = blockIdx.y;
  const int m0 = tm * NT_BM, n0 = tn * NT_BN;

  // ---- staging assignment: 4 chunks of A and 4 of B per thread per k-tile
  const int crow = tid >> 3, cchunk = tid & 7;
  const ARow ar0 = make_arow<bf16>(g.a, m0 + crow, g.M, grp);
  const ARow ar1 = make_arow<bf16>(g.a, m0 + crow + 32, g.M, grp);
  const ARow ar2 = make_arow<bf16>(g.a, m0 + crow + 64, g.M, grp);
  const ARow ar3 = make_arow<bf16>(g.a, m0 + crow + 96, g.M, grp);
  const char* bbase = (const char*)g.B + ((long long)grp * g.b_gstride) * 2;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define NT_LOAD_B(i) \
  ((n0 + crow + 32 * (i) < g.N && kk_ < g.K) ? *(const uint4*)(bbase + ((long long)(n0 + crow + 32 * (i)) * g.ldb + kk_) * 2) : zero4)
#define load_tile(kt)                                                   \
  {                                                                     \
    const int kk_ = (kt) * NT_BK + cchunk * 8;                          \
    ra0 = load_a_chunk<bf16>(g.a, ar0, kk_, g.K);                       \
    ra1 = load_a_chunk<bf16>(g.a, ar1, kk_, g.K);                       \
    ra2 = load_a_chunk<bf16>(g.a, ar2, kk_, g.K);                       \
    ra3 = load_a_chunk<bf16>(g.a, ar3, kk_, g.K);                       \
    rb0 = NT_LOAD_B(0); rb1 = NT_LOAD_B(1); rb2 = NT_LOAD_B(2); rb3 = NT_LOAD_B(3); \
  }
#define store_tile(buf_)                                                \
  {                                                                     \
    *(uint4*)(sA_(buf_) + swz128(crow, cchunk)) = ra0;                  \
    *(uint4*)(sA_(buf_) + swz128(crow + 32, cchunk)) = ra1;             \
    *(uint4*)(sA_(buf_) + swz128(crow + 64, cchunk)) = ra2;             \
    *(uint4*)(sA_(buf_) + swz128(crow + 96, cchunk)) = ra3;             \
    *(uint4*)(sB_(buf_) + swz128(crow, cchunk)) = rb0;                  \
    *(uint4*)(sB_(buf_) + swz128(crow + 32, cchunk)) = rb1;             \
    *(uint4*)(sB_(buf_) + swz128(crow + 64, cchunk)) = rb2;             \
    *(uint4*)(sB_(buf_) + swz128(crow + 96, cchunk)) = rb3;             \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + NT_BK - 1) / NT_BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        fa[i] = *(const bf16x8*)(sA_(buf) + swz128(r, ks * 4 + fq));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + fr;
        fb[j] = *(const bf16x8*)(sB_(buf) + swz128(r, ks * 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = A2V_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue through a per-wave LDS stage (64 x 68 floats)
  float* stage = (float*)smem + wid * (64 * 68);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(i * 16 + fq * 4 + r) * 68 + j * 16 + fr] = acc[i][j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  wave_epilogue<bf16>(g, stage, lane, m0 + wm * 64, n0 + wn * 64, grp);
}

#undef load_tile
#undef store_tile
// =============================================================================================
//                                   fp32  NT   (128 x 128 x 16)   -- parity mode
// =============================================================================================
#define F32_BK 16
#define F32_PITCH 20   // floats per LDS row (16 + 4 pad, keeps 16-B alignment)

__global__ __launch_bounds__(NT_THREADS) void gemm_nt_f32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#define fA_(b) ((float*)smem + (b) * 128 * F32_PITCH)
#define fB_(b) ((float*)smem + (2 + (b)) * 128 * F32_PITCH)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (g.N + NT_BN - 1) / NT_BN;
  const int tiles_m = (g.M + NT_BM - 1) / NT_BM;
  const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = id / tiles_n, tn = id - tm * tiles_n;
  const int grp = blockIdx.y;
  const int m0 = tm * NT_BM, n0 = tn * NT_BN;

  // 128 rows x 4 chunks (of 4 floats) = 512 chunks per operand -> 2 per thread
  const int crow = tid >> 2, cchunk = tid & 3;
  const ARow ar0 = make_arow<float>(g.a, m0 + crow, g.M, grp);
  const ARow ar1 = make_arow<float>(g.a, m0 + crow + 64, g.M, grp);
  const char* bbase = (const char*)g.B + ((long long)grp * g.b_gstride) * 4;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
  uint4 ra0, ra1, rb0, rb1;
#define F32_LOAD_B(i) \
  ((n0 + crow + 64 * (i) < g.N && kk_ < g.K) ? *(const uint4*)(bbase + ((long long)(n0 + crow + 64 * (i)) * g.ldb + kk_) * 4) : zero4)
#define load_tile(kt)                                   \
  {                                                     \
    const int kk_ = (kt) * F32_BK + cchunk * 4;         \
    ra0 = load_a_chunk<float>(g.a, ar0, kk_, g.K);      \
    ra1 = load_a_chunk<float>(g.a, ar1, kk_, g.K);      \
    rb0 = F32_LOAD_B(0); rb1 = F32_LOAD_B(1);           \
  }
#define store_tile(buf_)                                                   \
  {                                                                        \
    *(uint4*)(fA_(buf_) + crow * F32_PITCH + cchunk * 4) = ra0;            \
    *(uint4*)(fA_(buf_) + (crow + 64) * F32_PITCH + cchunk * 4) = ra1;     \
    *(uint4*)(fB_(buf_) + crow * F32_PITCH + cchunk * 4) = rb0;            \
    *(uint4*)(fB_(buf_) + (crow + 64) * F32_PITCH + cchunk * 4) = rb1;     \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + F32_BK - 1) / F32_BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = fA_(buf)[(wm * 64 + i * 16 + fr) * F32_PITCH + ks * 4 + fq];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = fB_(buf)[(wn * 64 + j * 16 + fr) * F32_PITCH + ks * 4 + fq];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }
  float* stage = (float*)smem + wid * (64 * 68);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(i * 16 + fq * 4 + r) * 68 + j * 16 + fr] = acc[i][j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  wave_epilogue<float>(g, stage, lane, m0 + wm * 64, n0 + wn * 64, grp);
}

#undef load_tile
#undef store_tile
// =============================================================================================
//                         bf16  TN  (weight gradient)   128(n) x 128(kk), M step 64
// =============================================================================================

#define TN_MSTEP 64

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#define sP_(b) (smem + (b) * 16384)
#define sQ_(b) (smem + 32768 + (b) * 16384)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_k = (g.K + 127) / 128;
  const int tn_ = blockIdx.x / tiles_k, tk = blockIdx.x - tn_ * tiles_k;
  const int grp = blockIdx.y;
  const int n0 = tn_ * 128, k0 = tk * 128;
  const int mbeg = blockIdx.z * g.m_chunk;
  const int mend = min(g.M, mbeg + g.m_chunk);

  // staging: tile = 64 rows x 16 chunks(16 B) = 1024 chunks per operand, 4 per thread
  const int crow = tid >> 4, cchunk = tid & 15;   // rows crow + 16*i
  const bf16* P = (const bf16*)g.P;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
  uint4 rp0, rp1, rp2, rp3, rq0, rq1, rq2, rq3;
  const int pn = n0 + cchunk * 8;
#define TN_LOAD_P(i) \
  ((mt_ + crow + 16 * (i) < mend && pn < g.N) ? *(const uint4*)(P + (long long)(mt_ + crow + 16 * (i)) * g.ldp + (long long)grp * g.p_goff + pn) : zero4)
#define TN_LOAD_Q(i) load_a_chunk<bf16>(g.a, make_arow<bf16>(g.a, mt_ + crow + 16 * (i), mend, grp), k0 + cchunk * 8, g.K)
#define load_tile(mt)                                                          \
  {                                                                            \
    const int mt_ = (mt);                                                      \
    rp0 = TN_LOAD_P(0); rp1 = TN_LOAD_P(1); rp2 = TN_LOAD_P(2); rp3 = TN_LOAD_P(3); \
    rq0 = TN_LOAD_Q(0); rq1 = TN_LOAD_Q(1); rq2 = TN_LOAD_Q(2); rq3 = TN_LOAD_Q(3); \
  }
#define TN_OFF(r) ((r) * 256 + ((((cchunk >> 1) ^ tn_f(r)) & 7) << 5) + ((cchunk & 1) << 4))
#define store_tile(buf_)                                   \
  {                                                        \
    *(uint4*)(sP_(buf_) + TN_OFF(crow)) = rp0;             \
    *(uint4*)(sP_(buf_) + TN_OFF(crow + 16)) = rp1;        \
    *(uint4*)(sP_(buf_) + TN_OFF(crow + 32)) = rp2;        \
    *(uint4*)(sP_(buf_) + TN_OFF(crow + 48)) = rp3;        \
    *(uint4*)(sQ_(buf_) + TN_OFF(crow)) = rq0;             \
    *(uint4*)(sQ_(buf_) + TN_OFF(crow + 16)) = rq1;        \
    *(uint4*)(sQ_(buf_) + TN_OFF(crow + 32)) = rq2;        \
    *(uint4*)(sQ_(buf_) + TN_OFF(crow + 48)) = rq3;        \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nsteps = (mend - mbeg + TN_MSTEP - 1) / TN_MSTEP;
  if (nsteps <= 0) return;
  load_tile(mbeg);
  store_tile(0);
  __syncthreads();

  const int gq = lane >> 4;           // lane group 0..3 -> k rows 8*gq .. 8*gq+7
  const int q = (lane & 15) >> 2, p = lane & 3;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) load_tile(mbeg + (st + 1) * TN_MSTEP);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
      const int r_lo = ks * 32 + 8 * gq + q;       // rows for elements 0..3
      const int r_hi = r_lo + 4;                   // rows for elements 4..7
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = wm * 64 + i * 16 + 4 * p;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sP_(buf) + tn_off(r_lo, col)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sP_(buf) + tn_off(r_hi, col)));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        fa[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wn * 64 + j * 16 + 4 * p;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sQ_(buf) + tn_off(r_lo, col)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sQ_(buf) + tn_off(r_hi, col)));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        fb[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = A2V_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nsteps) store_tile(buf ^ 1);
    __syncthreads();
  }
  // accumulate: D[row=(lane>>4)*4+r][col=lane&15] -> W[g][n][kk]
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wm * 64 + i * 16 + fq * 4 + r;
      if (n >= g.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k0 + wn * 64 + j * 16 + fr;
        if (kk < g.K) atomicAdd(g.W + (long long)grp * g.w_gstride + (long long)n * g.ldw + kk, acc[i][j][r]);
      }
    }
}

#undef load_tile
#undef store_tile
// fp32 TN: 128(n) x 128(kk), M step 16, v_mfma_f32_16x16x4_f32
#define TNF_PITCH 132
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#define fP_(b) ((float*)smem + (b) * 16 * TNF_PITCH)
#define fQ_(b) ((float*)smem + (2 + (b)) * 16 * TNF_PITCH)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_k = (g.K + 127) / 128;
  const int tn_ = blockIdx.x / tiles_k, tk = blockIdx.x - tn_ * tiles_k;
  const int grp = blockIdx.y;
  const int n0 = tn_ * 128, k0 = tk * 128;
  const int mbeg = blockIdx.z * g.m_chunk;
  const int mend = min(g.M, mbeg + g.m_chunk);
  // tile 16 rows x 32 chunks (4 floats) = 512 chunks -> 2 per thread
  const int crow = tid >> 5, cchunk = tid & 31;   // rows crow + 8*i
  const float* P = (const float*)g.P;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
  uint4 rp0, rp1, rq0, rq1;
  const int pn = n0 + cchunk * 4;
#define TNF_LOAD_P(i) \
  ((mt_ + crow + 8 * (i) < mend && pn < g.N) ? *(const uint4*)(P + (long long)(mt_ + crow + 8 * (i)) * g.ldp + (long long)grp * g.p_goff + pn) : zero4)
#define TNF_LOAD_Q(i) load_a_chunk<float>(g.a, make_arow<float>(g.a, mt_ + crow + 8 * (i), mend, grp), k0 + cchunk * 4, g.K)
#define load_tile(mt)                               \
  {                                                 \
    const int mt_ = (mt);                           \
    rp0 = TNF_LOAD_P(0); rp1 = TNF_LOAD_P(1);       \
    rq0 = TNF_LOAD_Q(0); rq1 = TNF_LOAD_Q(1);       \
  }
#define store_tile(buf_)                                                   \
  {                                                                        \
    *(uint4*)(fP_(buf_) + crow * TNF_PITCH + cchunk * 4) = rp0;            \
    *(uint4*)(fP_(buf_) + (crow + 8) * TNF_PITCH + cchunk * 4) = rp1;      \
    *(uint4*)(fQ_(buf_) + crow * TNF_PITCH + cchunk * 4) = rq0;            \
    *(uint4*)(fQ_(buf_) + (crow + 8) * TNF_PITCH + cchunk * 4) = rq1;      \
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nsteps = (mend - mbeg + 15) / 16;
  if (nsteps <= 0) return;
  load_tile(mbeg);
  store_tile(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) load_tile(mbeg + (st + 1) * 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = fP_(buf)[(ks * 4 + fq) * TNF_PITCH + wm * 64 + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = fQ_(buf)[(ks * 4 + fq) * TNF_PITCH + wn * 64 + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nsteps) store_tile(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wm * 64 + i * 16 + fq * 4 + r;
      if (n >= g.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k0 + wn * 64 + j * 16 + fr;
        if (kk < g.K) atomicAdd(g.W + (long long)grp * g.w_gstride + (long long)n * g.ldw + kk, acc[i][j][r]);
      }
    }
}

#undef load_tile
#undef store_tile
// =============================================================================================
//                                         C ABI
// =============================================================================================
static int check_conv(const ConvA& a, int esz) {
  const int per16 = 16 / esz;
  if (a.Cg <= 0 || a.Cg % per16 || a.Cv % per16 || a.pitch % per16 || a.goff % per16 || a.bstride % per16) return A2V_ERR_ARG;
  if (a.taps <= 0 || a.stride <= 0 || a.T_in <= 0 || a.T_out <= 0) return A2V_ERR_ARG;
  if (((uintptr_t)a.ptr) & 15) return A2V_ERR_ARG;
  return A2V_OK;
}

static int conv_gemm_impl(const int* rowidx, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                             int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                             const void* Bw, long long ldb,
                             long long b_gstride, void* C, long long ldc, int c_goff, void* C2, const void* R,
                             long long ldr, const float* bias, float* colsum, int M, int N, int K, int G, int flags,
                             float drop_p, unsigned drop_seed, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || G <= 0) return A2V_ERR_ARG;
  if (dtype != A2V_F32 && dtype != A2V_BF16 && dtype != A2V_BF16X3 && dtype != A2V_BF16X3S) return A2V_ERR_DTYPE;
  const int esz = dtype == A2V_F32 ? 4 : 2;
  GemmArgs g;
  g.a = ConvA{A, a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, (Cv <= 0 || Cv > Cg) ? Cg : Cv, rowmask, bdiv < 1 ? 1 : bdiv, rowidx};
  int rc = check_conv(g.a, esz);
  if (rc) return rc;
  if (K != taps * Cg) return A2V_ERR_ARG;
  if (N % 4 || ldc % 4 || c_goff % 4 || (ldb * esz) % 16 || (b_gstride * esz) % 16) return A2V_ERR_ARG;
  if ((flags & (F_RES | F_GELU_BWD)) && (!R || ldr % 4)) return A2V_ERR_ARG;
  if ((flags & F_BIAS) && !bias) return A2V_ERR_ARG;
  if ((flags & F_STORE_PRE) && !C2) return A2V_ERR_ARG;
  if ((flags & F_COLSUM) && !colsum) return A2V_ERR_ARG;
  g.B = Bw; g.ldb = ldb; g.b_gstride = b_gstride;
  g.C = C; g.ldc = ldc; g.c_goff = c_goff; g.C2 = C2; g.R = R; g.ldr = ldr;
  g.bias = bias; g.colsum = colsum; g.M = M; g.N = N; g.K = K; g.G = G; g.flags = flags;
  g.drop = make_drop((flags & F_DROPOUT) ? drop_p : 0.f, drop_seed);
  g.drop_base = 0;
  if ((flags & F_DROPOUT) && !(drop_p > 0.f && drop_p < 1.f)) return A2V_ERR_ARG;
  const int tiles = cdiv(M, NT_BM) * cdiv(N, NT_BN);
  dim3 grid(tiles, G, 1);
  if (dtype == A2V_BF16X3S) {        // [hi ; lo] stacked operands: the three products of a 64-wide grouped conv in one pass
    if (rowidx || M % T_out) return A2V_ERR_UNSUPPORTED;
    const long long a_lo = (long long)cdiv(M / T_out, g.a.bdiv) * a_bstride;
    const long long b_lo = G > 1 ? (long long)G * b_gstride : (long long)N * ldb;
    if (!a2v_launch_conv_slab3(g, a_lo, b_lo, stream)) return A2V_ERR_UNSUPPORTED;
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (flags & F_DACT8) {
    // the 8-bit derivative tensor (include/a2v.h: A2V_EPI_DACT8): bf16 Linear products only, as the fc1 forward (gelu + stored
    // derivative) or the fc2 input gradient (derivative multiplied in) -- the persistent kernel's fixed flavours, else the 128 x 128 kernels
    const bool st = (flags & (F_DACT | F_GELU | F_STORE_PRE)) == (F_DACT | F_GELU | F_STORE_PRE) && !(flags & F_GELU_BWD);
    const bool mu = (flags & (F_DACT | F_GELU_BWD)) == (F_DACT | F_GELU_BWD) && !(flags & (F_STORE_PRE | F_RES));
    if (dtype != A2V_BF16 || !(st || mu) || rowidx || rowmask || G != 1 || taps != 1 || stride != 1 || (st ? ldc % 8 : ldr % 8))
      return A2V_ERR_UNSUPPORTED;
    if (g_a2v_gemm_p8 && a2v_launch_nt_bf16_p8(g, stream)) {
      A2V_CHECK_LAUNCH();
      return A2V_OK;
    }
    if (!a2v_launch_nt_bf16_fast(g, stream)) {
      static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        attr = true;
      }
      hipLaunchKernelGGL(gemm_nt_bf16_kernel, grid, dim3(NT_THREADS), NT_LDS_BYTES, stream, g);
    }
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (flags & F_QKNORM) {
    // row-norm maxima per clip and 32-column segment (include/a2v.h: A2V_EPI_QKNORM) into `colsum`: one fixed flavour of the persistent
    // kernel forms them, nothing else does -- a launch it does not take is refused, never run without them
    if (dtype != A2V_BF16 || flags != (F_BIAS | F_QKNORM) || rowidx || rowmask || G != 1 || taps != 1 || stride != 1 || !colsum ||
        N % 32 || T_in != T_out || M % T_out || !g_a2v_gemm_p8 || !a2v_launch_nt_bf16_p8(g, stream))
      return A2V_ERR_UNSUPPORTED;
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if ((flags & F_X3IMG) && (dtype != A2V_BF16X3 || rowidx || !a2v_launch_nt_x3_p8(g, stream))) return A2V_ERR_UNSUPPORTED;
  if (flags & F_X3IMG) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16X3) {         // split bf16 operands, fp32 outputs: the persistent 8-phase kernel, else the 128 x 128 LDS-DMA kernel
    if (!(!rowidx && a2v_launch_conv_slab_o32(g, stream)) && !(!rowidx && a2v_launch_nt_x3_p8(g, stream)) &&
        !a2v_launch_nt_x3_fast(g, stream))
      return A2V_ERR_UNSUPPORTED;
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16 && !rowidx && a2v_launch_conv_dec48(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16 && !rowidx && g_a2v_conv_slab && a2v_launch_conv_slab(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16 && !rowidx && g_a2v_gemm_p8 && a2v_launch_conv_strided_nt(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
#ifdef A2V_W2
  if (dtype == A2V_BF16 && !rowidx && g_a2v_gemm_p8 && a2v_launch_nt_bf16_w2(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
#endif
  if (dtype == A2V_BF16 && !rowidx && g_a2v_gemm_p8 && a2v_launch_nt_bf16_p8(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16 && a2v_launch_nt_bf16_fast(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16) {
    static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_nt_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
      attr = true;
    }
    hipLaunchKernelGGL(gemm_nt_bf16_kernel, grid, dim3(NT_THREADS), NT_LDS_BYTES, stream, g);
  } else {
    static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
      attr = true;
    }
    hipLaunchKernelGGL(gemm_nt_f32_kernel, grid, dim3(NT_THREADS), NT_LDS_BYTES, stream, g);
  }
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_conv_gemm(int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                             int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                             const void* Bw, long long ldb,
                             long long b_gstride, void* C, long long ldc, int c_goff, void* C2, const void* R,
                             long long ldr, const float* bias, float* colsum, int M, int N, int K, int G, int flags,
                             float drop_p, unsigned drop_seed, hipStream_t stream) {
  return conv_gemm_impl(nullptr, dtype, A, a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, Cv, rowmask, bdiv, Bw, ldb, b_gstride, C, ldc, c_goff, C2, R, ldr, bias, colsum, M, N, K, G, flags, drop_p, drop_seed, stream);
}
extern "C" int a2v_conv_gemm_rows(const int* rowidx, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                             int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                             const void* Bw, long long ldb,
                             long long b_gstride, void* C, long long ldc, int c_goff, void* C2, const void* R,
                             long long ldr, const float* bias, float* colsum, int M, int N, int K, int G, int flags,
                             float drop_p, unsigned drop_seed, hipStream_t stream) {
  return conv_gemm_impl(rowidx, dtype, A, a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, Cv, rowmask, bdiv, Bw, ldb, b_gstride, C, ldc, c_goff, C2, R, ldr, bias, colsum, M, N, K, G, flags, drop_p, drop_seed, stream);
}

int g_a2v_wgrad_sparse = 1;   // a2v_debug_set(37, v): row-sparse weight gradients of the positional encoder (a2v_wgrad_sparse.hip) on / off
static int conv_wgrad_impl(const int* rowidx, int n_seq, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                              int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                              const void* P, long long ldp,
                              int p_goff, float* W, long long ldw, long long w_gstride, int M, int N, int K, int G,
                              hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || G <= 0) return A2V_ERR_ARG;
  if (dtype != A2V_F32 && dtype != A2V_BF16) return A2V_ERR_DTYPE;
  const int esz = dtype == A2V_F32 ? 4 : 2;
  WgradArgs g;
  g.a = ConvA{A, a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, (Cv <= 0 || Cv > Cg) ? Cg : Cv, rowmask, bdiv < 1 ? 1 : bdiv, rowidx};
  int rc = check_conv(g.a, esz);
  if (rc) return rc;
  if (K != taps * Cg) return A2V_ERR_ARG;
  const int per16 = 16 / esz;
  if (N % per16 || ldp % per16 || p_goff % per16 || (((uintptr_t)P) & 15)) return A2V_ERR_ARG;
  g.P = P; g.ldp = ldp; g.p_goff = p_goff; g.W = W; g.ldw = ldw; g.w_gstride = w_gstride;
  g.M = M; g.N = N; g.K = K; g.G = G;
  if (dtype == A2V_BF16 && !rowidx && g_a2v_tn_p8 && a2v_launch_conv_strided_tn(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16 && !rowidx && g_a2v_tn_p8 && a2v_launch_tn_bf16_p8(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16 && !rowidx && rowmask && g_a2v_wgrad_sparse) {
    // first pos-conv layer: X is zero at the masked frames.  (The kernel behind it runs only if the sparse kernel's table builder
    // found a window it cannot take: WgradArgs::skip_if.)
    const int* ok = nullptr;
    if (a2v_launch_wgrad_sparse(g, 0, stream, &ok) == 1) {
      A2V_CHECK_LAUNCH();
      g.skip_if = ok;
    }
  }
  if (dtype == A2V_BF16 && !rowidx && g_a2v_wgrad_slab && a2v_launch_wgrad_slab(g, stream)) {
    A2V_CHECK_LAUNCH();
    return A2V_OK;
  }
  if (dtype == A2V_BF16 && rowidx && n_seq > 0 && g_a2v_wgrad_sparse) {
    // last pos-conv layer: dY exists at the kept frames only.  The sparse kernel needs an ascending row list; its table builder
    // checks that on the device, and the gathered TN kernel below runs only if the check failed.
    const int* ok = nullptr;
    if (a2v_launch_wgrad_sparse(g, n_seq, stream, &ok) == 2) {
      A2V_CHECK_LAUNCH();
      g.skip_if = ok;
    }
  }
  const int tiles = dtype == A2V_BF16 ? a2v_tn_bf16_tiles(N, K) : cdiv(N, 128) * cdiv(K, 128);
  const int step = dtype == A2V_BF16 ? TN_MSTEP : 16;
  // split M so that the grid has ~>= 1024 workgroups, each split a multiple of the M step
  int splits = 1024 / (tiles * G);
  if (splits < 1) splits = 1;
  int max_splits = cdiv(M, step * 4);
  if (max_splits < 1) max_splits = 1;
  if (splits > max_splits) splits = max_splits;
  int chunk = cdiv(cdiv(M, splits), step) * step;
  splits = cdiv(M, chunk);
  g.m_chunk = chunk;
  dim3 grid(tiles, G, splits);
  if (dtype == A2V_BF16) {
    a2v_launch_tn_bf16_fast(g, splits, stream);
  } else {
    hipLaunchKernelGGL(gemm_tn_f32_kernel, grid, dim3(256), 64 * TNF_PITCH * 4, stream, g);
  }
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

extern "C" int a2v_conv_wgrad(int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                              int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                              const void* P, long long ldp,
                              int p_goff, float* W, long long ldw, long long w_gstride, int M, int N, int K, int G,
                              hipStream_t stream) {
  return conv_wgrad_impl(nullptr, 0, dtype, A, a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, Cv, rowmask, bdiv, P, ldp, p_goff, W, ldw, w_gstride, M, N, K, G, stream);
}
extern "C" int a2v_conv_wgrad_rows(const int* rowidx, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                              int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                              const void* P, long long ldp,
                              int p_goff, float* W, long long ldw, long long w_gstride, int M, int N, int K, int G,
                              hipStream_t stream) {
  return conv_wgrad_impl(rowidx, 0, dtype, A, a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, Cv, rowmask, bdiv, P, ldp, p_goff, W, ldw, w_gstride, M, N, K, G, stream);
}
extern "C" int a2v_conv_wgrad_kept(const int* rowidx, int n_seq, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                              int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                              const void* P, long long ldp,
                              int p_goff, float* W, long long ldw, long long w_gstride, int M, int N, int K, int G,
                              hipStream_t stream) {
  if (!rowidx || n_seq <= 0 || M % n_seq) return A2V_ERR_ARG;
  return conv_wgrad_impl(rowidx, n_seq, dtype, A, a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, Cv, rowmask, bdiv, P, ldp, p_goff, W, ldw, w_gstride, M, N, K, G, stream);
}

// Two Linear weight gradients over the same M rows (gw_a[Na,Ka] += dy_a^T x_a, gw_b[Nb,Kb] += dy_b^T x_b) in one launch
// when the 256x256 TN kernel takes both (fewer M chunks -> fewer fp32 atomic adds per element); otherwise two launches.
extern "C" int a2v_linear_wgrad_pair(int dtype, int M, const void* xa, long long lda, const void* dya, long long ldpa, float* gwa,
                                     long long ldwa, int Na, int Ka, const void* xb, long long ldb_, const void* dyb,
                                     long long ldpb, float* gwb, long long ldwb, int Nb, int Kb, hipStream_t stream) {
  if (dtype == A2V_BF16 && g_a2v_tn_p8 && M > 0) {
    WgradArgs a, b;
    a.a = ConvA{xa, lda, 0, 0, M, M, 1, 0, 1, Ka, Ka, nullptr, 1, nullptr};
    a.P = dya; a.ldp = ldpa; a.p_goff = 0; a.W = gwa; a.ldw = ldwa; a.w_gstride = 0; a.M = M; a.N = Na; a.K = Ka; a.G = 1; a.m_chunk = 0;
    b.a = ConvA{xb, ldb_, 0, 0, M, M, 1, 0, 1, Kb, Kb, nullptr, 1, nullptr};
    b.P = dyb; b.ldp = ldpb; b.p_goff = 0; b.W = gwb; b.ldw = ldwb; b.w_gstride = 0; b.M = M; b.N = Nb; b.K = Kb; b.G = 1; b.m_chunk = 0;
    const int per16 = 8;
    const bool ok = !(Na % per16 || ldpa % per16 || lda % per16 || Nb % per16 || ldpb % per16 || ldb_ % per16 ||
                      ((((uintptr_t)xa) | ((uintptr_t)dya) | ((uintptr_t)xb) | ((uintptr_t)dyb)) & 15));
    if (ok && a2v_launch_tn_bf16_p8_pair(&a, &b, stream)) {
      A2V_CHECK_LAUNCH();
      return A2V_OK;
    }
  }
  int rc = conv_wgrad_impl(nullptr, 0, dtype, xa, lda, 0, 0, M, M, 1, 0, 1, Ka, 0, nullptr, 1, dya, ldpa, 0, gwa, ldwa, 0, M, Na, Ka, 1, stream);
  if (rc) return rc;
  return conv_wgrad_impl(nullptr, 0, dtype, xb, ldb_, 0, 0, M, M, 1, 0, 1, Kb, 0, nullptr, 1, dyb, ldpb, 0, gwb, ldwb, 0, M, Nb, Kb, 1, stream);
}

// ---- per-stream workspaces (a2v_common.h)
#include <mutex>
#include <vector>
// (256-byte aligned like the workspaces a caller binds: u32 atomic counters at offset 0, fp32 regions written as 16-B vectors behind them)
static_assert(A2V_WS_BYTES % 256 == 0 && A2V_WS_SPREAD % 16 == 0 && A2V_WS_QKB % 16 == 0 && A2V_WS_DAL % 16 == 0 && A2V_WS_CSPART % 16 == 0,
              "workspace layout: every region 16-byte aligned, the blob a multiple of 256 bytes");
__device__ __attribute__((aligned(256))) char g_ws_static[A2V_MAX_STREAMS][A2V_WS_BYTES];      // the library-owned sets: zero at load, self-cleaning afterwards
static std::mutex g_ws_mu;
static hipStream_t g_ws_stream[A2V_MAX_STREAMS];
static bool g_ws_used[A2V_MAX_STREAMS];
struct WsBound { hipStream_t stream; int device; char* ptr; };
static std::vector<WsBound> g_ws_bound;                           // caller-owned workspaces (a2v_bind_workspace)
char* a2v_stream_ws(hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  const int dev = a2v_device();
  for (const WsBound& b : g_ws_bound)
    if (b.stream == stream && b.device == dev) return b.ptr;
  static char* base_dev[A2V_MAX_DEVICES]; char*& base = base_dev[dev];        // address of the device symbol: looked up once per device
  if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_ws_static)) != hipSuccess) return nullptr;
  for (int i = 0; i < A2V_MAX_STREAMS; ++i)
    if (g_ws_used[i] && g_ws_stream[i] == stream) return base + (size_t)i * A2V_WS_BYTES;
  for (int i = 0; i < A2V_MAX_STREAMS; ++i)
    if (!g_ws_used[i]) { g_ws_used[i] = true; g_ws_stream[i] = stream; return base + (size_t)i * A2V_WS_BYTES; }
  return nullptr;
}
extern "C" int a2v_stream_release(hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (int i = 0; i < A2V_MAX_STREAMS; ++i)
    if (g_ws_used[i] && g_ws_stream[i] == stream) { g_ws_used[i] = false; return A2V_OK; }
  return A2V_ERR_ARG;
}
extern "C" size_t a2v_query_workspace(void) { return (size_t)A2V_WS_BYTES; }
extern "C" int a2v_bind_workspace(hipStream_t stream, void* ws, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  const int dev = a2v_device();
  for (size_t i = 0; i < g_ws_bound.size(); ++i)
    if (g_ws_bound[i].stream == stream && g_ws_bound[i].device == dev) {
      if (!ws) { g_ws_bound.erase(g_ws_bound.begin() + (long)i); return A2V_OK; }
      if (bytes < (size_t)A2V_WS_BYTES || (((uintptr_t)ws) & 255)) return A2V_ERR_ARG;
      g_ws_bound[i].ptr = (char*)ws;
      return A2V_OK;
    }
  if (!ws) return A2V_ERR_ARG;                                   // nothing bound to this stream
  if (bytes < (size_t)A2V_WS_BYTES || (((uintptr_t)ws) & 255)) return A2V_ERR_ARG;
  g_ws_bound.push_back(WsBound{stream, dev, (char*)ws});
  return A2V_OK;
}

// debugging / A-B benchmarking knob: key 0 = bf16 NT kernel variant (0 register-staged, 1 LDS-DMA)
extern int g_a2v_gemm_variant;
extern int g_a2v_p8_ih;
extern int g_a2v_p8_fx;
extern int g_a2v_conv_strided;
extern int g_a2v_p8_tail;
extern int g_a2v_p8_tmode;
extern int g_a2v_wgrad_slab_split;
extern int g_a2v_p8_stagger;
extern int g_a2v_p8_dynamic;
extern int g_a2v_attn_long;
extern int g_a2v_p8_direct;
extern int g_a2v_conv_dec;
extern int g_a2v_conv_slab2;
extern int g_a2v_attn_seq48;
extern int g_a2v_attn_fwd48;
extern int g_a2v_attn_xcd;
extern int g_a2v_attn_msum;
extern int g_a2v_attn_band;
extern int g_a2v_attn_fwd32;
extern int g_a2v_attn_seq2;
extern "C" int a2v_debug_set(int key, int value) {
  if (key == 0) { g_a2v_gemm_variant = value; return A2V_OK; }
  if (key == 1) { g_a2v_gemm_p8 = value; return A2V_OK; }     // 256x256 8-phase kernel on / off
  if (key == 2) { g_a2v_wgrad_slab = value; return A2V_OK; }  // slab weight-gradient kernel on / off
  if (key == 3) { g_a2v_conv_slab = value; return A2V_OK; }   // slab grouped-conv kernel on / off
  if (key == 4) { g_a2v_tn_p8 = value; return A2V_OK; }       // 256x256 TN weight-gradient kernel on / off
  if (key == 8) { g_a2v_attn_long = value; return A2V_OK; }        // long-sequence attention backward: two-pass kernel on / off
  if (key == 7) { g_a2v_p8_dynamic = value; return A2V_OK; }  // persistent kernels: dynamic tile hand-out on / off
  if (key == 6) { g_a2v_p8_stagger = value; return A2V_OK; }  // 8-phase kernels: stagger the two wave groups on / off
  if (key == 20) { g_a2v_p8_fx = value; return A2V_OK; }    // 0: no fixed-flavour p8 epilogues
  if (key == 21) { g_a2v_conv_strided = value; return A2V_OK; }
  if (key == 22) { g_a2v_p8_tail = value; return A2V_OK; }        // persistent NT kernel: nearly empty last round -> 128 x 128 kernel   // dense strided convs on the 8-phase GEMMs on / off
  if (key == 40) { g_a2v_wgrad_slab_split = value; return A2V_OK; }   // pos-conv weight gradient: kk tiles over two workgroups per CU on / off
  if (key == 23) { g_a2v_p8_tmode = value; return A2V_OK; }   // persistent NT kernel: tile order (0: 8 m-tiles x all n-tiles, 1: W-stationary n-groups)
#ifdef A2V_W2      // experiment library only (python -m animal2vec_amd.build --variant w2 -DA2V_W2): profiles/r04_w2_experiment.txt
  { extern int g_a2v_gemm_w2; if (key == 25) { g_a2v_gemm_w2 = value; return A2V_OK; } }   // 256 x 128 two-per-CU NT kernel: 0 off, 1 K <= 1024, 2 all
#endif
  { extern int g_a2v_ln_stream, g_a2v_ln_rows, g_a2v_ln_wpc;   // streaming LayerNorm (a2v_norm.hip): on / off bits, rows per batch, workgroups per CU
    if (key == 30) { g_a2v_ln_stream = value; return A2V_OK; }
    if (key == 31) { g_a2v_ln_rows = value; return A2V_OK; }
    if (key == 32) { g_a2v_ln_wpc = value; return A2V_OK; }
    extern int g_a2v_ln_bpw, g_a2v_ln_bpw_bwd;
    if (key == 33) { g_a2v_ln_bpw = value; return A2V_OK; }
    if (key == 34) { g_a2v_ln_bpw_bwd = value; return A2V_OK; } }
  { extern int g_a2v_slab_order; if (key == 35) { g_a2v_slab_order = value; return A2V_OK; } }   // grouped-conv launch orders (a2v_conv_slab.hip)
  if (key == 5) { g_a2v_p8_ih = value; return A2V_OK; }       // force the p8 tile height (2 / 3 / 4; 0 = auto)
  if (key == 14) { g_a2v_conv_slab2 = value; return A2V_OK; } // 384-row two-per-CU slab kernel on / off
  if (key == 13) { g_a2v_conv_dec = value; return A2V_OK; }   // decoder grouped-conv kernel (a2v_conv_dec.hip) on / off
  { extern int g_a2v_dec_bpw; if (key == 38) { g_a2v_dec_bpw = value; return A2V_OK; } }   // conv_dec_kernel: 64-row blocks per workgroup (0 auto)
  if (key == 37) { g_a2v_wgrad_sparse = value; return A2V_OK; }  // row-sparse pos-conv weight gradients on / off
  if (key == 39) { g_a2v_attn_seq2 = value; return A2V_OK; }   // S <= 160 bf16 attention forward + backward: round-6 five-wave kernels on / off
  if (key == 36) { g_a2v_attn_fwd32 = value; return A2V_OK; }  // long-sequence attention forward: 32 x 32 x 16 kernel on / off
  if (key == 19) { g_a2v_attn_band = value; return A2V_OK; }  // long-sequence attention forward: ALiBi band limit on / off
  if (key == 18) { g_a2v_attn_msum = value; return A2V_OK; }  // forward-only long-sequence attention: P row sums on the matrix cores on / off
  if (key == 17) { g_a2v_attn_xcd = value; return A2V_OK; }   // long-sequence attention forward: XCD-aware workgroup order on / off
  if (key == 16) { g_a2v_attn_fwd48 = value; return A2V_OK; } // S <= 144 attention forward: whole-sequence three-wave kernel on / off
  if (key == 15) { g_a2v_attn_seq48 = value; return A2V_OK; } // S <= 144 attention backward: 48-row three-wave kernel on / off
  if (key == 10) { g_a2v_p8_direct = value; return A2V_OK; }  // p8 NT kernel: direct-store epilogue on / off
  return A2V_ERR_ARG;
}
