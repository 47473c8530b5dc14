// bf16 throughput kernels of the implicit-GEMM family (see a2v_gemm.hip for the contract).
//
// What differs from the general kernels: the operand loaders are branch-free and almost VALU-free.
//   * eligible shapes have Cg % 64 == 0, so a 64-wide K tile never straddles two conv taps: the tap
//     index and the byte offset inside the source row are wave-uniform scalars that advance with a
//     couple of SALU ops per K tile; each thread keeps one 64-bit base pointer per staged row.
//   * out-of-range rows (conv padding, masked clone rows) read a safe address and are zeroed with
//     v_cndmask instead of being branched around; M/N tails read a clamped valid row (the epilogue
//     drops them), so plain Linear layers run with no predicate at all.
//   * tile shapes: 128x128 (2x2 waves) for wide N, 256x64 (4x1 waves) for the grouped convolutions whose
//     per-group N is 64 / 48 -- no half-empty MFMA tiles.
#include "a2v_gemm.h"

// component-wise select (a vector ?: on uint4 is lowered through scratch memory by hipcc)
__device__ __forceinline__ uint4 sel4(bool ok, uint4 v) {
  uint4 r;
  r.x = ok ? v.x : 0u; r.y = ok ? v.y : 0u; r.z = ok ? v.z : 0u; r.w = ok ? v.w : 0u;
  return r;
}

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_nt_bf16_fast_kernel(GemmArgs g) {
  constexpr int WN = BN / 64, WM = 4 / WN;
  static_assert(BM == 64 * WM, "wave tile is 64x64");
  constexpr int A_CH = BM / 32, B_CH = BN / 32;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  int tm, tn;
  tile_coords(id, tiles_m, tiles_n, tm, tn);
  const int grp = blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;
  const int crow = tid >> 3, cchunk = tid & 7;
  const ConvA& a = g.a;
  const char* const safe = (const char*)a.ptr;

  // Staging state lives in NAMED scalars (not arrays): arrays here end up in scratch memory once the
  // schedule barriers below are present.
#define A2V_REP4(X) X(0) X(1) X(2) X(3)
#define A2V_REP48(X) X(4) X(5) X(6) X(7)
#define DECL_A(i) const char* pa##i = safe; int ta##i = 0; const unsigned char* ma##i = nullptr; \
                  uint4 ra##i##_0 = {0u, 0u, 0u, 0u}, ra##i##_1 = {0u, 0u, 0u, 0u};
#define DECL_B(i) const char* pb##i = safe; uint4 rb##i##_0 = {0u, 0u, 0u, 0u}, rb##i##_1 = {0u, 0u, 0u, 0u};
  A2V_REP4(DECL_A) A2V_REP48(DECL_A) A2V_REP4(DECL_B)
#define INIT_A(i)                                                                                             \
  if (i < A_CH) {                                                                                             \
    int m = m0 + crow + 32 * i;                                                                               \
    m = m < g.M ? m : g.M - 1;                                                                                \
    if (a.rowidx) m = a.rowidx[m];                                                                            \
    const int b = m / a.T_out, t = m - b * a.T_out;                                                           \
    ta##i = t * a.stride - a.pad;                                                                             \
    pa##i = (const char*)a.ptr + ((long long)(b / a.bdiv) * a.bstride + (long long)grp * a.goff + cchunk * 8 + \
                                  (long long)ta##i * a.pitch) * 2;                                            \
    ma##i = a.rowmask ? a.rowmask + (long long)b * a.T_in : nullptr;                                          \
  }
#define INIT_B(i)                                                                                             \
  if (i < B_CH) {                                                                                             \
    int n = n0 + crow + 32 * i;                                                                               \
    n = n < g.N ? n : g.N - 1;                                                                                \
    pb##i = (const char*)g.B + ((long long)grp * g.b_gstride + (long long)n * g.ldb + cchunk * 8) * 2;        \
  }
  A2V_REP4(INIT_A) A2V_REP48(INIT_A) A2V_REP4(INIT_B)
  long long soffA = 0, soffB = 0;     // wave-uniform byte offsets of the current K tile
  int tap = 0, ci0 = 0;
  const long long tap_step = (a.pitch - (a.Cg - 64)) * 2;

#define LOAD_A(i, S)                                                           \
  if (i < A_CH) {                                                              \
    if constexpr (CONV) {                                                      \
      const int row = ta##i + tap;                                             \
      bool ok = (unsigned)row < (unsigned)a.T_in && (ci0 + cchunk * 8 < a.Cv); \
      if (a.rowmask) ok = ok && (ma##i[ok ? row : 0] == 0);                    \
      const char* p = ok ? pa##i + soffA : safe;                               \
      ra##i##_##S = sel4(ok, *(const uint4*)p);                                \
    } else {                                                                   \
      ra##i##_##S = *(const uint4*)(pa##i + soffA);                            \
    }                                                                          \
  }
#define LOAD_B(i, S) if (i < B_CH) rb##i##_##S = *(const uint4*)(pb##i + soffB);
#define LOAD_TILE(S)                                                           \
  {                                                                            \
    LOAD_A(0, S) LOAD_A(1, S) LOAD_A(2, S) LOAD_A(3, S) LOAD_A(4, S) LOAD_A(5, S) LOAD_A(6, S) LOAD_A(7, S) \
    LOAD_B(0, S) LOAD_B(1, S) LOAD_B(2, S) LOAD_B(3, S)                        \
    if (kt_next + 1 < nk) {      /* wave-uniform: stay on the last tile instead of running past K */ \
      soffB += 128;                                                            \
      ci0 += 64;                                                               \
      if (ci0 == a.Cg) { ci0 = 0; ++tap; soffA += tap_step; } else { soffA += 128; } \
    }                                                                          \
    ++kt_next;                                                                 \
  }
#define STORE_A(i, S) if (i < A_CH) *(uint4*)(sa_ + swz128(crow + 32 * i, cchunk)) = ra##i##_##S;
#define STORE_B(i, S) if (i < B_CH) *(uint4*)(sa_ + A_BYTES + swz128(crow + 32 * i, cchunk)) = rb##i##_##S;
#define STORE_TILE(buf_, S)                                                    \
  {                                                                            \
    char* sa_ = smem + (buf_) * STAGE;                                         \
    STORE_A(0, S) STORE_A(1, S) STORE_A(2, S) STORE_A(3, S) STORE_A(4, S) STORE_A(5, S) STORE_A(6, S) STORE_A(7, S) \
    STORE_B(0, S) STORE_B(1, S) STORE_B(2, S) STORE_B(3, S)                    \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / 64;
  int kt_next = 0;               // index of the tile the next LOAD_TILE() fetches
  const int fr = lane & 15, fq = lane >> 4;
#define COMPUTE(buf_)                                                                                   \
  {                                                                                                     \
    const char* sa_ = smem + (buf_) * STAGE;                                                            \
    const char* sb_ = sa_ + A_BYTES;                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                  \
      bf16x8 fa[4], fb[4];                                                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
          fa[i] = *(const bf16x8*)(sa_ + swz128(wm * 64 + i * 16 + fr, ks * 4 + fq));                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
          fb[j] = *(const bf16x8*)(sb_ + swz128(wn * 64 + j * 16 + fr, ks * 4 + fq));                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
          acc[i][j] = A2V_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);        \
    }                                                                                                   \
  }
  // Two register sets: the loads of tile kt+2 are issued before tile kt is computed and are only written to
  // LDS at the end of iteration kt+1, so a global load has ~2 K-tiles of matrix work to land
  // (one K-tile was not enough: SQ_WAIT_ANY was 27 % of wave time with a single set).
  LOAD_TILE(0);                  // tile 0
  STORE_TILE(0, 0);
  LOAD_TILE(1);                  // tile 1
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    LOAD_TILE(0);                // tile kt+2 (clamped to the last tile at the tail; never consumed there)
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE(0);
    STORE_TILE(1, 1);            // tile kt+1
    __syncthreads();
    if (kt + 1 >= nk) break;
    LOAD_TILE(1);                // tile kt+3
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE(1);
    STORE_TILE(0, 0);            // tile kt+2
    __syncthreads();
  }
#undef COMPUTE
#undef LOAD_TILE
#undef STORE_TILE
#undef LOAD_A
#undef LOAD_B
#undef STORE_A
#undef STORE_B
#undef INIT_A
#undef INIT_B
#undef DECL_A
#undef DECL_B

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

// =============================================================================================
//   bf16 NT with direct-to-LDS staging (global_load_lds_dwordx4): no VGPR round trip, no ds_write.
//   Measured on the register-staged kernel: the VGPR->LDS store path (ds_write_b128 = 13 cycles per KiB per
//   wave-instruction, ~79 B/clk/CU) plus fragment reads kept the LDS pipe busier than the MFMA pipe
//   (MFMA busy 39 %).  LDS-DMA writes a wave-instruction's 1 KiB (8 rows x 128 B) linearly, so the XOR
//   swizzle of the 16-B chunks is applied to the per-lane SOURCE address instead (same involution as
//   the ds_read side).  Conv / masked A operands (per-row predicates) keep the register path; weights
//   always go through LDS-DMA.
// =============================================================================================
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
__device__ __forceinline__ void glds16(const char* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)gsrc, (lds_void_t*)lds_dst, 16, 0, 0);
}

// OutT = float: bf16 (split) operands with fp32 C / C2 / R -- dtype A2V_BF16X3 for the shapes the persistent kernel does not take
// (convolutions: implicit im2col of the split activation image, see a2v_split.hip)
template <int BM, int BN, bool CONV, typename OutT = bf16>
__global__ __launch_bounds__(256) void gemm_nt_bf16_glds_kernel(GemmArgs g) {
  constexpr int WN = BN / 64, WM = 4 / WN;
  static_assert(BM == 64 * WM, "wave tile is 64x64");
  constexpr int A_CH = BM / 32;                         // register path: chunks per thread
  constexpr int A_SL = BM / 32, B_SL = BN / 32;         // LDS-DMA path: 8-row slabs per wave
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  int tm, tn;
  tile_coords(id, tiles_m, tiles_n, tm, tn);
  const int grp = blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;
  const int crow = tid >> 3, cchunk = tid & 7;
  const ConvA& a = g.a;
  const char* const safe = (const char*)a.ptr;
  const int lr = lane >> 3;                             // row inside an 8-row slab
  // source chunk of LDS chunk position (lane & 7) in row r = slab*8 + lr:  c = pos ^ ((r >> 1) & 7)
#define SRC_CHUNK(slab) (((lane & 7) ^ ((((slab) & 1) << 2) + (lane >> 4))) & 7)

#define A2V_REP4(X) X(0) X(1) X(2) X(3)
#define A2V_REP48(X) X(4) X(5) X(6) X(7)
#define DECL_A(i) const char* pa##i = safe; int ta##i = 0; const unsigned char* ma##i = nullptr; uint4 ra##i = {0u, 0u, 0u, 0u};
#define DECL_B(i) const char* pb##i = safe;
  A2V_REP4(DECL_A) A2V_REP48(DECL_A) A2V_REP4(DECL_B)
#define INIT_A(i)                                                                                             \
  if (i < (CONV ? A_CH : A_SL)) {                                                                             \
    const int slab = wid + 4 * i;                                                                             \
    int m = CONV ? m0 + crow + 32 * i : m0 + slab * 8 + lr;                                                   \
    m = m < g.M ? m : g.M - 1;                                                                                \
    if (a.rowidx) m = a.rowidx[m];                                                                            \
    const int b = m / a.T_out, t = m - b * a.T_out;                                                           \
    ta##i = t * a.stride - a.pad;                                                                             \
    const int ch = CONV ? cchunk : SRC_CHUNK(slab);                                                           \
    pa##i = (const char*)a.ptr + ((long long)(b / a.bdiv) * a.bstride + (long long)grp * a.goff + ch * 8 +    \
                                  (long long)ta##i * a.pitch) * 2;                                            \
    ma##i = a.rowmask ? a.rowmask + (long long)b * a.T_in : nullptr;                                          \
  }
#define INIT_B(i)                                                                                             \
  if (i < B_SL) {                                                                                             \
    const int slab = wid + 4 * i;                                                                             \
    int n = n0 + slab * 8 + lr;                                                                               \
    n = n < g.N ? n : g.N - 1;                                                                                \
    pb##i = (const char*)g.B + ((long long)grp * g.b_gstride + (long long)n * g.ldb + SRC_CHUNK(slab) * 8) * 2; \
  }
  A2V_REP4(INIT_A) A2V_REP48(INIT_A) A2V_REP4(INIT_B)
  long long soffA = 0, soffB = 0;
  int tap = 0, ci0 = 0;
  const long long tap_step = (a.pitch - (a.Cg - 64)) * 2;

#define LOAD_A(i)                                                              \
  if (i < (CONV ? A_CH : A_SL)) {                                              \
    if constexpr (CONV) {                                                      \
      const int row = ta##i + tap;                                             \
      bool ok = (unsigned)row < (unsigned)a.T_in && (ci0 + cchunk * 8 < a.Cv); \
      if (a.rowmask) ok = ok && (ma##i[ok ? row : 0] == 0);                    \
      const char* p = ok ? pa##i + soffA : safe;                               \
      ra##i = sel4(ok, *(const uint4*)p);                                      \
    } else {                                                                   \
      glds16(pa##i + soffA, sdst_ + (wid + 4 * i) * 1024);                     \
    }                                                                          \
  }
#define LOAD_B(i) if (i < B_SL) glds16(pb##i + soffB, sdst_ + A_BYTES + (wid + 4 * i) * 1024);
#define LOAD_TILE(buf_)                                                        \
  {                                                                            \
    char* sdst_ = smem + (buf_) * STAGE;                                       \
    A2V_REP4(LOAD_B) A2V_REP4(LOAD_A) A2V_REP48(LOAD_A)                        \
    if (kt_next + 1 < nk) {                                                    \
      soffB += 128;                                                            \
      ci0 += 64;                                                               \
      if (ci0 == a.Cg) { ci0 = 0; ++tap; soffA += tap_step; } else { soffA += 128; } \
    }                                                                          \
    ++kt_next;                                                                 \
  }
#define STORE_A(i) if (i < A_CH) *(uint4*)(sa_ + swz128(crow + 32 * i, cchunk)) = ra##i;
#define STORE_TILE(buf_)                                                       \
  if constexpr (CONV) {                                                        \
    char* sa_ = smem + (buf_) * STAGE;                                         \
    A2V_REP4(STORE_A) A2V_REP48(STORE_A)                                       \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / 64;
  int kt_next = 0;
  LOAD_TILE(0);
  STORE_TILE(0);
  __syncthreads();               // also drains the LDS-DMA (vmcnt(0))

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    LOAD_TILE(buf ^ 1);          // tile kt+1 straight into the other stage (the tail re-fetches tile nk-1)
    if constexpr (CONV) __builtin_amdgcn_sched_barrier(0);
    const char* sa_ = smem + buf * STAGE;
    const char* sb_ = sa_ + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8*)(sa_ + swz128(wm * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(sb_ + swz128(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = A2V_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    STORE_TILE(buf ^ 1);
    __syncthreads();
  }
#undef LOAD_TILE
#undef STORE_TILE
#undef LOAD_A
#undef LOAD_B
#undef STORE_A
#undef INIT_A
#undef INIT_B
#undef DECL_A
#undef DECL_B
#undef SRC_CHUNK

  // epilogue through a per-wave LDS stage (A/B test: a register epilogue with swapped MFMA operands and 8-B
  // stores was 3-12 % slower -- 32-B store segments and +100 VGPRs)
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
  wave_epilogue<OutT>(g, stage, lane, m0 + wm * 64, n0 + wn * 64, grp);
}

int g_a2v_gemm_variant = 1;      // 0: register-staged kernel, 1: LDS-DMA kernel (a2v_debug_set(0, v))

template <int BM, int BN, bool CONV, typename OutT = bf16>
static void launch_nt_glds(const GemmArgs& g, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int LDS = 2 * STAGE > NT_LDS_BYTES ? 2 * STAGE : NT_LDS_BYTES;
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_glds_kernel<BM, BN, CONV, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  dim3 grid(cdiv(g.M, BM) * cdiv(g.N, BN), g.G, 1);
  hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<BM, BN, CONV, OutT>), grid, dim3(256), LDS, stream, g);
}

// bf16 split operands, fp32 outputs (A2V_BF16X3), any shape with 64-wide taps
bool a2v_launch_nt_x3_fast(const GemmArgs& g, hipStream_t stream) {
  if (g.a.Cg % 64 != 0 || g.K % 64 != 0) return false;
  if (g.N <= 64) launch_nt_glds<256, 64, true, float>(g, stream); else launch_nt_glds<128, 128, true, float>(g, stream);
  return true;
}

template <int BM, int BN, bool CONV>
static void launch_nt(const GemmArgs& g, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int LDS = 2 * STAGE > NT_LDS_BYTES ? 2 * STAGE : NT_LDS_BYTES;
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_fast_kernel<BM, BN, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  dim3 grid(cdiv(g.M, BM) * cdiv(g.N, BN), g.G, 1);
  hipLaunchKernelGGL((gemm_nt_bf16_fast_kernel<BM, BN, CONV>), grid, dim3(256), LDS, stream, g);
}

bool a2v_launch_nt_bf16_fast(const GemmArgs& g, hipStream_t stream) {
  if (g.a.Cg % 64 != 0 || g.K % 64 != 0) return false;
  const bool conv = g.a.taps > 1 || g.a.pad != 0 || g.a.rowmask != nullptr || g.a.rowidx != nullptr || g.a.stride != 1 ||
                    g.a.T_in != g.a.T_out ||
                    g.a.Cv != g.a.Cg;
  if (g_a2v_gemm_variant == 1) {
    if (g.N <= 64) {
      if (conv) launch_nt_glds<256, 64, true>(g, stream); else launch_nt_glds<256, 64, false>(g, stream);
    } else {
      if (conv) launch_nt_glds<128, 128, true>(g, stream); else launch_nt_glds<128, 128, false>(g, stream);
    }
    return true;
  }
  if (g.N <= 64) {
    if (conv) launch_nt<256, 64, true>(g, stream); else launch_nt<256, 64, false>(g, stream);
  } else {
    if (conv) launch_nt<128, 128, true>(g, stream); else launch_nt<128, 128, false>(g, stream);
  }
  return true;
}

// =============================================================================================
//      bf16 TN (weight gradient): incremental row bookkeeping, branch-free loads,
//      tiles 128(n) x 128(kk)  or  64(n) x 256(kk) (grouped convs with <= 64 outputs per group)
// =============================================================================================
template <int ROWB> __device__ __forceinline__ int tr_off(int r, int col) {   // byte offset of element (r, col)
  const int blk = col >> 4;
  const int f = ROWB == 128 ? (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) : tn_f(r);
  return r * ROWB + ((blk ^ f) << 5) + ((col & 15) << 1);
}

template <int TNN, int TNK>   // (128,128) or (64,128): 4 waves as 2(n) x 2(kk), wave tile (TNN/2) x 64
__global__ __launch_bounds__(256) void gemm_tn_bf16_fast_kernel(WgradArgs g, int splits) {
  constexpr int PCH = TNN / 8, QCH = TNK / 8;          // 16-B chunks per tile row
  constexpr int NP = PCH / 4, NQ = QCH / 4;            // staged rows per thread
  constexpr int PROWS = 256 / PCH, QROWS = 256 / QCH;  // rows covered per pass
  constexpr int P_BYTES = 64 * TNN * 2, Q_BYTES = 64 * TNK * 2, STAGE = P_BYTES + Q_BYTES;
  constexpr int WK = 2;                                // waves along kk
  constexpr int WNN = TNN / 2;                         // n extent per wave
  constexpr int TI = WNN / 16;                         // 16-wide n tiles per wave
  static_assert(TNK == 128, "kk tile is 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (g.skip_if && *g.skip_if) return;                 // (launched behind conv_wgrad_sparse_kernel: only runs if that one could not)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WK, wn = wid % WK;
  const int tiles_k = (g.K + TNK - 1) / TNK;
  // 1-D grid, XCD-aware: the tiles of one (M split, group) are contiguous logical ids, so they run on ONE XCD
  // at about the same time and share the dY / x row slabs out of its L2 (a 3-D grid dealt them round-robin over
  // the 8 XCDs: the pos-conv wgrad pulled 5.2 GB per launch through the fabric for 0.8 GB of operands).
  const int tiles_nn = (g.N + TNN - 1) / TNN;
  const int tiles = tiles_nn * tiles_k;
  const int lid = xcd_remap(blockIdx.x, tiles * g.G * splits);
  const int split = lid / (tiles * g.G);
  const int lrem = lid - split * tiles * g.G;
  const int grp = lrem / tiles;
  int tn_, tk;
  tile_coords(lrem - grp * tiles, tiles_nn, tiles_k, tn_, tk);
  const int n0 = tn_ * TNN, k0 = tk * TNK;
  const int mbeg = split * g.m_chunk;
  const int mend = min(g.M, mbeg + g.m_chunk);
  const int nsteps = (mend - mbeg + 63) / 64;
  if (nsteps <= 0) return;
  const ConvA& a = g.a;
  const char* const safe = (const char*)a.ptr;

  // P operand (dY): fixed column chunk per thread, rows advance by 64 per step
  const int prow = tid / PCH, pchunk = tid % PCH;
  const int pn = n0 + pchunk * 8;
  const bool pn_ok = pn < g.N;
  const char* pp[NP];
  int mp[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    mp[i] = mbeg + prow + PROWS * i;
    pp[i] = (const char*)g.P + ((long long)mp[i] * g.ldp + (long long)grp * g.p_goff + (pn_ok ? pn : 0)) * 2;
  }
  // Q operand (implicit im2col): fixed (tap, channel) per thread, rows advance
  const int qrow = tid / QCH, qchunk = tid % QCH;
  const int kk0 = k0 + qchunk * 8;
  const int tapq = kk0 < g.K ? kk0 / a.Cg : 0;
  const int ciq = kk0 < g.K ? kk0 - tapq * a.Cg : 0;
  const bool kk_ok = kk0 < g.K && ciq < a.Cv;
  int tq[NQ], rowq[NQ], mq[NQ];
  const char* qp[NQ];
  const unsigned char* mk[NQ];
#define TN_SETUP_ROW(i)                                                                                \
  {                                                                                                    \
    int mm = mq[i] < g.M ? mq[i] : g.M - 1;                                                            \
    if (a.rowidx) mm = a.rowidx[mm];                                                                   \
    const int b = mm / a.T_out;                                                                        \
    tq[i] = mm - b * a.T_out;                                                                          \
    rowq[i] = tq[i] * a.stride - a.pad + tapq;                                                         \
    qp[i] = (const char*)a.ptr +                                                                       \
            ((long long)(b / a.bdiv) * a.bstride + (long long)rowq[i] * a.pitch + (long long)grp * a.goff + ciq) * 2; \
    mk[i] = a.rowmask ? a.rowmask + (long long)b * a.T_in : nullptr;                                   \
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    mq[i] = mbeg + qrow + QROWS * i;
    TN_SETUP_ROW(i);
  }
  const long long p_step = 64LL * g.ldp * 2;
  const long long q_step = 64LL * a.stride * a.pitch * 2;
  uint4 rp[NP], rq[NQ];

#define LOAD_TILE()                                                                                   \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                  \
      const bool okp = pn_ok && mp[i] < mend;                                                         \
      const uint4 v = *(const uint4*)(okp ? pp[i] : safe);                                            \
      rp[i] = sel4(okp, v);                                                                           \
      pp[i] += p_step;                                                                                \
      mp[i] += 64;                                                                                    \
    }                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NQ; ++i) {                                                  \
      bool okq = kk_ok && mq[i] < mend && (unsigned)rowq[i] < (unsigned)a.T_in;                       \
      if (a.rowmask) okq = okq && (mk[i][okq ? rowq[i] : 0] == 0);                                    \
      const uint4 w = *(const uint4*)(okq ? qp[i] : safe);                                            \
      rq[i] = sel4(okq, w);                                                                           \
      mq[i] += 64;                                                                                    \
      tq[i] += 64;                                                                                    \
      rowq[i] += 64 * a.stride;                                                                       \
      qp[i] += q_step;                                                                                \
      if (a.rowidx || tq[i] >= a.T_out) TN_SETUP_ROW(i);   /* gathered rows, or crossed into the next batch item */ \
    }                                                                                                 \
  }
#define STORE_TILE(buf_)                                                                              \
  {                                                                                                   \
    char* sp_ = smem + (buf_) * STAGE;                                                                \
    char* sq_ = sp_ + P_BYTES;                                                                        \
    _Pragma("unroll") for (int i = 0; i < NP; ++i)                                                    \
        *(uint4*)(sp_ + tr_off<TNN * 2>(prow + PROWS * i, pchunk * 8)) = rp[i];                       \
    _Pragma("unroll") for (int i = 0; i < NQ; ++i)                                                    \
        *(uint4*)(sq_ + tr_off<TNK * 2>(qrow + QROWS * i, qchunk * 8)) = rq[i];                       \
  }

  f32x4 acc[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  LOAD_TILE();
  STORE_TILE(0);
  __syncthreads();

  const int gq = lane >> 4;
  const int q = (lane & 15) >> 2, p = lane & 3;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) LOAD_TILE();
    __builtin_amdgcn_sched_barrier(0);
    const char* sp_ = smem + buf * STAGE;
    const char* sq_ = sp_ + P_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[TI], fb[4];
      const int r_lo = ks * 32 + 8 * gq + q;
      const int r_hi = r_lo + 4;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int col = wm * WNN + i * 16 + 4 * p;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sp_ + tr_off<TNN * 2>(r_lo, col)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sp_ + tr_off<TNN * 2>(r_hi, col)));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        fa[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wn * 64 + j * 16 + 4 * p;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sq_ + tr_off<TNK * 2>(r_lo, col)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sq_ + tr_off<TNK * 2>(r_hi, col)));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        fb[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = A2V_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (st + 1 < nsteps) STORE_TILE(buf ^ 1);
    __syncthreads();
  }
#undef LOAD_TILE
#undef STORE_TILE
#undef TN_SETUP_ROW
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wm * WNN + i * 16 + fq * 4 + r;
      if (n >= g.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k0 + wn * 64 + j * 16 + fr;
        if (kk < g.K) atomicAdd(g.W + (long long)grp * g.w_gstride + (long long)n * g.ldw + kk, acc[i][j][r]);
      }
    }
}

template <int TNN, int TNK>
static void launch_tn(const WgradArgs& g, int splits, hipStream_t stream) {
  constexpr int LDS = 2 * (64 * TNN * 2 + 64 * TNK * 2);
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_fast_kernel<TNN, TNK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  dim3 grid(cdiv(g.N, TNN) * cdiv(g.K, TNK) * g.G * splits);
  hipLaunchKernelGGL((gemm_tn_bf16_fast_kernel<TNN, TNK>), grid, dim3(256), LDS, stream, g, splits);
}

// returns the number of (n, kk) tiles per group so that the caller can size the M split
int a2v_tn_bf16_tiles(int N, int K) { return N <= 64 ? cdiv(N, 64) * cdiv(K, 128) : cdiv(N, 128) * cdiv(K, 128); }

bool a2v_launch_tn_bf16_fast(const WgradArgs& g, int splits, hipStream_t stream) {
  if (g.N <= 64)
    launch_tn<64, 128>(g, splits, stream);
  else
    launch_tn<128, 128>(g, splits, stream);
  return true;
}
