// 256 x 256 x 64 bf16 TN GEMM for the weight gradients of the Linear layers, gfx950:
//
//     W[n][kk] += sum_m dY[m][n] * X[m][kk]        (fp32 accumulate into the flat gradient buffer)
//
// Same machine as a2v_gemm_p8.hip (8 waves, LDS-DMA half-tiles, counted vmcnt(6), persistent stream across work units) -- since round 6
// with TWO phases of 32 MFMAs per 64-row K-tile on 64 x 64 wave tiles (half the fragment reads of the four-phase form) -- with the
// reduction running over ROWS: both operands are staged
// row-major as [64 m][128 cols] half-tiles (256-B rows, 32-B blocks XOR-swizzled by f(m) = (m & 3) | ((m >> 3) & 1) << 2,
// the swizzle applied to the LDS-DMA source address) and every MFMA fragment is a ds_read_b64_tr_b16 pair.
// A work unit is (output tile, M chunk); units = tiles x ceil(CUs / tiles), a unit's result leaves with fp32
// atomics (the 128x128 kernel with its 1024 smaller units issued 4x the atomic traffic for the same output).
// Eligibility: plain row-major X (the Linear layers), N % 256 == 0, K % 256 == 0, at least 12 K-tiles per unit.
#include "a2v_gemm.h"

typedef __attribute__((address_space(3))) void t8_lds_void_t;
typedef __attribute__((address_space(1))) const void t8_gbl_cvoid_t;
__device__ __forceinline__ void t8_glds16(const char* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((t8_gbl_cvoid_t*)gsrc, (t8_lds_void_t*)lds_dst, 16, 0, 0);
}

// T8_EXP (timing builds, wrong results; tools/tn_probe.py): 1 no MFMA, 2 no LDS-DMA in the K loop, 3 no fragment reads, 4 no atomics
#ifndef T8_EXP
#define T8_EXP 0
#endif
#define T8_BUF 65536                  // one K-tile: P_lo, P_hi, Q_lo, Q_hi (16 KiB half-tiles: 64 rows x 256 B)
#define T8_PLO 0
#define T8_PHI 16384
#define T8_QLO 32768
#define T8_QHI 49152
#define T8_LDS (2 * T8_BUF)

__device__ uint4 g_t8_zero[1];        // rows past the end of M are fetched from here

struct T8Prob {       // one weight-gradient problem of a (possibly paired) launch; all share M
  const char* P;      // dY (M x N, row pitch ldp elements)
  const char* Q;      // X  (M x K, row pitch ldq elements)
  float* W;           // fp32 (N x K, row pitch ldw)
  long long ldp, ldq, ldw;
  int N, K, tiles;    // tiles = (N/256) * (K/256)
};
struct T8Args { T8Prob pr[2]; int nprob; int M; };

struct T8Unit {       // wave-uniform description of a work unit
  const char* pb;     // P + (mbeg * ldp + n0) * 2
  const char* qb;     // Q + (mbeg * ldq + k0) * 2
  float* wb;          // W + n0 * ldw + k0
  unsigned ldp2, ldq2;  // row pitches in bytes
  long long ldw;
  int nk;             // K-tiles (64 rows) in this unit (even, >= 2)
  int mbeg;           // first row of the unit
};

__global__ __launch_bounds__(512) void gemm_tn_bf16_p8_kernel(T8Args g, int chunks, int kt_per_chunk, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int tiles = g.pr[0].tiles + (g.nprob > 1 ? g.pr[1].tiles : 0);
  const int units = tiles * chunks;
  const char* const zsrc = (const char*)g_t8_zero;
  const int nk_total = (g.M + 63) / 64;

  // LDS position chunk c (16 B) of row r holds source chunk (((c >> 1) ^ f(r)) << 1) | (c & 1); this thread stages
  // rows mrow0 / mrow1 of every K-tile
  // (a macro, evaluated from a laundered copy of `lane` at the top of every work unit: as loop invariants these four values
  //  were live across the K loop, spilled at 256 registers, and their scratch reloads at each unit's start drew three
  //  s_waitcnt vmcnt(0) -- a full drain of the previous unit's 128 atomics per wave while the next unit's K-tiles were already staged)
#define T8_LANE_CONSTS(l_)                                                                                        \
  const int c16 = (l_) & 15;                                                                                      \
  const int mrow0 = wid * 4 + ((l_) >> 4), mrow1 = mrow0 + 32;                                                    \
  const int f0_ = (mrow0 & 3) | (((mrow0 >> 3) & 1) << 2), f1_ = (mrow1 & 3) | (((mrow1 >> 3) & 1) << 2);        \
  const unsigned sc0 = (unsigned)(((((c16 >> 1) ^ f0_) << 1) | (c16 & 1)) * 16);                                  \
  const unsigned sc1 = (unsigned)(((((c16 >> 1) ^ f1_) << 1) | (c16 & 1)) * 16);
  auto make_unit = [&](int seq) {
    T8Unit u;
    const int id = xcd_remap(seq < units ? seq : units - 1, units);
    const int chunk = id / tiles;
    int tile = id - chunk * tiles;
    const bool second = g.nprob > 1 && tile >= g.pr[0].tiles;
    if (second) tile -= g.pr[0].tiles;
    const T8Prob& pr = g.pr[second ? 1 : 0];
    const int tiles_k = pr.K / 256;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int kt0 = chunk * kt_per_chunk;
    u.nk = min(kt_per_chunk, nk_total - kt0);
    u.nk += u.nk & 1;                               // even; the extra K-tile reads zeros
    u.mbeg = kt0 * 64;
    u.pb = pr.P + ((long long)u.mbeg * pr.ldp + tn * 256) * 2;
    u.qb = pr.Q + ((long long)u.mbeg * pr.ldq + tk * 256) * 2;
    u.wb = pr.W + (long long)tn * 256 * pr.ldw + tk * 256;
    u.ldp2 = (unsigned)(pr.ldp * 2);
    u.ldq2 = (unsigned)(pr.ldq * 2);
    u.ldw = pr.ldw;
    return u;
  };
  char* const dst0 = smem + wid * 1024;            // + buffer + half offset (+ 8192 for the second slab)

  // stage one half-tile of K-tile kt of unit U: 2 LDS-DMA per thread; rows >= M come from the zero page
#define T8_STAGE(ub, ld2, U, kt, hi, lds_off)                                                       \
  if (T8_EXP != 2) {                                                                                \
    const char* kb_ = U.ub + ((long long)(kt) * 64 * U.ld2 + (hi) * 256);                           \
    const int mr_ = U.mbeg + (kt) * 64;                                                             \
    t8_glds16(mr_ + mrow0 < g.M ? kb_ + ((unsigned)mrow0 * U.ld2 + sc0) : zsrc, dst0 + (lds_off));  \
    t8_glds16(mr_ + mrow1 < g.M ? kb_ + ((unsigned)mrow1 * U.ld2 + sc1) : zsrc, dst0 + (lds_off) + 8192); \
  }

  // ---- fragment addresses: lane (gq, q, p) reads rows 8 gq + q (+4 for the upper half, +32 for ks = 1), column
  //      group 4 p of 16-column tile t; f is the same for all of those rows, so tile t sits at block t ^ f
  const int gq = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int f = q | ((gq & 1) << 2);
  const int lds0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int rowb = lds0 + (8 * gq + q) * 256 + 8 * p;
  // Wave (wr, wc) owns rows (n) ha * 128 + wr * 64 + [0, 64) of both A halves and columns (kk) wc * 64 + [0, 64): ONE Q half (lo for wc < 2,
  // hi above), whose eight fragments are read once per K-tile and kept across both phases; a 64 x 64 wave tile takes 8 fragments per 16
  // MFMAs where the 64 x 32 tile of the four-phase form took 12 -- 24 instead of 48 fragments (48 instead of 96 ds_read_b64_tr_b16) per
  // wave and K-tile.
  int pa[4], qa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pa[i] = rowb + (((wr * 4 + i) ^ f) << 5);
#pragma unroll
  for (int j = 0; j < 4; ++j) qa[j] = rowb + ((((wc & 1) * 4 + j) ^ f) << 5) + (wc >> 1) * (T8_QHI - T8_QLO);

  f32x4 acc[8][4];
  bf16x8 fa[4][2], fb[4][2];
  typedef __attribute__((ext_vector_type(2))) int t8_i2;
#define T8_FRAG(dst, addr, off)                                                                          \
  {                                                                                                      \
    t8_i2 lo_, hi_;                                                                                      \
    if (T8_EXP == 3) { asm volatile("" : "=v"(lo_) : "v"(addr)); asm volatile("" : "=v"(hi_) : "v"(addr)); } else {  \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo_) : "v"(addr), "n"(off));               \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi_) : "v"(addr), "n"((off) + 1024)); }    \
    union { struct { t8_i2 a, b; } s; bf16x8 v; } u_;                                                    \
    u_.s.a = lo_; u_.s.b = hi_;                                                                          \
    dst = u_.v;                                                                                          \
  }
#define T8_READ_A(cur, half_off)                                                                         \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
    const int ad_ = (cur) ? pa[i] + T8_BUF : pa[i];                                                      \
    T8_FRAG(fa[i][0], ad_, (half_off)) T8_FRAG(fa[i][1], ad_, (half_off) + 8192)                         \
  }
#define T8_READ_B(cur)                                                                                   \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
    const int ad_ = (cur) ? qa[j] + T8_BUF : qa[j];                                                      \
    T8_FRAG(fb[j][0], ad_, T8_QLO) T8_FRAG(fb[j][1], ad_, T8_QLO + 8192)                                 \
  }
#define T8_TIE8(x) "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1])
#define T8_LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" : T8_TIE8(fa), T8_TIE8(fb));
#define T8_MFMA(ibase)                                                                                   \
  T8_LGKM0            /* before the barrier: the other wave group may re-stage what was just read one phase later */ \
  __builtin_amdgcn_s_barrier();                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
    if (T8_EXP != 1) acc[(ibase) + i][j] = A2V_MFMA_16x16x32(fa[i][ks], fb[j][ks], acc[(ibase) + i][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                         \
  __builtin_amdgcn_s_barrier();

  // one K-tile held in buffer `cur`; U1 / k1: unit and K index of the K-tile after it (P_hi goes to `oth`),
  // U2 / k2: of the one after that (P_lo, Q_lo, Q_hi re-stage `cur`)
  // (two phases of 32 MFMAs per wave: phase 1 = P_lo x this wave's Q half, phase 2 = P_hi x the same Q fragments.  P_lo, Q_lo and Q_hi
  //  are dead behind phase 1's "reads done" barrier and are re-staged for K-tile t + 2 during phase 2; P_hi of K-tile t + 1 goes to the
  //  other buffer during phase 1.  The counted wait leaves the six DMAs of phase 2 in flight.)
#define T8_KTILE(cur, oth, U1, k1, U2, k2)                                                               \
  T8_READ_B(cur)                                                                                         \
  T8_READ_A(cur, T8_PLO)                                                                                 \
  T8_STAGE(pb, ldp2, U1, k1, 1, (oth) + T8_PHI)                                      \
  T8_MFMA(0)                                                                                             \
  T8_READ_A(cur, T8_PHI)                                                                                 \
  T8_STAGE(pb, ldp2, U2, k2, 0, (cur) + T8_PLO)                                      \
  T8_STAGE(qb, ldq2, U2, k2, 0, (cur) + T8_QLO)                                      \
  T8_STAGE(qb, ldq2, U2, k2, 1, (cur) + T8_QHI)                                      \
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                       \
  T8_MFMA(4)

  int seq = blockIdx.x;
  T8Unit U = make_unit(seq);
  {
  T8_LANE_CONSTS(lane)
  T8_STAGE(pb, ldp2, U, 0, 0, T8_PLO)
  T8_STAGE(qb, ldq2, U, 0, 0, T8_QLO)
  T8_STAGE(qb, ldq2, U, 0, 1, T8_QHI)
  T8_STAGE(pb, ldp2, U, 0, 1, T8_PHI)
  T8_STAGE(pb, ldp2, U, 1, 0, T8_BUF + T8_PLO)
  T8_STAGE(qb, ldq2, U, 1, 0, T8_BUF + T8_QLO)
  T8_STAGE(qb, ldq2, U, 1, 1, T8_BUF + T8_QHI)
  }
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (stagger && wr == 1) __builtin_amdgcn_s_barrier();         // wave groups half a phase apart (see a2v_gemm_p8.hip)

  for (; seq < units; seq += gridDim.x) {
    int ls;                                          // the lane id from the hardware, not from a register kept across the K loop
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ls));
    T8_LANE_CONSTS(ls)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = U.nk;
    for (int kt = 0; kt < nk - 2; kt += 2) {
      T8_KTILE(0, T8_BUF, U, kt + 1, U, kt + 2)
      T8_KTILE(T8_BUF, 0, U, kt + 2, U, kt + 3)
    }
    // last two K-tiles: the stream continues with K-tiles 0 / 1 of the next unit (or re-fetches this unit's last
    // two K-tiles into dead buffers when there is none)
    const bool has_next = seq + (int)gridDim.x < units;
    const T8Unit Un = has_next ? make_unit(seq + gridDim.x) : U;
    const int kn0 = has_next ? 0 : nk - 2, kn1 = has_next ? 1 : nk - 1;
    T8_KTILE(0, T8_BUF, U, nk - 1, Un, kn0)
    T8_KTILE(T8_BUF, 0, Un, kn0, Un, kn1)

    // ---- epilogue: fp32 atomics; D rows n = 4 fq + r, columns kk = fr
    int le;                                          // (again from the hardware: lane-derived addresses stay out of the K loop's live set)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
    const int fre = le & 15, fqe = le >> 4;
    float* const wt = U.wb + (long long)(wr * 64 + fqe * 4) * U.ldw + wc * 64 + fre;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (T8_EXP != 4) atomicAdd(wt + (long long)(ha * 128 + i * 16 + r) * U.ldw + j * 16, acc[ha * 4 + i][j][r]);
            else asm volatile("" :: "v"(acc[ha * 4 + i][j][r]));
    U = Un;
  }
  if (stagger && wr == 0) __builtin_amdgcn_s_barrier();         // pairs with the wr = 1 group's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern int g_a2v_p8_stagger;
static bool t8_eligible(const WgradArgs& g) {
  const ConvA& a = g.a;
  const bool plain = a.taps == 1 && a.pad == 0 && a.rowmask == nullptr && a.rowidx == nullptr && a.stride == 1 &&
                     a.T_in == a.T_out && a.Cv == a.Cg && a.bdiv == 1 && g.G == 1;
  return plain && g.N % 256 == 0 && g.K % 256 == 0 && g.M >= 2048 && g.ldp * 2 * 64 < (1LL << 31) && a.pitch * 2 * 64 < (1LL << 31);
}
static T8Prob t8_prob(const WgradArgs& g) {
  T8Prob p;
  p.P = (const char*)g.P; p.Q = (const char*)g.a.ptr; p.W = g.W;
  p.ldp = g.ldp; p.ldq = g.a.pitch; p.ldw = g.ldw;
  p.N = g.N; p.K = g.K; p.tiles = (g.N / 256) * (g.K / 256);
  return p;
}

// one or two weight-gradient problems over the same M rows in one launch: the units of both share the CUs, so the M
// range is cut into fewer chunks and every output element receives fewer fp32 atomic adds
bool a2v_launch_tn_bf16_p8_pair(const WgradArgs* ga, const WgradArgs* gb, hipStream_t stream) {
  if (!t8_eligible(*ga) || (gb && (!t8_eligible(*gb) || gb->M != ga->M))) return false;
  T8Args args;
  args.pr[0] = t8_prob(*ga);
  args.pr[1] = gb ? t8_prob(*gb) : args.pr[0];
  args.nprob = gb ? 2 : 1;
  args.M = ga->M;
  const int tiles = args.pr[0].tiles + (gb ? args.pr[1].tiles : 0);
  static int cus_dev[A2V_MAX_DEVICES]; int& cus = cus_dev[a2v_device()];     // (per device)
  if (!cus) {
    hipDeviceProp_t p;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&p, dev);
    cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS);
  }
  const int nk_total = cdiv(args.M, 64);
  int chunks = cus / tiles;
  if (chunks < 1) chunks = 1;
  int kpc = cdiv(nk_total, chunks);
  kpc += kpc & 1;                                     // even number of K-tiles per chunk
  if (kpc < 12) return false;                         // too few K-tiles per unit: the atomics of a 256x256 tile would dominate
  chunks = cdiv(nk_total, kpc);
  const int units = tiles * chunks;
  dim3 grid(units < cus ? units : cus, 1, 1);
  hipLaunchKernelGGL(gemm_tn_bf16_p8_kernel, grid, dim3(512), T8_LDS, stream, args, chunks, kpc, g_a2v_p8_stagger == 2 ? 1 : 0);
  return true;
}

bool a2v_launch_tn_bf16_p8(const WgradArgs& g, hipStream_t stream) { return a2v_launch_tn_bf16_p8_pair(&g, nullptr, stream); }
