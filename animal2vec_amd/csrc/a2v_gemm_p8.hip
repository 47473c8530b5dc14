// 256 x 256 x 64 bf16 NT GEMM for the plain Linear layers (out[M,N] = epi(x[M,K] @ w[N,K]^T)), gfx950.
//
// The 128x128 kernels of a2v_gemm_fast.hip read 16 KB of LDS fragments per 32 MFMAs per wave (wave tile 64x64):
// the LDS pipe and the MFMA pipe are 1:1, two barriers' worth of latency sits in every K step, and they stop at
// 600-1080 TFLOP/s.  This kernel follows the deep-pipelined structure of cdna_hip_programming.md section 5
// ("The 256^2 8-phase template"):
//   * 8 waves (2 x 4), one workgroup per CU, 128 KiB of LDS = 2 K-tile buffers x {A_lo, A_hi, B_lo, B_hi}
//     half-tiles of 128 rows x 64 k (16 KiB each, XOR-swizzled 16-B chunks, filled by LDS-DMA with the swizzle
//     applied to the per-lane SOURCE address);
//   * every K-tile is 4 phases; in phase p ALL waves work on one 128 x 128 quadrant of the workgroup tile
//     (lo/lo, lo/hi, hi/hi, hi/lo), each wave a 64 x 32 piece = 16 MFMAs, so a half-tile is dead one or two
//     phases after it was read and is re-staged while the rest of the K-tile is still being consumed;
//   * each phase issues ONE half-tile of LDS-DMA (2 x global_load_lds_dwordx4 per thread): A_hi(t+1) in phase 1,
//     then A_lo / B_lo / B_hi of K-tile t+2; the only wait is a counted s_waitcnt vmcnt(6) in phase 4 (three
//     half-tiles stay in flight across the barriers), barriers are raw s_barrier + lgkmcnt(0);
//   * PERSISTENT: one workgroup per CU walks its list of output tiles and the K-tile stream continues across the
//     tile seam (the last two K-tiles of a tile stage K-tiles 0 / 1 of the next one), so the epilogue -- and the
//     128 KB burst of output stores every CU would otherwise issue in lock step (measured: 16 us of a 39 us
//     K = 1024 tile) -- overlaps the next tile's loads and MFMAs.  The epilogue stages 16 rows x 64 columns per
//     wave at a time through the remaining 32 KiB of LDS.
// Wave (wr, wc) of quadrant (ha, hb) owns rows ha*128 + wr*64 + [0,64), columns hb*128 + wc*32 + [0,32).
// Eligibility (else the 128x128 kernel runs): plain row-major A (no taps / padding / masks), K % 128 == 0,
// K >= 256, N % 256 == 0, operands < 4 GiB.  The M tail is clamped on load and dropped by the epilogue.
#include "a2v_gemm.h"

typedef __attribute__((address_space(3))) void p8_lds_void_t;
typedef __attribute__((address_space(1))) const void p8_gbl_cvoid_t;
#ifndef P8_DMA_AUX
#define P8_DMA_AUX 0      /* cache-policy bits of the LDS-DMA loads (experiment builds): 1 sc0, 2 sc1, 4 nt */
#endif
__device__ __forceinline__ void p8_glds16(const char* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((p8_gbl_cvoid_t*)gsrc, (p8_lds_void_t*)lds_dst, 16, 0, P8_DMA_AUX);
}

// LDS map (16 KiB half-tiles): [A_lo0 A_hi0 A_lo1 A_hi1 | B_lo0 B_hi0 B_lo1 B_hi1]: the A region and the B region are
// 64 KiB each, so every fragment read is ONE address register (a_off / b_off, plus its ^64 twin) + a 16-bit immediate
// (buffer, half and row-tile offsets all fold into ds_read's offset field, which stops at 65535) -- with the two K-tile
// buffers 64 KiB apart the +buffer variants were four more address registers per operand.
#define P8_BUF 32768                  // K-tile buffer 1 relative to buffer 0, inside either region
#define P8_ALO 0
#define P8_AHI 16384
#define P8_BLO 65536
#define P8_BHI 81920
#define P8_BREG 65536                 // start of the B region (folded into b_off)
#define P8_STG 131072                 // epilogue staging: 8 waves x 16 rows x 64 fp32
#define P8_LDS (131072 + 8 * 4096)    // = 160 KiB

// Dynamic tile hand-out: [x] (x < 8) = tiles handed out to the workgroups of XCD x beyond their first one (ids keep the
// residue x mod 8, so xcd_remap still gives every XCD a contiguous, L2-friendly chunk of the tile space), [8] =
// workgroups finished (the last one resets all, so the next launch on the stream starts from zero).  A workgroup that starts late -- its CU held by
// another kernel, e.g. RCCL's all-reduce overlapping the backward pass -- then simply takes fewer tiles instead of
// stretching the launch by a whole extra round of its statically assigned ones.
// Column sums of the output (F_COLSUM, a bias gradient): every (m-tile, wave row) stores its partial sums to its own row
// of this scratch and a small kernel folds them.  As atomics on the N addresses of the final buffer they sat in vmcnt
// ahead of the next tile's first counted wait (stores and atomics share that counter on gfx9) and, contended 120-fold
// per address, held the K loop back: +23 us on a 144 us GEMM.
#define P8_CS_PART_MAX (A2V_WS_CSPART_BYTES / 4)
__global__ __launch_bounds__(256) void p8_colsum_fold_kernel(float* __restrict__ colsum, int N, int parts, const char* __restrict__ ws) {
  const float* const g_p8_cs_part = (const float*)(ws + A2V_WS_CSPART);
  const int n = blockIdx.x * 256 + threadIdx.x;        // blockIdx.y: one of 16 interleaved slices of the partial rows
  if (n >= N) return;
  float s0 = 0.f, s1 = 0.f;
  int p = blockIdx.y;
  for (; p + 16 < parts; p += 32) {
    s0 += g_p8_cs_part[(long long)p * N + n];
    s1 += g_p8_cs_part[(long long)(p + 16) * N + n];
  }
  if (p < parts) s0 += g_p8_cs_part[(long long)p * N + n];
  atomicAdd(colsum + n, s0 + s1);
}

// (tile counters: 9 x u32 at A2V_WS_CTR of the stream's workspace)
#ifdef P8_STAMP
// Diagnostic build only (MI355X_MICROARCH.md, DVFS give-back item 6; tools/inkernel_clock.py): wave 0 of every workgroup stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz) around its whole tile loop into this buffer, which nothing else reads:
// in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz.  The product library contains none of this.
__device__ unsigned long long g_p8_stamp[256][4];
extern "C" int a2v_diag_p8_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_p8_stamp), sizeof(unsigned long long) * 256 * 4) == hipSuccess ? 0 : -1;
}
#endif
#ifdef P8_STAMP2
// Diagnostic build only (tools/p8_probe.py): wave 0 of every workgroup keeps, per output tile, five stamps in the unused part of
// its epilogue staging area (LDS: no vector-memory operation is added to the kernel's in-order vmcnt stream) and copies them out
// when the tile list is done: [A] s_memrealtime and [1] s_memtime at the top of the epilogue, [2] s_memtime at its end (last store
// issued), [3] / [4] after the first / second K-tile of the NEXT tile.
#define P8_ST2_MAX 96
__device__ unsigned long long g_p8_stamp2[256][P8_ST2_MAX][5];
__device__ int g_p8_stamp2_n[256];
extern "C" int a2v_diag_p8_stamps2(unsigned long long* host_out, int* n_out) {
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_p8_stamp2), sizeof(unsigned long long) * 256 * P8_ST2_MAX * 5) != hipSuccess) return -1;
  return hipMemcpyFromSymbol(n_out, HIP_SYMBOL(g_p8_stamp2_n), sizeof(int) * 256) == hipSuccess ? 0 : -1;
}
#define P8_ST2(slot_) if (tid == 0 && st2_i < P8_ST2_MAX) { const unsigned long long v_ = __builtin_amdgcn_s_memtime(); \
    asm volatile("ds_write_b64 %0, %1" :: "v"(lds0 + P8_STG + 256 + (st2_i * 5 + (slot_)) * 8), "v"(v_) : "memory"); }
#define P8_ST2R(slot_) if (tid == 0 && st2_i < P8_ST2_MAX) { const unsigned long long v_ = __builtin_amdgcn_s_memrealtime(); \
    asm volatile("ds_write_b64 %0, %1" :: "v"(lds0 + P8_STG + 256 + (st2_i * 5 + (slot_)) * 8), "v"(v_) : "memory"); }
#else
#define P8_ST2(slot_)
#define P8_ST2R(slot_)
#endif

struct P8Tile {      // byte offsets (from the tile's A / B base) of the 16-B chunks this thread stages
  unsigned al0, al1, ah0, ah1, bl0, bl1, bh0, bh1;
  int m0, n0;        // first output row / column of the tile
  int mc;            // DS: first row actually staged = min(m0, M - BM): the last m-tile is shifted up instead of
};                   // clamping rows, its epilogue skips the rows below m0

// IH = 16-row MFMA tiles per wave in the upper A half: the workgroup tile is BM = 128 + 32 IH rows (256 / 224 / 192).
// M = 13440 (96 clips x 140 kept frames) is 60 x 224: with BM = 224 the student's Linears fill 240 of 256 CUs per
// round with no M tail, 12.5 % less work per round than 53 ragged 256-row tiles.
// loads + stores issued in row blocks [b0, b1) of the direct-store epilogue after a given residual load: each block issues the
// load RQ blocks ahead (while one exists) and one store
__host__ __device__ constexpr int p8_later(int /*k*/, int b0, int b1, int RQ, int NT) {
  int n = 0;
  for (int b = b0; b < b1; ++b) n += (b + RQ < NT ? 1 : 0) + 1;
  return n;
}
// DS ("direct store") = the epilogue writes the accumulators straight to global memory, no LDS staging: the MFMA
// operands are swapped (D = W_frag x X_frag, so a lane holds 4 consecutive output COLUMNS of one row instead of 4 rows
// of one column) and the rows of every 32-row group of the B half-tiles are permuted when they are staged (LDS row
// 16 j + f of a group holds weight row 8 (f >> 2) + 4 j + (f & 3); the per-lane DMA source address makes that free), so
// the two column blocks j = 0, 1 of a wave give each lane 8 consecutive columns = one 16-B bf16 store; the four lanes
// with the same MFMA column cover 64 contiguous bytes of an output row.  Replaces 128 ds_write_b32 + 32 ds_read_b128 +
// 16 wave barriers per wave and tile (LDS-store bound: 4 KiB per wave and pass at 64 B/clk/CU).
// FX > 0: the epilogue flags are fixed at compile time to the value FX (the launcher instantiates the 256- and 224-row tiles for the eight
// flavours a training step launches most: P8_FX_* below).  The generic epilogue allocates registers for the union of every
// path -- 16 bias values, the residual queue, dropout state, column sums, GELU temporaries -- which at 256 rows leaves room for
// only two residual row blocks in flight and none for the gelu + gelu' store; a fixed flavour keeps four in flight (-2.5 % per
// launch on the K = 1024 shapes).
#define P8_FX_FC1 (F_BIAS | F_GELU | F_STORE_PRE | F_DACT)      /* fc1 forward of a training step */
#define P8_FX_FC2D (F_GELU_BWD | F_DACT | F_COLSUM)             /* fc2 input gradient */
#define P8_FX_BR (F_BIAS | F_RES)                               /* proj / fc2, eval (teacher) */
#define P8_FX_BRD (F_BIAS | F_RES | F_DROPOUT)                  /* proj / fc2, train */
#define P8_FX_R (F_RES)                                         /* fc1 / qkv input gradients joined with a second gradient */
#define P8_FX_B (F_BIAS)                                        /* qkv, the teacher's upper fc2 */
#define P8_FX_BG (F_BIAS | F_GELU)                              /* fc1, eval (teacher) */
#define P8_FX_PLAIN 256                                         /* no flag at all (proj input gradient); bit 8 marks "fixed" */
#define P8_FX_FC1_8 (P8_FX_FC1 | F_DACT8)                       /* fc1 forward, the kept derivative in 8 bits (a2v_common.h dact8_*) */
#define P8_FX_FC2D_8 (P8_FX_FC2D | F_DACT8)                     /* fc2 input gradient reading it */
#define P8_FX_BQ (F_BIAS | F_QKNORM)                            /* the teacher's qkv: + the row-norm maxima of the attention's band limit */
#define P8_FXM 0x6FF                                            /* the epilogue flags a fixed flavour encodes (bit 8 is the marker above) */
// O32 (dtype A2V_BF16X3 of the C ABI): bf16 operands, fp32 C / C2 / R.  The caller hands over SPLIT operand images --
// A3 = [A_hi | A_lo | A_hi], B3 = [B_hi | B_hi | B_lo] along K (a2v_split_bf16), so that this ordinary bf16 product over 3 K
// is A_hi B_hi + A_lo B_hi + A_hi B_lo: every product exact, fp32 accumulation, 2^-17-relative operands -- and gets fp32-grade
// outputs at a third of the bf16 MFMA rate instead of the 1/16 of v_mfma_f32_16x16x4_f32.  With K three times as long the
// epilogue is a small share of a tile, so it is plain C++ here (compiler-placed waits), not the hand-counted form below.
template <int IH, bool DS, int FX = 0, bool O32 = false>
__global__ __launch_bounds__(512) void gemm_nt_bf16_p8_kernel(GemmArgs g, int stagger, int dyn_tiles, char* wsp, int cs_part) {
  static_assert(!O32 || (DS && FX == 0), "fp32 outputs: direct-store form, generic flags");
  unsigned* const g_p8_ctr = (unsigned*)(wsp + A2V_WS_CTR);
  float* const g_p8_cs_part = (float*)(wsp + A2V_WS_CSPART);
  constexpr int BM = 128 + 32 * IH;
  // the round-4 tile seam (P8_KTILE_FIRST / P8_KTILE_LAST below) for the fixed-flavour instantiations only: in the generic kernels its
  // two extra live offsets do not fit (12-49 spilled registers at the 256-row tile, one of them the tile claim's destination, stored
  // before the atomic had returned: repeating tile ids, a launch that never ends -- tests/test_abi.py's ISA check)
  constexpr bool SEAM2 = DS && FX > 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int tiles_n = g.N / 256;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int grp = blockIdx.y;
  const int tmode = (stagger >> 1) & 1;          // tile order (tile_coords): 1 = W-stationary n-groups
  const ConvA& a = g.a;
  const int lr = lane >> 3;
  const char* const abase = (const char*)a.ptr + (long long)grp * a.goff * 2;
  const char* const bbase = (const char*)g.B + (long long)grp * g.b_gstride * 2;

  // source chunk of LDS chunk position (lane & 7) in row r = slab*8 + lr:  c = pos ^ ((r >> 1) & 7)
#define P8_SRC_CHUNK(slab) (((lane & 7) ^ ((((slab) & 1) << 2) + (lane >> 4))) & 7)
  auto make_tile = [&](int seq) {
    P8Tile t;
    const int id = xcd_remap(seq < ntiles ? seq : ntiles - 1, ntiles);
    int tm, tn;
    tile_coords(id, tiles_m, tiles_n, tm, tn, tmode);
    t.m0 = tm * BM;
    t.n0 = tn * 256;
    t.mc = t.m0;
    unsigned ao[4], bo[4];
    if constexpr (DS) {
      // rows of A are contiguous (launcher): address = row * pitch, so the offsets of one tile are those of another
      // plus a uniform delta -- the tile seam adds that delta in place (next_tile below): no second set of offsets, no
      // division at the seam.
      t.mc = t.m0 + BM <= g.M ? t.m0 : g.M - BM;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int slab = wid + 8 * (q & 1);
        if ((q >> 1) && slab >= IH * 4) slab = wid & 3;
        ao[q] = (unsigned)(((long long)(t.mc + (q >> 1) * 128 + slab * 8 + lr) * a.pitch + P8_SRC_CHUNK(slab) * 8) * 2);
        const int bslab = wid + 8 * (q & 1);
        int brow = bslab * 8 + lr;
        brow = (brow & ~31) + 8 * ((brow & 15) >> 2) + 4 * ((brow >> 4) & 1) + (brow & 3);
        bo[q] = (unsigned)(((long long)(t.n0 + (q >> 1) * 128 + brow) * g.ldb + P8_SRC_CHUNK(bslab) * 8) * 2);
      }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // q = half * 2 + i: rows half*128 + (wid + 8 i) * 8 + lr
      int slab = wid + 8 * (q & 1);
      if ((q >> 1) && slab >= IH * 4) slab = wid & 3;     // short upper half: the spare DMA re-fetches a slab in place
      int m = t.m0 + (q >> 1) * 128 + slab * 8 + lr;
      m = m < g.M ? m : g.M - 1;
      const int b = m / a.T_out, tt = m - b * a.T_out;
      ao[q] = (unsigned)(((long long)(b / a.bdiv) * a.bstride + (long long)tt * a.pitch + P8_SRC_CHUNK(slab) * 8) * 2);
      const int bslab = wid + 8 * (q & 1);            // B halves are always 128 rows
      const int n = t.n0 + (q >> 1) * 128 + bslab * 8 + lr;
      bo[q] = (unsigned)(((long long)n * g.ldb + P8_SRC_CHUNK(bslab) * 8) * 2);
    }
    }
    t.al0 = ao[0]; t.al1 = ao[1]; t.ah0 = ao[2]; t.ah1 = ao[3];
    t.bl0 = bo[0]; t.bl1 = bo[1]; t.bh0 = bo[2]; t.bh1 = bo[3];
    return t;
  };
  char* const dst0 = smem + wid * 1024;            // + buffer + half offset (+ 8192 for the second slab)
  const int ah_second = (wid + 8 < IH * 4) ? 8192 : ((wid & 3) - wid) * 1024;   // A_hi: second DMA's slab relative to dst0

  const int nk = g.K / 64;
  const unsigned d64a = (unsigned)(a.pitch * 128), d64b = (unsigned)(g.ldb * 128);   // bytes of 64 rows of A / of B
  // stage one half-tile: 2 LDS-DMA per thread
#define P8_STAGE2(base, o0, o1, kt, lds_off, second)            \
  {                                                            \
    const unsigned ko_ = (unsigned)(kt) * 128u;                \
    p8_glds16(base + (o0 + ko_), dst0 + (lds_off));            \
    p8_glds16(base + (o1 + ko_), dst0 + (lds_off) + (second)); \
  }
#define P8_STAGE(base, o0, o1, kt, lds_off) P8_STAGE2(base, o0, o1, kt, lds_off, 8192)

  // ---- fragment read addresses (byte offsets inside a half-tile), ks = 0; ks = 1 flips chunk bit 2 (^ 64 B)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;                       // (row >> 1) & 7 for rows base*16 + fr
  const int lds0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS address of smem[0]
  const int a_off = lds0 + (wr * 64 + fr) * 128 + ((fq ^ sw) << 4);       // + i * 2048
  const int ah_off = lds0 + (wr * IH * 16 + fr) * 128 + ((fq ^ sw) << 4);  // upper half: IH row tiles per wave
  const int b_off = lds0 + P8_BREG + (wc * 32 + fr) * 128 + ((fq ^ sw) << 4);       // + j * 2048

  f32x4 acc[4 + IH][4];
  bf16x8 fa[4][2], fbl[2][2], fbh[2][2];

// Fragment reads are inline asm: hipcc orders a C++ LDS load after every LDS-DMA it cannot prove disjoint with an
// s_waitcnt vmcnt(0) -- at the top of every K-tile in this persistent form -- which would drain the three
// half-tiles deliberately left in flight.  The counted waits below (vmcnt(6) in phase 4, lgkmcnt(0) after each
// phase's first barrier, tied to the fragment registers so that no MFMA is scheduled above it) order them instead.
#define P8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define P8_READ_A(cur, half_off)                                                                       \
  {                                                                                                    \
    const int a0_ = a_off, a1_ = a_off ^ 64;                                                           \
    P8_DSR(fa[0][0], a0_, (cur) + (half_off)); P8_DSR(fa[0][1], a1_, (cur) + (half_off));              \
    P8_DSR(fa[1][0], a0_, (cur) + (half_off) + 2048); P8_DSR(fa[1][1], a1_, (cur) + (half_off) + 2048); \
    P8_DSR(fa[2][0], a0_, (cur) + (half_off) + 4096); P8_DSR(fa[2][1], a1_, (cur) + (half_off) + 4096); \
    P8_DSR(fa[3][0], a0_, (cur) + (half_off) + 6144); P8_DSR(fa[3][1], a1_, (cur) + (half_off) + 6144); \
  }
#define P8_READ_AH(cur)                                                                                 \
  {                                                                                                    \
    const int a0_ = ah_off, a1_ = ah_off ^ 64;                                                         \
    P8_DSR(fa[0][0], a0_, (cur) + P8_AHI); P8_DSR(fa[0][1], a1_, (cur) + P8_AHI);                      \
    P8_DSR(fa[1][0], a0_, (cur) + P8_AHI + 2048); P8_DSR(fa[1][1], a1_, (cur) + P8_AHI + 2048);        \
    if (IH > 2) { P8_DSR(fa[2][0], a0_, (cur) + P8_AHI + 4096); P8_DSR(fa[2][1], a1_, (cur) + P8_AHI + 4096); } \
    if (IH > 3) { P8_DSR(fa[3][0], a0_, (cur) + P8_AHI + 6144); P8_DSR(fa[3][1], a1_, (cur) + P8_AHI + 6144); } \
  }
#define P8_READ_B(fb, cur, half_off)                                                                   \
  {                                                                                                    \
    const int b0_ = b_off, b1_ = b_off ^ 64;          /* b_off includes P8_BREG */                     \
    P8_DSR(fb[0][0], b0_, (cur) + (half_off) - P8_BREG); P8_DSR(fb[0][1], b1_, (cur) + (half_off) - P8_BREG); \
    P8_DSR(fb[1][0], b0_, (cur) + (half_off) - P8_BREG + 2048); P8_DSR(fb[1][1], b1_, (cur) + (half_off) - P8_BREG + 2048); \
  }
#define P8_TIE4(x) "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1])
#define P8_LGKM0                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                  \
               : P8_TIE4(fa), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]), P8_TIE4(fbl), P8_TIE4(fbh));
#define P8_MFMA_BODY(ibase, jbase, fb, NI)                                                             \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
  _Pragma("unroll") for (int i = 0; i < (NI); ++i)                                                     \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
    acc[(ibase) + i][(jbase) + j] = DS                                                                 \
        ? A2V_MFMA_16x16x32(fb[j][ks], fa[i][ks], acc[(ibase) + i][(jbase) + j], 0, 0, 0) \
        : A2V_MFMA_16x16x32(fa[i][ks], fb[j][ks], acc[(ibase) + i][(jbase) + j], 0, 0, 0);
#define P8_MFMA(ibase, jbase, fb, NI)                                                                  \
  P8_LGKM0            /* before the barrier: the other wave group may re-stage what was just read one phase later */ \
  __builtin_amdgcn_s_barrier();                                                                        \
  __builtin_amdgcn_s_setprio(1);                                                                       \
  P8_MFMA_BODY(ibase, jbase, fb, NI)                                                                   \
  __builtin_amdgcn_s_setprio(0);                                                                       \
  __builtin_amdgcn_s_barrier();
// both B halves' fragments against one A half: 32 MFMAs between one pair of barriers (P8_TWO_PHASE)
#define P8_MFMA2(ibase, NI)                                                                            \
  P8_LGKM0                                                                                             \
  __builtin_amdgcn_s_barrier();                                                                        \
  __builtin_amdgcn_s_setprio(1);                                                                       \
  P8_MFMA_BODY(ibase, 0, fbl, NI)                                                                      \
  P8_MFMA_BODY(ibase, 2, fbh, NI)                                                                      \
  __builtin_amdgcn_s_setprio(0);                                                                       \
  __builtin_amdgcn_s_barrier();

  // one K-tile held in buffer `cur`; T1 / k1 = tile and K index of the K-tile after it (A_hi goes to `oth`),
  // T2 / k2 = of the one after that (A_lo, B_lo, B_hi re-stage `cur`)
// DS: the second chunk of a half-tile is 64 rows below the first = a uniform byte delta (d64a / d64b), added on the
// scalar side of the address, so only the first chunk's offset lives in a VGPR (opaque copies: hoisted out of the K loop
// the sums would be VGPRs again).  The short upper half (IH < 4) re-fetches a slab, its second offset stays a register.
#define P8_AL1(T_) (DS ? T_.al0 + d64a_ : T_.al1)
#define P8_AH1(T_) ((DS && IH == 4) ? T_.ah0 + d64a_ : T_.ah1)
#define P8_BL1(T_) (DS ? T_.bl0 + d64b_ : T_.bl1)
#define P8_BH1(T_) (DS ? T_.bh0 + d64b_ : T_.bh1)
// PH1 = the phase-1 stage of A_hi(T1, k1) (absent in a tile's first K-tile: staged before the seam, below), PH4X = an extra stage in
// phase 4 (the seam's last K-tile), WAITN = the counted wait of phase 4.
#ifndef P8_TWO_PHASE
#define P8_KTILE_X(cur, oth, T1, k1, T2, k2, PH1, PH4X, WAITN)                                          \
  unsigned d64a_ = d64a, d64b_ = d64b;                                                                 \
  asm volatile("" : "+s"(d64a_), "+s"(d64b_));                                                         \
  /* phase 1: quadrant lo/lo */                                                                        \
  P8_READ_B(fbl, cur, P8_BLO)                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  P8_READ_A(cur, P8_ALO)                                                                               \
  PH1                                                                                                  \
  P8_MFMA(0, 0, fbl, 4)                                                                                  \
  /* phase 2: lo/hi */                                                                                 \
  P8_READ_B(fbh, cur, P8_BHI)                                                                          \
  P8_STAGE(abase, T2.al0, P8_AL1(T2), k2, (cur) + P8_ALO)                                                  \
  P8_MFMA(0, 2, fbh, 4)                                                                                  \
  /* phase 3: hi/hi */                                                                                 \
  P8_READ_AH(cur)                                                                                      \
  P8_STAGE(bbase, T2.bl0, P8_BL1(T2), k2, (cur) + P8_BLO)                                                  \
  P8_MFMA(4, 2, fbh, IH)                                                                                  \
  /* phase 4: hi/lo (B_lo fragments kept from phase 1) */                                              \
  P8_STAGE(bbase, T2.bh0, P8_BH1(T2), k2, (cur) + P8_BHI)                                                  \
  PH4X                                                                                                 \
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");                                        \
  P8_MFMA(4, 0, fbl, IH)
#else
// Experiment builds (-DP8_TWO_PHASE): two phases of 32 MFMAs -- the lower A half, then the upper one, each against the fragments of BOTH B
// halves (read once, in phase 1) -- with the four-phase form's column ownership and epilogue (profiles/r06_nt_two_phase.txt)
#define P8_KTILE_X(cur, oth, T1, k1, T2, k2, PH1, PH4X, WAITN)                                          \
  unsigned d64a_ = d64a, d64b_ = d64b;                                                                 \
  asm volatile("" : "+s"(d64a_), "+s"(d64b_));                                                         \
  P8_READ_B(fbl, cur, P8_BLO)                                                                          \
  P8_READ_B(fbh, cur, P8_BHI)                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  P8_READ_A(cur, P8_ALO)                                                                               \
  PH1                                                                                                  \
  P8_MFMA2(0, 4)                                                                                       \
  P8_READ_AH(cur)                                                                                      \
  P8_STAGE(abase, T2.al0, P8_AL1(T2), k2, (cur) + P8_ALO)                                              \
  P8_STAGE(bbase, T2.bl0, P8_BL1(T2), k2, (cur) + P8_BLO)                                              \
  P8_STAGE(bbase, T2.bh0, P8_BH1(T2), k2, (cur) + P8_BHI)                                              \
  PH4X                                                                                                 \
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");                                        \
  P8_MFMA2(4, IH)
#endif
#define P8_KTILE(cur, oth, T1, k1, T2, k2) \
  P8_KTILE_X(cur, oth, T1, k1, T2, k2, P8_STAGE2(abase, T1.ah0, P8_AH1(T1), k1, (oth) + P8_AHI, ah_second), , 6)
// The tile seam and vmcnt.  Loads, stores, atomics and LDS-DMA retire through ONE in-order counter, so a counted wait for a DMA
// issued AFTER the epilogue's output stores is also a wait for those stores' acknowledgements -- and a store into lines that are
// not in L2 is acknowledged ~3.5 us after it was issued (tools/p8_probe.py, per-tile stamps: the first K-tile after an epilogue took
// 3.9 us against 1.35 us for every other one; with the same stores aimed at L2-resident lines the whole launch is 12-13 % shorter).
// The first K-tile of a tile used to stage A_hi of K-tile 1 in its phase 1, i.e. behind the stores, and wait for it in phase 4.  Now
// that half-tile is staged in phase 4 of the PREVIOUS tile's last K-tile (its buffer was last read in that K-tile's phase 3), all
// of K-tile 1 is older than the epilogue, and the first K-tile's wait may leave the epilogue's operations in flight: vmcnt(6 + E),
// E = the number of vector-memory operations every wave certainly issues in the epilogue (fewer than the real count only makes the
// wait stronger than needed).  The stores then have until the end of the tile's SECOND K-tile to be acknowledged.
#define P8_KTILE_FIRST(cur, oth, T2, k2) P8_KTILE_X(cur, oth, T2, 0, T2, k2, , , P8_WAIT0)
#define P8_KTILE_LAST(cur, oth, T1, k1, T2, k2)                                                          \
  P8_KTILE_X(cur, oth, T1, k1, T2, k2, P8_STAGE2(abase, T1.ah0, P8_AH1(T1), k1, (oth) + P8_AHI, ah_second), \
             P8_STAGE2(abase, T2.ah0, P8_AH1(T2), k2, (cur) + P8_AHI, ah_second), 8)

  // (a single round -- ntiles == gridDim.x -- keeps the static first tile: handing it out gains nothing there, and the
  // claim-one-ahead scheme would give half of the workgroups two tiles and the rest none)
  const int dmode = gridDim.y == 1 ? ((dyn_tiles == 2 && ntiles <= (int)gridDim.x) ? 1 : dyn_tiles) : 0;
  const bool dynamic = dmode != 0;
  unsigned* const slot = (unsigned*)(smem + P8_STG);      // id of the tile after the current one (word 0 of wave 0's staging)
  unsigned fetched = 0;
  const unsigned xcd = blockIdx.x & 7u;
  // mode 1: the first tile is blockIdx.x, later ones come from this residue class's counter (gridDim.x is a multiple
  // of 8 whenever there is more than one round).  mode 2: the first tile is handed out too -- a workgroup that is
  // placed late (its CU's LDS held by another stream's kernel, e.g. an RCCL ring during backward) then finds its
  // share already taken by the others instead of adding a whole tile time to the launch.
  const unsigned id_base = dmode == 2 ? 0u : gridDim.x;
  int seq = blockIdx.x;
  if (dmode == 2) {
    if (tid == 0) {
      const unsigned a = atomicAdd(&g_p8_ctr[xcd], 2u);
      slot[1] = 8u * a + xcd;
      slot[0] = 8u * (a + 1u) + xcd;
    }
    __syncthreads();
    seq = (int)__builtin_amdgcn_readfirstlane(((volatile unsigned*)slot)[1]);
    if (seq >= ntiles) {                                    // cannot happen while residue classes match; stay safe
      if (tid == 0 && atomicAdd(&g_p8_ctr[8], 1u) == gridDim.x - 1)
        for (int i = 0; i < 9; ++i) g_p8_ctr[i] = 0u;
      return;
    }
  } else if (dynamic && tid == 0) {
    *slot = gridDim.x + 8u * atomicAdd(&g_p8_ctr[xcd], 1u) + xcd;
  }
  P8Tile T = make_tile(seq);
  // ---- prologue: K-tiles 0 and 1 complete
  {
    const unsigned d64a_ = d64a, d64b_ = d64b;
    P8_STAGE(abase, T.al0, P8_AL1(T), 0, P8_ALO)
    P8_STAGE(bbase, T.bl0, P8_BL1(T), 0, P8_BLO)
    P8_STAGE(bbase, T.bh0, P8_BH1(T), 0, P8_BHI)
    P8_STAGE2(abase, T.ah0, P8_AH1(T), 0, P8_AHI, ah_second)
    P8_STAGE(abase, T.al0, P8_AL1(T), 1, P8_BUF + P8_ALO)
    P8_STAGE(bbase, T.bl0, P8_BL1(T), 1, P8_BUF + P8_BLO)
    P8_STAGE(bbase, T.bh0, P8_BH1(T), 1, P8_BUF + P8_BHI)
    if constexpr (SEAM2) P8_STAGE2(abase, T.ah0, P8_AH1(T), 1, P8_BUF + P8_AHI, ah_second)
  }
  if constexpr (SEAM2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (both K-tiles: the first K-tile's wait assumes K-tile 1 was complete before "the epilogue")
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                      // (K-tile 0 complete, three half-tiles of K-tile 1 in flight)
  __builtin_amdgcn_s_barrier();
  // The two waves that share a SIMD (w and w + 4, i.e. wr = 0 / 1) run half a phase apart from here on: one extra
  // barrier for the wr = 1 group pairs its "reads done" barrier with the other group's "MFMAs done" barrier, so one
  // wave of a SIMD issues LDS reads / DMA while the other owns the MFMA pipe (in lock step both queue for it at once).
  if ((stagger & 1) && wr == 1) __builtin_amdgcn_s_barrier();

#ifdef P8_STAMP
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (wid == 0) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
  int seq_next = 0;
#ifdef P8_STAMP2
  int st2_i = -1;           // the tile whose epilogue came last (its [3] / [4] stamps are taken in the next tile's K loop)
#endif
  for (; seq < ntiles; seq = seq_next) {
    // (at the end of the list the "next tile" is this tile again: harmless re-fetches of valid data into dead
    // buffers keep the vmcnt accounting uniform)
#pragma unroll
    for (int i = 0; i < 4 + IH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // epilogue operations every wave issues for certain (see P8_KTILE_FIRST): the C stores, with fixed flags also the second output's
    // stores, the bias vectors and the residual rows; the staged epilogue's stores are conditional (E = 0)
    constexpr int P8_E_ = !DS ? 0
                          : 2 * (4 + IH) * (((FX & P8_FXM) & (F_STORE_PRE)) && FX > 0 ? 2 : 1) + ((FX > 0 && ((FX & P8_FXM) & F_BIAS)) ? 4 : 0) +
                                ((FX > 0 && ((FX & P8_FXM) & (F_RES | F_GELU_BWD)) && !O32) ? 2 * (4 + IH) : 0);
    constexpr int P8_WAIT0 = 6 + (P8_E_ > 48 ? 48 : P8_E_);
    if constexpr (SEAM2) {
      { P8_KTILE_FIRST(0, P8_BUF, T, 2) }
#ifdef P8_STAMP2
      if (st2_i >= 0) { P8_ST2(3) }
#endif
      { P8_KTILE(P8_BUF, 0, T, 2, T, 3) }          // (nk is even and >= 4: the launcher asks for K % 128 == 0, K >= 256)
#ifdef P8_STAMP2
      if (st2_i >= 0) { P8_ST2(4) }
#endif
      for (int kt = 2; kt < nk - 2; kt += 2) {
        { P8_KTILE(0, P8_BUF, T, kt + 1, T, kt + 2) }
        { P8_KTILE(P8_BUF, 0, T, kt + 2, T, kt + 3) }
      }
    } else {
      for (int kt = 0; kt < nk - 2; kt += 2) {
        { P8_KTILE(0, P8_BUF, T, kt + 1, T, kt + 2) }
        { P8_KTILE(P8_BUF, 0, T, kt + 2, T, kt + 3) }
      }
    }
#ifdef P8_STAMP2
    ++st2_i;
#endif
    // last two K-tiles: the stream continues with K-tiles 0 / 1 of the next tile (its offsets are formed here, not
    // at the top of the tile, so that they are not live -- and spilled -- across the main loop)
    // (an LDS-typed read: through a generic pointer this was a flat_load, whose s_waitcnt vmcnt(0) lgkmcnt(0) drained
    // the three half-tiles in flight at every tile seam)
    if (dynamic) {
      unsigned nx;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"(lds0 + P8_STG) : "memory");
      seq_next = (int)__builtin_amdgcn_readfirstlane(nx);
    } else {
      seq_next = seq + (int)gridDim.x;
    }
    const bool has_next = seq_next < ntiles;
    P8Tile Tn = T;
    if constexpr (DS) {
      if (has_next) {
        // opaque copies: the divisions' reciprocals are formed here, not hoisted out of the tile loop into registers
        // that live (and spill) across the K loop
        int tm_ = tiles_m, tn_ = tiles_n;
        asm volatile("" : "+s"(tm_), "+s"(tn_));
        const int idn = xcd_remap(seq_next, tm_ * tn_);
        int tmn, tnn;
        tile_coords(idn, tm_, tn_, tmn, tnn, tmode);
        Tn.m0 = tmn * BM;
        Tn.n0 = tnn * 256;
        Tn.mc = Tn.m0 + BM <= g.M ? Tn.m0 : g.M - BM;
        const unsigned da = (unsigned)((Tn.mc - T.mc) * (int)a.pitch * 2), db = (unsigned)((Tn.n0 - T.n0) * (int)g.ldb * 2);
        Tn.al0 += da; Tn.ah0 += da; Tn.ah1 += da;        // (al1, bl1, bh1 -- and ah1 at IH = 4 -- are derived, dead here)
        Tn.bl0 += db; Tn.bh0 += db;
      }
    } else {
      if (has_next) Tn = make_tile(seq_next);
    }
    const int kn0 = has_next ? 0 : nk - 2, kn1 = has_next ? 1 : nk - 1;
    { P8_KTILE(0, P8_BUF, T, nk - 1, Tn, kn0) }
    if constexpr (SEAM2) { P8_KTILE_LAST(P8_BUF, 0, Tn, kn0, Tn, kn1) } else { P8_KTILE(P8_BUF, 0, Tn, kn0, Tn, kn1) }
    // the "+v" ties of P8_LGKM0 make every fragment register look live from one phase to the next, and so across the
    // epilogue into the next tile's first phase: redefine them (no instruction) so that the epilogue may use them
#define P8_KILL4(x) "=v"(x[0][0]), "=v"(x[0][1]), "=v"(x[1][0]), "=v"(x[1][1])
    asm volatile("" : P8_KILL4(fa), "=v"(fa[2][0]), "=v"(fa[2][1]), "=v"(fa[3][0]), "=v"(fa[3][1]), P8_KILL4(fbl), P8_KILL4(fbh));
#undef P8_KILL4

    // ---- epilogue (overlaps the LDS-DMA already in flight for the next tile): 8 passes of 16 rows per wave
    // lane-derived values are re-formed from an opaque copy of the lane id: hoisted out of the tile loop they
    // would be live -- and spilled -- across the K loop, and a spill reload is a VMEM load whose wait drains the DMA
    // Claim the tile after the next one HERE, at the top of the epilogue, as a raw-asm returning atomic (written in C++ the
    // compiler's atomic optimizer turns it into a wave-wide form whose result it consumes at once: s_waitcnt vmcnt(0), every DMA
    // in flight, right where it is issued).  Its result lands in `fetched` asynchronously; the counted wait in front of the slot
    // publication at the end of the epilogue (at least 9 vector-memory operations later: the tile's stores) retires it.  It used to
    // be issued two K-tiles earlier so that the K loop's own waits retired it -- but then the register had to survive the most
    // register-starved stretch of the kernel untouched, and hipcc once split its live range there (a v_mov read it before the
    // atomic had returned, the stale copy came back afterwards: repeating tile ids, a launch that never ends).  In the epilogue the
    // 64 fragment registers are free; tests/test_abi.py still checks in the ISA that nothing touches the register in between.
    if (dynamic && tid == 0) {
      const unsigned one = 1u, zoff = 0u;
      // s_nop 4: the pointer's SGPRs may have just been restored by v_readlane (SGPR spill); a VMEM instruction reading an
      // SGPR written by VALU needs 5 wait states, which the compiler only inserts around instructions it can see
      asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(fetched) : "v"(zoff), "v"(one), "s"(&g_p8_ctr[xcd]) : "memory");
    }
    // lane-derived values are re-formed in the epilogue from an opaque copy of the lane id.  The 8-bit-derivative fc1 flavour needs
    // one register more than the bf16 one at the 256-row tile (a second per-lane store offset): there `lane` itself was spilled
    // across the K loop and reloaded behind a full vmcnt(0) -- the tile claim's round trip -- so that flavour re-reads the hardware
    // lane id instead (volatile asm: as builtins the two instructions are loop invariant and hoisted in front of the tile loop)
    int le = lane;
    if constexpr (FX > 0 && (FX & F_DACT8) != 0 && (FX & F_STORE_PRE) != 0)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
    else
      asm volatile("" : "+v"(le));
    bf16* const C = (bf16*)g.C;
    bf16* const C2 = (bf16*)g.C2;
    const bf16* const R = (const bf16*)g.R;
    const int flags = FX > 0 ? (FX & P8_FXM) : g.flags;
    const bool need_r = (flags & (F_RES | F_GELU_BWD)) != 0;
    if constexpr (DS) {
      // The two wave groups share ONE workgroup barrier and run one barrier apart.  Without a barrier inside the epilogue, the
      // group that gets here first stops arriving at barriers, so the other group's last barrier waits for this group's WHOLE
      // epilogue, and then this group's first barrier of the next tile waits for the other's: the two epilogues ran one after
      // the other (s_memtime stamps: 2 x 3.1 us on a 27 us K = 1024 tile).  One extra barrier per tile -- taken BEFORE the
      // epilogue by the leading group (wr = 0) and AFTER it by the trailing one -- pairs the leader's with the trailer's last
      // K-loop barrier and the trailer's with the leader's first barrier of the next tile: both epilogues run at the same time,
      // two latency-bound waves per SIMD.  (Every wave executes exactly one of the two: the counts stay equal.)
      if (!(stagger & 1) || wr == 0) __builtin_amdgcn_s_barrier();
      P8_ST2R(0) P8_ST2(1)
      if constexpr (O32) {
        // fp32 outputs: lane (fq, fr) holds row i*16 + fr, columns hb*128 + wc*32 + 8 fq + [0, 8) = two 16-B stores
        const int fre = le & 15, fqe = le >> 4;
        const long long tile_off = (long long)T.mc * g.ldc + (long long)grp * g.c_goff + T.n0;
        float* const Ct = (float*)g.C + tile_off;
        // (F_X3IMG: ldc is the pitch of the bf16 image in C; the fp32 pre-activation / derivative tensor C2 is dense, pitch N)
        const long long ldc2 = (flags & F_X3IMG) ? (long long)g.N : g.ldc;
        float* const C2t = (float*)g.C2 + ((long long)T.mc * ldc2 + (long long)grp * g.c_goff + T.n0);
        const float* const Rt = (const float*)g.R + ((long long)T.mc * g.ldr + (long long)grp * g.c_goff + T.n0);
        const int rskip = T.m0 - T.mc;
        const int rlo = wr * 64 + fre, rhi = 128 + wr * IH * 16 + fre;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int cl = hb * 128 + wc * 32 + fqe * 8;
          float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (flags & F_BIAS) Vec8<float>::load(g.bias + (long long)grp * g.N + T.n0 + cl, bias8);
          float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4 + IH; ++i) {
            const int row = i < 4 ? rlo + i * 16 : rhi + (i - 4) * 16;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * hb][r] + bias8[r]; v[4 + r] = acc[i][2 * hb + 1][r] + bias8[4 + r]; }
            const long long eo = (long long)row * g.ldc + cl;
            if ((flags & (F_DACT | F_GELU | F_STORE_PRE)) == (F_DACT | F_GELU | F_STORE_PRE)) {
              float gr[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = gelu_with_grad(v[q], gr[q]);
              Vec8<float>::store(C2t + (long long)row * ldc2 + cl, gr);
            } else {
              if (flags & F_STORE_PRE) Vec8<float>::store(C2t + (long long)row * ldc2 + cl, v);
              if (flags & F_GELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = gelu_f(v[q]);
              }
            }
            if (flags & F_DROPOUT) {
              const unsigned long long e0 = (unsigned long long)(T.mc + row) * g.ldc + (unsigned long long)((long long)grp * g.c_goff + T.n0 + cl);
              drop_apply_n<8>(g.drop, e0, v);
            }
            if (need_r) {
              float h[8];
              Vec8<float>::load(Rt + (long long)row * g.ldr + cl, h);
              if (flags & F_RES) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += h[q];
              }
              if (flags & F_GELU_BWD) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= (flags & F_DACT) ? h[q] : gelu_grad_f(h[q]);
              }
            }
            if (flags & F_X3IMG) {         // the result leaves as its [hi | lo | hi] image (bf16, pitch ldc): nothing reads it as fp32
              bf16* const ci = (bf16*)g.C + ((long long)(T.mc + row) * g.ldc + T.n0 + cl);
              bf16x8 hh, ll;
#pragma unroll
              for (int q = 0; q < 8; ++q) { hh[q] = (bf16)v[q]; ll[q] = (bf16)(v[q] - (float)hh[q]); }
              *(bf16x8*)ci = hh;
              *(bf16x8*)(ci + g.N) = ll;
              *(bf16x8*)(ci + 2 * g.N) = hh;
            } else {
              Vec8<float>::store(Ct + eo, v);
            }
            if (flags & F_COLSUM) {
              const float own = row >= rskip ? 1.f : 0.f;
#pragma unroll
              for (int q = 0; q < 8; ++q) cs[q] = fmaf(own, v[q], cs[q]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (flags & F_COLSUM) {
#pragma unroll
            for (int q = 0; q < 8; ++q) cs[q] = row16_sum(cs[q]);
            if (fre == 0) {
              const int ncol = T.n0 + cl;
              if (cs_part) {
                float* const dst = g_p8_cs_part + ((long long)(T.m0 / BM) * 2 + wr) * g.N + ncol;
#pragma unroll
                for (int q = 0; q < 8; ++q) dst[q] = cs[q];
              } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) atomicAdd(g.colsum + (long long)grp * g.N + ncol + q, cs[q]);
              }
            }
          }
        }
      } else {
      // lane (fq, fr) holds, for row block i and column half hb: row i*16 + fr, columns hb*128 + wc*32 + 8 fq + [0, 8).
      // Addresses: a uniform 64-bit tile base (SGPRs) + a 32-bit per-lane element offset inside the tile's rows.
      const int fre = le & 15, fqe = le >> 4;
      const long long tile_off = (long long)T.mc * g.ldc + (long long)grp * g.c_goff + T.n0;
      // (tile bases made uniform explicitly, as Rt below: the raw-asm stores take them as SGPR pairs)
#define P8_UNI_(ptr_) ((__typeof__(ptr_))(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(ptr_) >> 32)) << 32) | \
                                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(ptr_))))
      bf16* const Ct = P8_UNI_(C + tile_off);
      bf16* const C2t = P8_UNI_(C2 + tile_off);
      unsigned char* const C2t8 = P8_UNI_((unsigned char*)g.C2 + tile_off);      // (F_DACT8: one byte per element)
#undef P8_UNI_
      // (made uniform explicitly: the raw-asm loads below take it as an SGPR pair, and hipcc does not always prove that itself)
      const unsigned long long rt_ = (FX > 0 && (FX & F_DACT8) != 0)
          ? (unsigned long long)((const unsigned char*)g.R + ((long long)T.mc * g.ldr + (long long)grp * g.c_goff + T.n0))
          : (unsigned long long)(R + ((long long)T.mc * g.ldr + (long long)grp * g.c_goff + T.n0));
      const bf16* const Rt = (const bf16*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rt_ >> 32)) << 32) |
                                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rt_));
      const int rskip = T.m0 - T.mc;                        // the shifted last m-tile: rows below m0 belong to its neighbour
      const int rlo = wr * 64 + fre, rhi = 128 + wr * IH * 16 + fre;
      // F_QKNORM: the tile's rows belong to clip qk_c0 below in-tile row qk_rb and to clip qk_c0 + 1 from there on (a clip is
      // a.T_out >= BM rows: at most one boundary per tile; uniform values)
      constexpr bool QKN = FX > 0 && (FX & F_QKNORM) != 0;
      int qk_c0 = 0, qk_rb = 0x7fffffff;
      if constexpr (QKN) {
        qk_c0 = T.mc / a.T_out;
        qk_rb = (qk_c0 + 1) * a.T_out - T.mc;
      }
      const bool qk_split = qk_rb < BM;
      // Residual / pre-activation rows are fetched RQ row blocks ahead of their use, across both column halves (2 NB row blocks
      // form one sequence).  Loads, stores, atomics and the LDS-DMA share ONE in-order counter (vmcnt): the data of a load issued
      // behind a store is only there once that store is acknowledged.  hipcc, which must assume the two kinds complete out of
      // order, waits with vmcnt(0) wherever both are pending -- that sat out the acknowledgement of the store issued just before
      // (and, at the top of the epilogue, the next tile's LDS-DMA) in front of every residual row block: +6 us on a 20 us K = 1024
      // tile.  The residual loads are therefore raw asm (invisible to the compiler's bookkeeping) with hand-counted waits:
      // vmcnt(P8_RN(k)) = the number of vector-memory operations certainly issued after load k (later residual loads + one store
      // per finished row block; anything more -- the pre-activation store, column sums -- only makes the wait stronger than needed).
      // Both halves' bias vectors are read before the first store for the same reason.  Stores are unconditional: the rows a
      // shifted last m-tile shares with its neighbour are computed from the same operands in the same order, so both write the
      // same bits (the launcher refuses C overlapping R).
      constexpr int RQ = (IH == 4 && FX == 0) ? 2 : 4;    // residual row blocks in flight (4 VGPRs each; the generic 256-row tile has fewer to spare)
      constexpr int NB = 4 + IH, NT = 2 * NB;
      typedef unsigned p8_u32x4 __attribute__((ext_vector_type(4)));
      typedef unsigned p8_u32x2 __attribute__((ext_vector_type(2)));
      constexpr bool D8 = FX > 0 && (FX & F_DACT8) != 0;      // the derivative tensor (C2 / R) holds one byte per element
      p8_u32x4 rq[D8 ? 1 : RQ];              // (a native vector: asm operands cannot be members of the HIP uint4 struct)
      p8_u32x2 rq8[D8 ? RQ : 1];
#define P8_DROW(i_) ((i_) < 4 ? rlo + (i_) * 16 : rhi + ((i_) - 4) * 16)
      // P8_ST_AUX (experiment builds): cache-policy bits on the output stores -- 0 plain stores (product), 1 nt, 2 sc1, 3 sc0 sc1,
      // 4 nt sc1, 5 nt sc0, 6 nt sc0 sc1; non-zero = raw asm in the SGPR-base + 32-bit-offset form (the builtin nontemporal store
      // only sets nt; the trailing s_nop 1: a VALU write of the data registers of a 16-byte store needs two wait states on gfx940+,
      // and hipcc's hazard pass does not look inside asm).  Measured (profiles/r06_store_policy.txt): "nt sc1" (streaming, written
      // through L2) takes 3-7 % off the K = 1024 launches run back to back on their own -- and ADDS 1-9 % to the same launches inside
      // the training step (whole step -0.7 %), where the acknowledgement of a written-through store waits behind the memory traffic
      // of the neighbouring kernels.  Not the product.
#ifndef P8_ST_AUX
#define P8_ST_AUX 0
#endif
#if P8_ST_AUX == 0
#define P8_STV(base_, boff_, val_) (*(__typeof__(val_)*)((char*)(base_) + (boff_)) = (val_))
#else
#if P8_ST_AUX == 1
#define P8_ST_AUXS " nt"
#elif P8_ST_AUX == 2
#define P8_ST_AUXS " sc1"
#elif P8_ST_AUX == 3
#define P8_ST_AUXS " sc0 sc1"
#elif P8_ST_AUX == 4
#define P8_ST_AUXS " nt sc1"
#elif P8_ST_AUX == 5
#define P8_ST_AUXS " nt sc0"
#else
#define P8_ST_AUXS " nt sc0 sc1"
#endif
#define P8_STV(base_, boff_, val_)                                                                                              \
  {                                                                                                                             \
    if constexpr (sizeof(val_) == 16) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" P8_ST_AUXS "\n\ts_nop 1" :: "v"(boff_), "v"(val_), "s"(base_) : "memory"); \
    else asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" P8_ST_AUXS :: "v"(boff_), "v"(val_), "s"(base_) : "memory");  \
  }
#endif
#define P8_ST16(base_, eo_, val_) P8_STV(base_, (eo_) * 2u, val_)
#define P8_ST8BF(base_, eo_, v_)                                                              \
  {                                                                                           \
    bf16x8 pk_;                                                                               \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) pk_[q_] = (bf16)(v_)[q_];               \
    P8_STV(base_, (eo_) * 2u, pk_);                                                           \
  }
      // (uniform 64-bit base in SGPRs + a 32-bit byte offset per lane, as the compiler's own loads here; s_nop: the base may have
      //  just been written by v_readfirstlane / v_readlane, and a VMEM instruction reading such an SGPR needs 5 wait states)
#define P8_ROFF(k_) ((unsigned)(P8_DROW((k_) % NB) * (int)g.ldr + ((k_) / NB) * 128 + wc * 32 + fqe * 8) * 2u)
#define P8_RLOAD(k_)                                                                                                             \
  {                                                                                                                              \
    if constexpr (D8) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(rq8[(k_) % RQ]) : "v"(P8_ROFF(k_) >> 1), "s"(Rt) : "memory"); \
    else asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(rq[(k_) % RQ]) : "v"(P8_ROFF(k_)), "s"(Rt) : "memory");   \
  }
      // operations issued after residual load k and before its use in row block k (see above)
#define P8_RN(k_) ((k_) < RQ ? (RQ - 1 - (k_)) + p8_later((k_) , 0, (k_), RQ, NT) : 1 + p8_later((k_), (k_) - RQ + 1, (k_), RQ, NT))
      if (need_r) {
#pragma unroll
        for (int k = 0; k < RQ; ++k) {
          P8_RLOAD(k);
        }
      }
      float bias16[2][8];
      {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
          for (int q = 0; q < 8; ++q) bias16[hb][q] = 0.f;
          if (flags & F_BIAS) Vec8<float>::load(g.bias + (long long)grp * g.N + T.n0 + hb * 128 + wc * 32 + fqe * 8, bias16[hb]);
        }
      }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int cl = hb * 128 + wc * 32 + fqe * 8;          // column inside the tile
        __builtin_amdgcn_sched_barrier(0);
        float bias8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) bias8[q] = bias16[hb][q];
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(bias8[i]));     // consumed here (see the staged form below)
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float qm0 = 0.f, qm1 = 0.f;              // F_QKNORM: running maxima of this lane's rows, first / second clip of the tile
#pragma unroll
        for (int i = 0; i < 4 + IH; ++i) {
          const int row = P8_DROW(i);
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * hb][r]; v[4 + r] = acc[i][2 * hb + 1][r]; }
          const unsigned eo = (unsigned)(row * (int)g.ldc + cl);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] += bias8[q];
          if constexpr (QKN) {
            // squared norm of the row's 32-column segment: 8 columns here, the other 24 in the lanes fq ^ 1, 2, 3 of the same fr
            // (v_permlane32_swap / v_permlane16_swap: no LDS round trip); fp32 values before the bf16 rounding -- the consumer's
            // 1 % margin covers (1 + 2^-9)^2
            float sq = v[0] * v[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) sq = fmaf(v[q], v[q], sq);
            {
              const unsigned u_ = __float_as_uint(sq);
              auto r_ = __builtin_amdgcn_permlane32_swap(u_, u_, false, false);
              sq = __uint_as_float(r_[0]) + __uint_as_float(r_[1]);
              const unsigned x_ = __float_as_uint(sq);
              auto t_ = __builtin_amdgcn_permlane16_swap(x_, x_, false, false);
              sq = __uint_as_float(t_[0]) + __uint_as_float(t_[1]);
            }
            if (qk_split) {
              const bool second = row >= qk_rb;
              qm0 = fmaxf(qm0, second ? 0.f : sq);
              qm1 = fmaxf(qm1, second ? sq : 0.f);
            } else {
              qm0 = fmaxf(qm0, sq);
            }
          }
          // (not at the 256-row tile: its register allocation has no room for this branch -- 8 spills, reloaded inside the tile
          //  seam; the launcher gives the flavour 224- or 192-row tiles)
          if ((IH < 4 || FX > 0) && (flags & (F_DACT | F_GELU | F_STORE_PRE)) == (F_DACT | F_GELU | F_STORE_PRE)) {
            if constexpr (FX > 0 && (FX & F_DACT8) != 0) {
              // the derivative in 8 bits (dact8_q): two dwords per 8 columns, one 8-B store -- the same store count (P8_E_), half the bytes
              p8_u32x2 g8 = {0u, 0u};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                a2v_f32x2 dd;
                const a2v_f32x2 hh = gelu_with_grad2((a2v_f32x2){v[2 * q], v[2 * q + 1]}, dd);
                v[2 * q] = hh[0];
                v[2 * q + 1] = hh[1];
                g8[q >> 1] = dact8_put(dact8_put(g8[q >> 1], 2u * (q & 1), dd[0]), 2u * (q & 1) + 1u, dd[1]);
                asm volatile("" : "+v"(v[2 * q]), "+v"(v[2 * q + 1]), "+v"(g8[q >> 1]));
              }
              P8_STV(C2t8, eo, g8);
            } else {
            p8_u32x4 gr;                             // gelu and gelu' from one evaluation; the derivative is what is kept,
#pragma unroll                                       // packed pair by pair, the order pinned (interleaved, the eight
            for (int q = 0; q < 4; ++q) {            // evaluations' temporaries spill at the 256-row tile)
              a2v_f32x2 dd;
              const a2v_f32x2 hh = gelu_with_grad2((a2v_f32x2){v[2 * q], v[2 * q + 1]}, dd);
              v[2 * q] = hh[0];
              v[2 * q + 1] = hh[1];
              const bf16 b0 = (bf16)dd[0], b1 = (bf16)dd[1];
              gr[q] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
              asm volatile("" : "+v"(v[2 * q]), "+v"(v[2 * q + 1]), "+v"(gr[q]));
            }
            P8_ST16(C2t, eo, gr);
            }
          } else {
            if (flags & F_STORE_PRE) P8_ST8BF(C2t, eo, v)
            if (flags & F_GELU) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = gelu_f(v[q]);
            }
          }
          if (flags & F_DROPOUT) {
            const unsigned long long e0 = (unsigned long long)(T.mc + row) * g.ldc + (unsigned long long)((long long)grp * g.c_goff + T.n0 + cl);
            drop_apply_n<8>(g.drop, e0, v);
          }
          if (need_r) {
            float h[8];
            const int k = hb * NB + i;
            if constexpr (D8) {
              asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rq8[k % RQ]) : "n"(P8_RN(k)) : "memory");
              dact8_unpack4(rq8[k % RQ][0], h);
              dact8_unpack4(rq8[k % RQ][1], h + 4);
            } else {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rq[k % RQ]) : "n"(P8_RN(k)) : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              h[2 * q] = a2v_lo16_f32(rq[k % RQ][q]);
              h[2 * q + 1] = a2v_hi16_f32(rq[k % RQ][q]);
            }
            }
            if (k + RQ < NT) P8_RLOAD(k + RQ);              // RQ row blocks ahead
            if (flags & F_RES) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] += h[q];
            }
            if (flags & F_GELU_BWD) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] *= (flags & F_DACT) ? h[q] : gelu_grad_f(h[q]);
            }
          }
          P8_ST8BF(Ct, eo, v)
          if (flags & F_COLSUM) {
            const float own = row >= rskip ? 1.f : 0.f;        // rows of the shifted tile's neighbour are summed there
#pragma unroll
            for (int q = 0; q < 8; ++q) cs[q] = fmaf(own, Cvt<bf16>::to(Cvt<bf16>::from(v[q])), cs[q]);
          }
          __builtin_amdgcn_sched_barrier(0);       // one row block at a time: interleaved, the blocks' temporaries spill
        }
        if constexpr (QKN) {                   // maximum over the 16 rows in fr (DPP), then one atomic maximum per clip and segment
#define P8_DPP_MAX(x_, ctrl_) x_ = fmaxf(x_, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x_), ctrl_, 0xF, 0xF, true)))
          P8_DPP_MAX(qm0, 0xB1); P8_DPP_MAX(qm0, 0x4E); P8_DPP_MAX(qm0, 0x141); P8_DPP_MAX(qm0, 0x140);
          if (qk_split) { P8_DPP_MAX(qm1, 0xB1); P8_DPP_MAX(qm1, 0x4E); P8_DPP_MAX(qm1, 0x141); P8_DPP_MAX(qm1, 0x140); }
#undef P8_DPP_MAX
          if (le == 0) {                       // (non-negative floats order as their bit patterns)
            unsigned* const nt_ = (unsigned*)g.colsum + (long long)qk_c0 * (g.N / 32) + (T.n0 + cl) / 32;
            atomicMax(nt_, __float_as_uint(qm0));
            if (qk_split) atomicMax(nt_ + g.N / 32, __float_as_uint(qm1));
          }
        }
        if (flags & F_COLSUM) {                // rows live in fr (the 16 lanes of a DPP row) and in i (in-lane)
#pragma unroll
          for (int q = 0; q < 8; ++q) cs[q] = row16_sum(cs[q]);
          if (fre == 0) {
            const int ncol = T.n0 + cl;
            if (cs_part) {                                   // own row of the scratch: plain stores, folded after the launch
              float* const dst = g_p8_cs_part + ((long long)(T.m0 / BM) * 2 + wr) * g.N + ncol;
#pragma unroll
              for (int q = 0; q < 8; ++q) dst[q] = cs[q];
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) atomicAdd(g.colsum + (long long)grp * g.N + ncol + q, cs[q]);
            }
          }
        }
      }
#undef P8_DROW
#undef P8_ST16
#undef P8_ST8BF
#undef P8_STV
#undef P8_ROFF
#undef P8_RLOAD
#undef P8_RN
      }      // !O32
      P8_ST2(2)
      if ((stagger & 1) && wr == 1) __builtin_amdgcn_s_barrier();     // the trailing group's half of the epilogue barrier (see above)
    } else {
    const int fre = le & 15, fqe = le >> 4;            // staging writes: MFMA layout (rows 4 fq + r, column fr)
    const int er = le >> 3, ec = (le & 7) * 8;           // read-out: 8 rows x 8 lanes, 8 columns (16 B of bf16) per lane
    float* const stg = (float*)(smem + P8_STG) + wid * 1024;
    const int ncol = T.n0 + wc * 32 + ec + (ec >= 32 ? 96 : 0);
    const long long ccol = (long long)grp * g.c_goff + ncol;
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g.flags & F_BIAS) Vec8<float>::load(g.bias + (long long)grp * g.N + ncol, bias8);
    // consume the load here: a load still pending on some path at the end of the epilogue makes hipcc put an
    // s_waitcnt vmcnt(0) at the head of the K loop (draining the LDS-DMA pipeline every K-tile)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(bias8[i]));
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // residual / pre-activation rows are fetched one pass (16 rows) ahead of their use
    Raw8<bf16> rq[2][2];
#define P8_RLOAD(ps_, slot)                                                                           \
  if (need_r) {                                                                                       \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                \
      int m_ = T.m0 + ((ps_) < 4 ? wr * 64 + (ps_) * 16 : 128 + wr * IH * 16 + ((ps_) - 4) * 16) + it * 8 + er; \
      m_ = m_ < g.M ? m_ : g.M - 1;                                                                   \
      rq[slot][it].load(R + (long long)m_ * g.ldr + ccol);                                            \
    }                                                                                                 \
  }
    P8_RLOAD(0, 0)
#pragma unroll
    for (int ps = 0; ps < 4 + IH; ++ps) {   // fully unrolled: a rolled loop turns acc[ps] into a scalar branch tree per value
      if (ps < 3 + IH) P8_RLOAD(ps + 1, (ps + 1) & 1)
      // stage[row][col ^ 16*((row >> 2) & 1)]: 2-way (= minimum) on the 4-B writes, conflict-free 16-B reads
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[ps][j][r];
          stg[(fqe * 4 + r) * 64 + ((j * 16 + fre) ^ ((fqe & 1) << 4))] = v;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int mrow0 = T.m0 + (ps < 4 ? wr * 64 + ps * 16 : 128 + wr * IH * 16 + (ps - 4) * 16);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r = it * 8 + er;
        const int m = mrow0 + r;
        const float* sp_ = stg + r * 64 + (ec ^ (((r >> 2) & 1) << 4));
        const f32x4 va = *(const f32x4*)sp_, vb = *(const f32x4*)(sp_ + 4);
        float v[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        if (m < g.M) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += bias8[i];
          if ((g.flags & (F_DACT | F_GELU | F_STORE_PRE)) == (F_DACT | F_GELU | F_STORE_PRE)) {
            float gr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = gelu_with_grad(v[i], gr[i]);
            Vec8<bf16>::store(C2 + (long long)m * g.ldc + ccol, gr);
          } else {
            if (g.flags & F_STORE_PRE) Vec8<bf16>::store(C2 + (long long)m * g.ldc + ccol, v);
            if (g.flags & F_GELU) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = gelu_f(v[i]);
            }
          }
          if (g.flags & F_DROPOUT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = drop_apply(g.drop, (unsigned long long)m * g.ldc + ccol + i, v[i]);   // (pair form: 10 spills at IH = 4 here)
          }
          if (need_r) {
            float h[8];
            rq[ps & 1][it].unpack(h);
            if (g.flags & F_RES) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] += h[i];
            }
            if (g.flags & F_GELU_BWD) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] *= (g.flags & F_DACT) ? h[i] : gelu_grad_f(h[i]);
            }
          }
          Vec8<bf16>::store(C + (long long)m * g.ldc + ccol, v);
          if (g.flags & F_COLSUM) {
#pragma unroll
            for (int i = 0; i < 8; ++i) cs[i] += Cvt<bf16>::to(Cvt<bf16>::from(v[i]));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (g.flags & F_COLSUM) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float s_ = cs[i];
        s_ += __shfl_xor(s_, 8, 64);
        s_ += __shfl_xor(s_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64);
        cs[i] = s_;
      }
      if (le < 8) {
        if (cs_part) {                                   // own row of the scratch: plain stores, folded after the launch
          float* const dst = g_p8_cs_part + ((long long)(T.m0 / BM) * 2 + wr) * g.N + ncol;
#pragma unroll
          for (int i = 0; i < 8; ++i) dst[i] = cs[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) atomicAdd(g.colsum + (long long)grp * g.N + ncol + i, cs[i]);
        }
      }
    }
#undef P8_RLOAD
    }
    if (dynamic && tid == 0) {
      // (the direct-store epilogue issued 2 (4 + IH) stores after the claim, the staged one 16: all but the oldest two may stay
      //  in flight -- waiting for more of them is waiting for store acknowledgements on wave 0, and so on every barrier after it)
      constexpr int CLAIM_N = DS ? 2 * (4 + IH) - 2 : 8;
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(fetched) : "n"(CLAIM_N) : "memory");
      *slot = id_base + 8u * fetched + xcd;                              // read by every wave at the next tile's tail, many barriers later
    }
    T = Tn;
  }
#ifdef P8_STAMP
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 256) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    g_p8_stamp[blockIdx.x][0] = st_c0; g_p8_stamp[blockIdx.x][1] = c1; g_p8_stamp[blockIdx.x][2] = st_r0; g_p8_stamp[blockIdx.x][3] = r1;
  }
#endif
#ifdef P8_STAMP2
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 256) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int n = st2_i + 1 < P8_ST2_MAX ? st2_i + 1 : P8_ST2_MAX;
    g_p8_stamp2_n[blockIdx.x] = n;
    const unsigned long long* src = (const unsigned long long*)(smem + P8_STG + 256);
    for (int i = 0; i < n * 5; ++i) g_p8_stamp2[blockIdx.x][i / 5][i % 5] = src[i];
  }
#endif
  if (dynamic && tid == 0) {
    if (atomicAdd(&g_p8_ctr[8], 1u) == gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) g_p8_ctr[i] = 0u;
    }
  }
  if ((stagger & 1) && wr == 0) __builtin_amdgcn_s_barrier();         // pairs with the wr = 1 group's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's re-fetches must land before the LDS is released
}

int g_a2v_p8_dynamic = 2;   // a2v_debug_set(7, v): dynamic tile hand-out in the persistent NT kernel (2: first tile too)
int g_a2v_p8_stagger = 2;   // a2v_debug_set(6, v): wave groups half a phase apart (NT and TN 8-phase kernels): 0 off, 1 only K >= 2048 (NT), 2 always
int g_a2v_p8_direct = 1;  // a2v_debug_set(10, v): direct-store epilogue (swapped MFMA operands, permuted B rows) on / off
int g_a2v_p8_fx = 1;      // a2v_debug_set(20, v): 0 = every flavour takes the generic epilogue (A/B)
int g_a2v_p8_ih = 0;      // a2v_debug_set(5, v): force the tile height (2 / 3 / 4), 0 = choose per shape
int g_a2v_p8_tmode = 0;   // a2v_debug_set(23, v): tile order of the persistent NT kernel: 0 = 8 m-tiles x all n-tiles, 1 = W-stationary groups of four n-tiles (N >= 2048)
int g_a2v_p8_tail = 1;    // a2v_debug_set(22, v): a last round of tiles that fills at most v quarters of the CUs goes to the 128 x 128 kernel (0: stays here)
static bool p8_launch_impl(const GemmArgs& g, hipStream_t stream, bool out32, bool allow_tail = true);
bool a2v_launch_nt_bf16_p8(const GemmArgs& g, hipStream_t stream) { return p8_launch_impl(g, stream, false); }
// bf16 (split) operands, fp32 C / C2 / R (dtype A2V_BF16X3): only the direct-store form exists
bool a2v_launch_nt_x3_p8(const GemmArgs& g, hipStream_t stream) { return p8_launch_impl(g, stream, true); }
static bool p8_launch_impl(const GemmArgs& g, hipStream_t stream, bool out32, bool allow_tail) {
  const ConvA& a = g.a;
  const bool plain = a.taps == 1 && a.pad == 0 && a.rowmask == nullptr && a.stride == 1 && a.T_in == a.T_out &&
                     a.Cv == a.Cg;
  if (!plain || g.K % 128 != 0 || g.K < 256 || g.N % 256 != 0 || g.M < 1024) return false;
  // the direct-store form addresses A rows as row * pitch (contiguous batch items); anything else takes the staged form
  const bool contiguous = a.bdiv == 1 && (a.T_out >= g.M || a.bstride == (long long)a.T_out * a.pitch);
  const int osz = out32 ? 4 : 2;
  const char *c0 = (const char*)g.C, *c1 = c0 + ((long long)(g.M - 1) * g.ldc + g.N) * osz;
  const char *r0 = (const char*)g.R, *r1 = r0 + ((long long)(g.M - 1) * g.ldr + g.N) * osz;
  const bool c_r_overlap = g.R != nullptr && r0 < c1 && c0 < r1;
  const bool direct = (g_a2v_p8_direct || out32) && contiguous && !c_r_overlap && g.G == 1 && (long long)256 * g.ldc * 2 < (1LL << 31) && (long long)256 * g.ldr * 2 < (1LL << 31);
  if (out32 && !direct) return false;
  if ((g.flags & F_X3IMG) && (!out32 || (g.flags & (F_RES | F_DROPOUT)) || g.ldc < 3LL * g.N || g.ldc % 8 || g.G != 1)) return false;
  // 32-bit byte offsets inside one group's operands
  const long long a_span = ((long long)((g.M - 1) / a.T_out / a.bdiv) * a.bstride + (long long)a.T_out * a.pitch + g.K) * 2;
  const long long b_span = ((long long)g.N * g.ldb + g.K) * 2;
  if (a_span >= (1LL << 32) || b_span >= (1LL << 32)) return false;
  static int cus_dev[A2V_MAX_DEVICES]; int& cus = cus_dev[a2v_device()];     // (per device)
  if (!cus) {
    hipDeviceProp_t p;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&p, dev);
    cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
#define P8_ATTR(IH_, DS_) (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p8_kernel<IH_, DS_>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    P8_ATTR(4, true) P8_ATTR(3, true) P8_ATTR(2, true) P8_ATTR(4, false) P8_ATTR(3, false) P8_ATTR(2, false)
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p8_kernel<4, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p8_kernel<3, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p8_kernel<2, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
#define P8_ATTRF(F_) (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p8_kernel<4, true, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS); \
  (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p8_kernel<3, true, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    P8_ATTRF(P8_FX_FC1_8) P8_ATTRF(P8_FX_FC2D_8) P8_ATTRF(P8_FX_FC1) P8_ATTRF(P8_FX_FC2D) P8_ATTRF(P8_FX_BR) P8_ATTRF(P8_FX_BRD) P8_ATTRF(P8_FX_R) P8_ATTRF(P8_FX_B) P8_ATTRF(P8_FX_BG) P8_ATTRF(P8_FX_PLAIN) P8_ATTRF(P8_FX_BQ)
#undef P8_ATTRF
#undef P8_ATTR
  }
  // tile height: rounds x rows per round is what one CU executes; ties go to the taller tile
  int best_ih = 4;
  long long best = -1;
  const bool dact_store = (g.flags & (F_DACT | F_GELU | F_STORE_PRE)) == (F_DACT | F_GELU | F_STORE_PRE);
  int fx = 0;                                // a fixed-flavour instantiation of the 256-row tile exists for these flag sets
  // (DACT only qualifies a pre-activation store or a GELU backward: alone it changes nothing)
  const int eff = (g.flags & (F_STORE_PRE | F_GELU_BWD)) ? g.flags : (g.flags & ~F_DACT);
  if (direct && g_a2v_p8_fx && !out32)
    for (int f : {(int)P8_FX_FC1_8, (int)P8_FX_FC2D_8, (int)P8_FX_FC1, (int)P8_FX_FC2D, (int)P8_FX_BR, (int)P8_FX_BRD, (int)P8_FX_R, (int)P8_FX_B, (int)P8_FX_BG, (int)P8_FX_BQ})
      if (eff == f) fx = f;
  if (direct && g_a2v_p8_fx && eff == 0 && !out32) fx = P8_FX_PLAIN;
  const bool fx1 = fx == P8_FX_FC1 || fx == P8_FX_FC1_8;
  for (int ih = (dact_store && direct && !fx1 && !out32) ? 3 : 4; ih >= 2; --ih) {
    if (g_a2v_p8_ih && ih != g_a2v_p8_ih && !(dact_store && direct && !fx1 && g_a2v_p8_ih == 4)) continue;
    if ((g.flags & F_QKNORM) && ih < 3 && best >= 0) continue;       // (its fixed flavour exists at 256 and 224 rows)
    const int bm = 128 + 32 * ih;
    const long long nt = (long long)cdiv(g.M, bm) * (g.N / 256);
    // what one CU executes = rounds x the time of one tile; a shorter tile is NOT proportionally cheaper (same LDS traffic for B,
    // same barriers, same seam: measured 0.97 of a 256-row tile for 224 rows at K = 1024 -- 53760 x 4096 x 1024 is 14 rounds of
    // 256 rows in 396 us against 15 rounds of 224 in 412)
    const long long cost = (long long)cdiv(nt, cus) * (ih == 4 ? 256 : (ih == 3 ? 248 : 240));
    if (best < 0 || cost < best) { best = cost; best_ih = ih; }
  }
  const int bm = 128 + 32 * best_ih;
  // the 8-bit derivative tensor exists in the two fixed flavours only (256- / 224-row tiles): anything else goes to the 128 x 128 kernels
  const bool d8 = (g.flags & F_DACT8) != 0;
  if (d8 && !((fx == P8_FX_FC1_8 || fx == P8_FX_FC2D_8) && best_ih >= 3)) return false;
  // the row-norm maxima exist in one fixed flavour only, over whole clips of at least a tile's rows; nothing else can honour the flag
  const bool qkn = (g.flags & F_QKNORM) != 0;
  if (qkn && !(fx == P8_FX_BQ && best_ih >= 3 && g.colsum && a.T_out >= 256 && g.M % a.T_out == 0 && g.N % 32 == 0)) return false;
  if (qkn) allow_tail = false;               // (the 128 x 128 kernel that takes a split-off last round does not form them)
  // A last round with few tiles leaves most CUs idle for a whole tile time (M = 768 x 151 kept frames, N = 1024: 7.08 rounds of
  // 256-row tiles -- 12 % of the launch for 1 % of the work).  The rows of that round go to the 128 x 128 LDS-DMA kernel instead: a
  // quarter of the work per tile, spread over four times as many CUs, issued right behind this launch.
  if (g_a2v_p8_tail && allow_tail && direct && !out32 && g.G == 1) {
    const int tiles_n = g.N / 256;
    const long long nt_all = (long long)cdiv(g.M, bm) * tiles_n;
    const long long full = nt_all / cus, rem = nt_all % cus;
    if (full >= 2 && rem > 0 && rem * 4 <= (long long)cus * g_a2v_p8_tail) {
      const long long rows_main = (full * cus / tiles_n) * bm;
      if (rows_main >= 1024 && rows_main < g.M && g.M - rows_main >= 64 && a.Cg % 64 == 0 && g.K % 64 == 0) {
        GemmArgs gm = g;
        gm.M = (int)rows_main;
        gm.a.T_in = gm.a.T_out = gm.M;
        GemmArgs gt = g;
        gt.M = g.M - (int)rows_main;
        gt.a.T_in = gt.a.T_out = gt.M;
        gt.a.bstride = 0;
        gt.a.ptr = (const char*)a.ptr + rows_main * a.pitch * 2;
        gt.C = (char*)g.C + rows_main * g.ldc * 2;
        if (g.C2) gt.C2 = (char*)g.C2 + rows_main * g.ldc * (d8 ? 1 : 2);
        if (g.R) gt.R = (const char*)g.R + rows_main * g.ldr * ((d8 && (g.flags & F_GELU_BWD)) ? 1 : 2);
        gt.drop_base = g.drop_base + rows_main * g.ldc;
        const bool ok = p8_launch_impl(gm, stream, false, false);     // (the main part is a whole number of rounds by construction)
        if (ok && a2v_launch_nt_bf16_fast(gt, stream)) return true;
        if (ok) return false;                            // cannot happen (eligibility checked above); never leave half a product
      }
    }
  }
  const int ntiles = cdiv(g.M, bm) * (g.N / 256);
  // staggering the wave groups: one group's epilogue (conversions, GELU, stores) runs under the other group's MFMAs.  With the
  // LDS-staged epilogue of round 1 it cost a few per cent at K = 1024; with the direct-store epilogue it pays on every shape
  // (53760 x 4096 x 1024: 430 -> 395 us, 13440 x 4096 x 1024: 117 -> 104 us, K = 4096 unchanged; weight gradients 407 -> 371 us)
  const int tiles_n_ = g.N / 256;
  const int stagger = ((g_a2v_p8_stagger == 2 || (g_a2v_p8_stagger == 1 && g.K >= 2048)) ? 1 : 0) |
                      ((g_a2v_p8_tmode && tiles_n_ > 4 && tiles_n_ % 4 == 0) ? 2 : 0);      // bit 1: tile order
  dim3 grid(ntiles < cus ? ntiles : cus, g.G, 1);
  const int parts = 2 * cdiv(g.M, bm);
  char* const ws = a2v_stream_ws(stream);
  if (!ws) return false;                     // no workspace left for this stream: the 128 x 128 kernel needs none
  const int cs_part = ((g.flags & F_COLSUM) && g.G == 1 && (long long)parts * g.N <= P8_CS_PART_MAX) ? 1 : 0;
#define P8_GO(IH_, DS_) hipLaunchKernelGGL((gemm_nt_bf16_p8_kernel<IH_, DS_>), grid, dim3(512), P8_LDS, stream, g, stagger, g_a2v_p8_dynamic, ws, cs_part)
  if (out32) {
#define P8_GO32(IH_) hipLaunchKernelGGL((gemm_nt_bf16_p8_kernel<IH_, true, 0, true>), grid, dim3(512), P8_LDS, stream, g, stagger, g_a2v_p8_dynamic, ws, cs_part)
    if (best_ih == 4) P8_GO32(4); else if (best_ih == 3) P8_GO32(3); else P8_GO32(2);
#undef P8_GO32
  } else if (direct) {
#define P8_GOF(F_)                                                                                                        \
  {                                                                                                                       \
    if (best_ih == 4)                                                                                                     \
      hipLaunchKernelGGL((gemm_nt_bf16_p8_kernel<4, true, F_>), grid, dim3(512), P8_LDS, stream, g, stagger, g_a2v_p8_dynamic, ws, cs_part); \
    else                                                                                                                  \
      hipLaunchKernelGGL((gemm_nt_bf16_p8_kernel<3, true, F_>), grid, dim3(512), P8_LDS, stream, g, stagger, g_a2v_p8_dynamic, ws, cs_part); \
  }
    if (best_ih >= 3 && fx == P8_FX_FC1_8) P8_GOF(P8_FX_FC1_8)
    else if (best_ih >= 3 && fx == P8_FX_FC2D_8) P8_GOF(P8_FX_FC2D_8)
    else if (best_ih >= 3 && fx == P8_FX_FC1) P8_GOF(P8_FX_FC1)
    else if (best_ih >= 3 && fx == P8_FX_FC2D) P8_GOF(P8_FX_FC2D)
    else if (best_ih >= 3 && fx == P8_FX_BR) P8_GOF(P8_FX_BR)
    else if (best_ih >= 3 && fx == P8_FX_BRD) P8_GOF(P8_FX_BRD)
    else if (best_ih >= 3 && fx == P8_FX_R) P8_GOF(P8_FX_R)
    else if (best_ih >= 3 && fx == P8_FX_B) P8_GOF(P8_FX_B)
    else if (best_ih >= 3 && fx == P8_FX_BG) P8_GOF(P8_FX_BG)
    else if (best_ih >= 3 && fx == P8_FX_BQ) P8_GOF(P8_FX_BQ)
    else if (best_ih >= 3 && fx == P8_FX_PLAIN) P8_GOF(P8_FX_PLAIN)
#undef P8_GOF
    else if (best_ih == 4) P8_GO(4, true); else if (best_ih == 3) P8_GO(3, true); else P8_GO(2, true);
  } else {
    if (best_ih == 4) P8_GO(4, false); else if (best_ih == 3) P8_GO(3, false); else P8_GO(2, false);
  }
#undef P8_GO
  if (cs_part) hipLaunchKernelGGL(p8_colsum_fold_kernel, dim3(cdiv(g.N, 256), 16), dim3(256), 0, stream, g.colsum, g.N, parts, ws);
  return true;
}
