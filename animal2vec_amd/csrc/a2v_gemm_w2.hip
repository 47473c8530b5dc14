// 256 x 128 x 32 bf16 NT GEMM for the plain Linear layers with TWO workgroups per CU (round 4), gfx950.
//
// Why a second kernel next to the persistent 256 x 256 one (a2v_gemm_p8.hip): profiles/r04_p8_seam_probe.txt.  That kernel's K
// loop keeps the matrix pipe 85 % busy, but its two wave groups consume ONE K-tile stream and are in their epilogues together:
// for 5-14 us of every 27-37 us K = 1024 tile no wave of the CU issues an MFMA, most of it spent in store issue (a CU drains a
// 128 KB burst of output lines to HBM at ~27 GB/s).  Nothing can run beside it: one workgroup owns the whole CU.
// Here a workgroup is FOUR waves (one per SIMD) with a 256 x 128 tile and a 3-stage ring of 32-deep K-steps (72 KB of LDS), so two
// workgroups share a CU with independent phases: while one sits in its prologue or in store issue, the other's K loop has the
// matrix pipe to itself; while both are in their K loops they alternate on it.
//   * wave (wr, wc), wr = wid >> 1, wc = wid & 1, owns rows wr*64 + [0,64) and 128 + wr*64 + [0,64), columns wc*32 + [0,32) of
//     both 64-column halves: 8 x 4 MFMA tiles = 128 accumulator registers, the register picture of the 256 x 256 kernel's waves,
//     so its direct-store epilogue (swapped MFMA operands, permuted weight rows, raw-asm residual loads with counted waits)
//     is used unchanged with a half width of 64 instead of 128;
//   * per K-step a wave reads 8 A + 4 B fragments (12 KB) for 32 MFMAs -- the LDS traffic per MFMA of the 256 x 256 kernel -- and
//     issues 6 LDS-DMA pieces (16 rows x 64 B each); rows are 64 B in LDS, chunk position p of row r holds source chunk
//     p ^ S[(r >> 2) & 3], S = {0, 2, 3, 1}: every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) then covers all 16 slots of
//     the 256-B bank row;
//   * one workgroup barrier per K-step; the only wait is a counted s_waitcnt vmcnt(6) (the next stage stays in flight);
//   * not persistent: a finished workgroup's successor is placed while its neighbour computes.
// RESULT (profiles/r04_w2_experiment.txt): correct (the 256 x 256 kernel's tests pass on it bit for bit), but SLOWER -- 950 TFLOP/s
// against 1215-1320 on the step's shapes -- and OFF by default (a2v_debug_set(25, v)).  With the LDS-DMA taken out of the K loop
// (timing build, -DW2_EXP=1) the same kernel runs 116 736 x 3072 x 1024 with its full epilogue at 1496 TFLOP/s and K = 4096 at
// 1677: two independently phased workgroups do hide the epilogue and keep the matrix pipe ~90 % busy.  What bounds it is the CU's
// operand intake: a 256 x 128 tile needs 1.5 x the global -> LDS bytes per flop of a 256 x 256 tile (36 B/clk/CU at 1500 TFLOP/s),
// and LDS-DMA delivers ~22-25 B/clk/CU here -- the rate the 256 x 256 kernel's K loop already runs at.  The accumulator file (128
// registers x 8 waves = one 256 x 256 fp32 tile per CU) cannot be split between two workgroups without that 1.5 x.
// Eligibility: as the 256 x 256 kernel's direct-store form (plain contiguous A, operands < 4 GiB, C not overlapping R), G = 1,
// K % 64 == 0, K >= 256, N % 128 == 0, M >= 256.
#include "a2v_gemm.h"

typedef __attribute__((address_space(3))) void w2_lds_void_t;
typedef __attribute__((address_space(1))) const void w2_gbl_cvoid_t;
#ifndef W2_DMA_AUX
#define W2_DMA_AUX 0      /* cache-policy bits of the LDS-DMA loads (experiment builds): 1 sc0, 2 sc1, 4 nt */
#endif
__device__ __forceinline__ void w2_glds16(const char* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((w2_gbl_cvoid_t*)gsrc, (w2_lds_void_t*)lds_dst, 16, 0, W2_DMA_AUX);
}
#define W2_STAGE 24576                 // A 256 rows x 64 B, then B 128 rows x 64 B
#define W2_BOFF 16384
#define W2_LDS (3 * W2_STAGE)

// fixed epilogue flavours, as the 256 x 256 kernel's (a2v_gemm_p8.hip)
#define W2_FX_FC1 (F_BIAS | F_GELU | F_STORE_PRE | F_DACT)
#define W2_FX_FC2D (F_GELU_BWD | F_DACT | F_COLSUM)
#define W2_FX_BR (F_BIAS | F_RES)
#define W2_FX_BRD (F_BIAS | F_RES | F_DROPOUT)
#define W2_FX_R (F_RES)
#define W2_FX_B (F_BIAS)
#define W2_FX_BG (F_BIAS | F_GELU)
#define W2_FX_PLAIN 256

// loads + stores issued in row blocks [b0, b1) of the direct-store epilogue after a given residual load (a2v_gemm_p8.hip)
__host__ __device__ constexpr int p8_later(int /*k*/, int b0, int b1, int RQ, int NT) {
  int n = 0;
  for (int b = b0; b < b1; ++b) n += (b + RQ < NT ? 1 : 0) + 1;
  return n;
}

template <int FX>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_w2_kernel(GemmArgs g, int cs_part, char* wsp) {
  constexpr int IH = 4, BM = 256, HBW = 64;
  float* const g_p8_cs_part = (float*)(wsp + A2V_WS_CSPART);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int tiles_n = g.N / 128;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int grp = 0;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn, 0);
  const int m0 = tm * BM, n0 = tn * 128;
  const int mc = m0 + BM <= g.M ? m0 : g.M - BM;       // the last m-tile is shifted up (its duplicate rows write identical bits)
  const ConvA& a = g.a;

  // ---- LDS-DMA sources: piece p = 4 q + wid covers LDS rows 16 p + (lane >> 2), position lane & 3
  const int sw4 = (0x1E >> (2 * ((lane >> 4) & 3))) & 3;          // S[(row >> 2) & 3], S = {0, 2, 3, 1} packed as 0b01'11'10'00
  const int chunk = (lane & 3) ^ sw4;
  const int rl = 16 * wid + (lane >> 2);                           // row inside a group of 64 LDS rows
  const char* const asrc = (const char*)a.ptr + ((long long)(mc + rl) * a.pitch + chunk * 8) * 2;
  const int brow = (rl & ~31) + 8 * ((rl & 15) >> 2) + 4 * ((rl >> 4) & 1) + (rl & 3);     // weight row held by LDS row rl of a half
  const char* const bsrc = (const char*)g.B + ((long long)(n0 + brow) * g.ldb + chunk * 8) * 2;
  const long long a64 = a.pitch * 128, b64 = g.ldb * 128;          // bytes of 64 rows
  char* const dstw = smem + wid * 1024;
#define W2_ISSUE(soff_, kt_)                                                              \
  {                                                                                       \
    const long long ko_ = (long long)(kt_) * 64;                                          \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) w2_glds16(asrc + ko_ + q * a64, dstw + (soff_) + q * 4096); \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) w2_glds16(bsrc + ko_ + q * b64, dstw + (soff_) + W2_BOFF + q * 4096); \
  }

  // ---- fragment read addresses
  const int fr = lane & 15, fq = lane >> 4;
  const int fpos = (fq ^ ((0x1E >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
  const int lds0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int a_addr = lds0 + (wr * 64 + fr) * 64 + fpos;                  // + i * 1024 (i < 4), + 8192 + (i - 4) * 1024
  const int b_addr = lds0 + W2_BOFF + (wc * 32 + fr) * 64 + fpos;        // + hb * 4096 + j * 1024

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Two fragment sets: while the MFMAs of K-step kt run on set c, the fragments of K-step kt + 1 are read into the other set and the
  // LDS-DMA of K-step kt + 3 is issued -- one ds_read / DMA instruction per few MFMAs, in the issue slots the matrix pipe leaves
  // (a wave that issued its 6 DMA pieces and 12 reads in a block before its MFMAs spent more cycles issuing than multiplying).
  bf16x8 fa[2][8], fb[2][4];
#define W2_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define W2_TIE(c) "+v"(fa[c][0]), "+v"(fa[c][1]), "+v"(fa[c][2]), "+v"(fa[c][3]), "+v"(fa[c][4]), "+v"(fa[c][5]), "+v"(fa[c][6]), "+v"(fa[c][7]), \
                  "+v"(fb[c][0]), "+v"(fb[c][1]), "+v"(fb[c][2]), "+v"(fb[c][3])
#define W2_MFMA4(c, i)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                  \
    acc[i][j] = A2V_MFMA_16x16x32(fb[c][j], fa[c][i], acc[i][j], 0, 0, 0);
#define W2_SB __builtin_amdgcn_sched_barrier(0);
  // one K-step: cur_o = ring offset of stage kt (free: refilled with stage kt + 3), stage kt + 1 sits in the next slot.
  // DMA / RD: compile-time 0 / 1; WAITN: the counted wait for stage kt + 1 (6: stage kt + 2 stays in flight, 0: nothing younger)
#ifndef W2_EXP
#define W2_EXP 0        /* timing experiments (wrong results): 1 no DMA in the loop, 2 no wait / barrier, 4 no fragment reads */
#endif
#define W2_STEP(c, n, DMA_, RD_, WAITN)                                                                          \
  {                                                                                                              \
    constexpr int DMA = (W2_EXP & 1) ? 0 : DMA_;                                                                 \
    constexpr int RD = (W2_EXP & 4) ? 0 : RD_;                                                                   \
    constexpr int SYNC = (W2_EXP & 2) ? 0 : RD_;                                                                                                              \
    const int nxt_o = cur_o == 2 * W2_STAGE ? 0 : cur_o + W2_STAGE;                                              \
    if (SYNC) {                                                                                                  \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");                                               \
      __builtin_amdgcn_s_barrier();                                                                              \
    }                                                                                                            \
    const int aa = a_addr + nxt_o, ba = b_addr + nxt_o;                                                          \
    const long long ko_ = (long long)(kt + 3) * 64;                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    W2_MFMA4(c, 0) if (RD) { W2_DSR(fb[n][0], ba, 0); W2_DSR(fb[n][1], ba, 1024); } if (DMA) w2_glds16(asrc + ko_, dstw + cur_o); W2_SB \
    W2_MFMA4(c, 1) if (RD) { W2_DSR(fb[n][2], ba, 4096); W2_DSR(fb[n][3], ba, 5120); } if (DMA) w2_glds16(asrc + ko_ + a64, dstw + cur_o + 4096); W2_SB \
    W2_MFMA4(c, 2) if (RD) { W2_DSR(fa[n][0], aa, 0); W2_DSR(fa[n][1], aa, 1024); } if (DMA) w2_glds16(asrc + ko_ + 2 * a64, dstw + cur_o + 8192); W2_SB \
    W2_MFMA4(c, 3) if (RD) { W2_DSR(fa[n][2], aa, 2048); W2_DSR(fa[n][3], aa, 3072); } if (DMA) w2_glds16(asrc + ko_ + 3 * a64, dstw + cur_o + 12288); W2_SB \
    W2_MFMA4(c, 4) if (RD) { W2_DSR(fa[n][4], aa, 8192); } if (DMA) w2_glds16(bsrc + ko_, dstw + cur_o + W2_BOFF); W2_SB \
    W2_MFMA4(c, 5) if (RD) { W2_DSR(fa[n][5], aa, 9216); } if (DMA) w2_glds16(bsrc + ko_ + b64, dstw + cur_o + W2_BOFF + 4096); W2_SB \
    W2_MFMA4(c, 6) if (RD) { W2_DSR(fa[n][6], aa, 10240); } W2_SB                                                \
    W2_MFMA4(c, 7) if (RD) { W2_DSR(fa[n][7], aa, 11264); }                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    if (RD) asm volatile("s_waitcnt lgkmcnt(0)" : W2_TIE(n));                                                    \
    cur_o = nxt_o;                                                                                               \
    ++kt;                                                                                                        \
  }

  const int nk = g.K / 32;                      // even, >= 8 (launcher)
  W2_ISSUE(0, 0)
  W2_ISSUE(W2_STAGE, 1)
  W2_ISSUE(2 * W2_STAGE, 2)
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  W2_DSR(fb[0][0], b_addr, 0); W2_DSR(fb[0][1], b_addr, 1024); W2_DSR(fb[0][2], b_addr, 4096); W2_DSR(fb[0][3], b_addr, 5120);
  W2_DSR(fa[0][0], a_addr, 0); W2_DSR(fa[0][1], a_addr, 1024); W2_DSR(fa[0][2], a_addr, 2048); W2_DSR(fa[0][3], a_addr, 3072);
  W2_DSR(fa[0][4], a_addr, 8192); W2_DSR(fa[0][5], a_addr, 9216); W2_DSR(fa[0][6], a_addr, 10240); W2_DSR(fa[0][7], a_addr, 11264);
  asm volatile("s_waitcnt lgkmcnt(0)" : W2_TIE(0));
  int kt = 0, cur_o = 0;
  while (kt + 5 <= nk) {
    W2_STEP(0, 1, 1, 1, 6)
    W2_STEP(1, 0, 1, 1, 6)
  }
  // nk is even (launcher: K % 64 == 0): four K-steps are left -- the last DMA, then the ring runs dry
  W2_STEP(0, 1, 1, 1, 6)
  W2_STEP(1, 0, 0, 1, 6)
  W2_STEP(0, 1, 0, 1, 0)
  W2_STEP(1, 0, 0, 0, 0)
  // the fragment registers are free for the epilogue
#define W2_KILL(c) "=v"(fa[c][0]), "=v"(fa[c][1]), "=v"(fa[c][2]), "=v"(fa[c][3]), "=v"(fa[c][4]), "=v"(fa[c][5]), "=v"(fa[c][6]), "=v"(fa[c][7]), \
                   "=v"(fb[c][0]), "=v"(fb[c][1]), "=v"(fb[c][2]), "=v"(fb[c][3])
  asm volatile("" : W2_KILL(0));
  asm volatile("" : W2_KILL(1));

  // ---- epilogue: the direct-store epilogue of the 256 x 256 kernel (a2v_gemm_p8.hip, commented there), half width 64
  int le = lane;
  asm volatile("" : "+v"(le));
  bf16* const C = (bf16*)g.C;
  bf16* const C2 = (bf16*)g.C2;
  const bf16* const R = (const bf16*)g.R;
  const int flags = FX > 0 ? (FX & 0xFF) : g.flags;
  const bool need_r = (flags & (F_RES | F_GELU_BWD)) != 0;
  {
      // lane (fq, fr) holds, for row block i and column half hb: row i*16 + fr, columns hb*128 + wc*32 + 8 fq + [0, 8).
      // Addresses: a uniform 64-bit tile base (SGPRs) + a 32-bit per-lane element offset inside the tile's rows.
      const int fre = le & 15, fqe = le >> 4;
      const long long tile_off = (long long)mc * g.ldc + (long long)grp * g.c_goff + n0;
      bf16* const Ct = C + tile_off;
      bf16* const C2t = C2 + tile_off;
      // (made uniform explicitly: the raw-asm loads below take it as an SGPR pair, and hipcc does not always prove that itself)
      const unsigned long long rt_ = (unsigned long long)(R + ((long long)mc * g.ldr + (long long)grp * g.c_goff + n0));
      const bf16* const Rt = (const bf16*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rt_ >> 32)) << 32) |
                                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rt_));
      const int rskip = m0 - mc;                        // the shifted last m-tile: rows below m0 belong to its neighbour
      const int rlo = wr * 64 + fre, rhi = 128 + wr * IH * 16 + fre;
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
      p8_u32x4 rq[RQ];                       // (a native vector: asm operands cannot be members of the HIP uint4 struct)
#define P8_DROW(i_) ((i_) < 4 ? rlo + (i_) * 16 : rhi + ((i_) - 4) * 16)
#define P8_ST16(base_, eo_, val_) *(p8_u32x4*)((base_) + (eo_)) = (val_)
      // (uniform 64-bit base in SGPRs + a 32-bit byte offset per lane, as the compiler's own loads here; s_nop: the base may have
      //  just been written by v_readfirstlane / v_readlane, and a VMEM instruction reading such an SGPR needs 5 wait states)
#define P8_ROFF(k_) ((unsigned)(P8_DROW((k_) % NB) * (int)g.ldr + ((k_) / NB) * HBW + wc * 32 + fqe * 8) * 2u)
#define P8_RLOAD(k_) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(rq[(k_) % RQ]) : "v"(P8_ROFF(k_)), "s"(Rt) : "memory")
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
          if (flags & F_BIAS) Vec8<float>::load(g.bias + (long long)grp * g.N + n0 + hb * HBW + wc * 32 + fqe * 8, bias16[hb]);
        }
      }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int cl = hb * HBW + wc * 32 + fqe * 8;          // column inside the tile
        __builtin_amdgcn_sched_barrier(0);
        float bias8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) bias8[q] = bias16[hb][q];
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(bias8[i]));     // consumed here (see the staged form below)
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4 + IH; ++i) {
          const int row = P8_DROW(i);
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * hb][r]; v[4 + r] = acc[i][2 * hb + 1][r]; }
          const unsigned eo = (unsigned)(row * (int)g.ldc + cl);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] += bias8[q];
          // (not at the 256-row tile: its register allocation has no room for this branch -- 8 spills, reloaded inside the tile
          //  seam; the launcher gives the flavour 224- or 192-row tiles)
          if ((IH < 4 || FX > 0) && (flags & (F_DACT | F_GELU | F_STORE_PRE)) == (F_DACT | F_GELU | F_STORE_PRE)) {
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
          } else {
            if (flags & F_STORE_PRE) Vec8<bf16>::store(C2t + eo, v);
            if (flags & F_GELU) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = gelu_f(v[q]);
            }
          }
          if (flags & F_DROPOUT) {
            const unsigned long long e0 = (unsigned long long)(mc + row) * g.ldc + (unsigned long long)((long long)grp * g.c_goff + n0 + cl);
            drop_apply_n<8>(g.drop, e0, v);
          }
          if (need_r) {
            float h[8];
            const int k = hb * NB + i;
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rq[k % RQ]) : "n"(P8_RN(k)) : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              h[2 * q] = __uint_as_float(rq[k % RQ][q] << 16);
              h[2 * q + 1] = __uint_as_float(rq[k % RQ][q] & 0xffff0000u);
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
          Vec8<bf16>::store(Ct + eo, v);
          if (flags & F_COLSUM) {
            const float own = row >= rskip ? 1.f : 0.f;        // rows of the shifted tile's neighbour are summed there
#pragma unroll
            for (int q = 0; q < 8; ++q) cs[q] = fmaf(own, Cvt<bf16>::to(Cvt<bf16>::from(v[q])), cs[q]);
          }
          __builtin_amdgcn_sched_barrier(0);       // one row block at a time: interleaved, the blocks' temporaries spill
        }
        if (flags & F_COLSUM) {                // rows live in fr (the 16 lanes of a DPP row) and in i (in-lane)
#pragma unroll
          for (int q = 0; q < 8; ++q) cs[q] = row16_sum(cs[q]);
          if (fre == 0) {
            const int ncol = n0 + cl;
            if (cs_part) {                                   // own row of the scratch: plain stores, folded after the launch
              float* const dst = g_p8_cs_part + ((long long)(m0 / BM) * 2 + wr) * g.N + ncol;
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
#undef P8_ROFF
#undef P8_RLOAD
#undef P8_RN

  }
}

int g_a2v_gemm_w2 = 0;    // a2v_debug_set(25, v): 0 = off, 1 = eligible Linears with K <= 1024, 2 = every eligible Linear

template <int FX>
static void w2_go(const GemmArgs& g, int cs_part, char* ws, hipStream_t stream) {
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_w2_kernel<FX>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
    attr = true;
  }
  const int ntiles = cdiv(g.M, 256) * (g.N / 128);
  hipLaunchKernelGGL((gemm_nt_bf16_w2_kernel<FX>), dim3(ntiles), dim3(256), W2_LDS, stream, g, cs_part, ws);
}

__global__ __launch_bounds__(256) void w2_colsum_fold_kernel(float* __restrict__ colsum, int N, int parts, const char* __restrict__ ws) {
  const float* const part = (const float*)(ws + A2V_WS_CSPART);
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s0 = 0.f, s1 = 0.f;
  int p = blockIdx.y;
  for (; p + 16 < parts; p += 32) {
    s0 += part[(long long)p * N + n];
    s1 += part[(long long)(p + 16) * N + n];
  }
  if (p < parts) s0 += part[(long long)p * N + n];
  atomicAdd(colsum + n, s0 + s1);
}

bool a2v_launch_nt_bf16_w2(const GemmArgs& g, hipStream_t stream) {
  if (!g_a2v_gemm_w2 || (g_a2v_gemm_w2 == 1 && g.K > 1024)) return false;
  const ConvA& a = g.a;
  const bool plain = a.taps == 1 && a.pad == 0 && a.rowmask == nullptr && a.rowidx == nullptr && a.stride == 1 && a.T_in == a.T_out && a.Cv == a.Cg;
  if (!plain || g.G != 1 || g.K % 64 != 0 || g.K < 256 || g.N % 128 != 0 || g.M < 1024) return false;
  const bool contiguous = a.bdiv == 1 && (a.T_out >= g.M || a.bstride == (long long)a.T_out * a.pitch);
  const char *c0 = (const char*)g.C, *c1 = c0 + ((long long)(g.M - 1) * g.ldc + g.N) * 2;
  const char *r0 = (const char*)g.R, *r1 = r0 + ((long long)(g.M - 1) * g.ldr + g.N) * 2;
  const bool c_r_overlap = g.R != nullptr && r0 < c1 && c0 < r1;
  if (!contiguous || c_r_overlap || (long long)256 * g.ldc * 2 >= (1LL << 31) || (long long)256 * g.ldr * 2 >= (1LL << 31)) return false;
  if (g.flags & F_X3IMG) return false;
  const int parts = 2 * cdiv(g.M, 256);
  char* const ws = a2v_stream_ws(stream);
  if (!ws) return false;
  const int cs_part = ((g.flags & F_COLSUM) && (long long)parts * g.N <= (A2V_WS_CSPART_BYTES / 4)) ? 1 : 0;
  // (experiment kernel: the generic epilogue and the bias-only flavour are enough to compare it with the 256 x 256 kernel)
  // (the generic epilogue of a 256-row tile has no room for the gelu + gelu' store: that flag set needs its fixed flavour)
  const int eff = (g.flags & (F_STORE_PRE | F_GELU_BWD)) ? g.flags : (g.flags & ~F_DACT);
  if (eff == W2_FX_B) w2_go<W2_FX_B>(g, cs_part, ws, stream);
  else if (eff == W2_FX_FC1) w2_go<W2_FX_FC1>(g, cs_part, ws, stream);
  else if (eff == W2_FX_FC2D) w2_go<W2_FX_FC2D>(g, cs_part, ws, stream);
  else w2_go<0>(g, cs_part, ws, stream);
  if (cs_part) hipLaunchKernelGGL(w2_colsum_fold_kernel, dim3(cdiv(g.N, 256), 16), dim3(256), 0, stream, g.colsum, g.N, parts, ws);
  return true;
}
