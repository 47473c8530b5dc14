// Decoder grouped convolution (Decoder1d, reference nn/modalities/modules.py:137-192: Conv1d(768, 768, k 7, groups 16)):
// forward and input gradient for 48 channels per group, bf16, gfx950.
//
//   out[m][g*48 + n] = epi( sum_tap sum_ci  X[m + tap - pad][g*48 + ci] * W[g][n][tap*64 + ci] )       (stride 1)
//
// 17 FLOP per byte moved: this operator is HBM-bound, not MFMA-bound.  The slab kernel (a2v_conv_slab.hip), built for
// the MFMA-bound k19 x 64-channel positional convolution, ran it at 0.27 of its HBM time: one workgroup per (group,
// 1024 rows) reads 96-byte pieces of 1536-byte rows (every row is fetched by 16 workgroups that are far apart in
// dispatch order), pads 48 channels to 64 on both GEMM axes (56 % useful MFMA work), and cannot overlap its slab load,
// its MFMAs and its LDS-staged stores (one 150 KB workgroup per CU).  Here:
//   * a workgroup owns rows x 8 GROUPS (768 contiguous bytes of every row it touches), 4 waves x 2 groups;
//   * the weights of a wave's two groups live in REGISTERS for the workgroup's whole life (2 x 33 KB per wave as MFMA
//     fragments: one wave per SIMD, 512 registers), K runs over the flattened (tap, channel) axis -- 336 = 10.5 K-steps
//     of 32 instead of 7 x 64 -- and N is three 16-column blocks: no padding on either axis;
//   * rows are walked in blocks of 64 (+ 6 halo rows); the next block's slab is fetched into registers while the
//     current one is multiplied, and written to the other LDS buffer afterwards (one barrier per block);
//   * swapped MFMA operands (D = W_frag x X_frag) and a permuted weight-row order give each lane 8 + 4 consecutive output
//     channels of one row: 16-B + 8-B stores straight from the accumulators, no LDS staging.
// Epilogues: bias and / or residual add (what Decoder1d's forward and its input gradient need); anything else, other
// shapes, masks and row gathers stay with the slab / implicit-GEMM kernels.
// Template <CI, CO, GPW>: input / output channels per group and groups per wave -- <48, 48, 2> the three 768 -> 768 layers
// and their input gradients (a workgroup covers 8 groups), <64, 48, 1> the first layer (1024 -> 768) and <48, 64, 1> its
// input gradient (4 groups per workgroup: the weight fragments of one group are 168 / 176 registers there).
#include "a2v_gemm.h"

#define CD_ROWS 64
#define CD_TAPS 7
#define CD_SLAB_ROWS (CD_ROWS + CD_TAPS - 1)      // 70

template <int CI, int CO, int GPW>
struct CdGeom {
  static constexpr int GW = 4 * GPW;                       // groups per workgroup
  static constexpr int RCH = GW * CI / 8;                  // 16-B chunks per slab row
  static constexpr int PITCH = GW * CI * 2 + 16;           // 16 consecutive rows start on 16 different 4-bank groups
  static constexpr int SLAB_BYTES = CD_SLAB_ROWS * PITCH;
  static constexpr int LDS = 2 * SLAB_BYTES + GW * CO * 4;    // two slabs + the workgroup's bias values
  static constexpr int CHUNKS = CD_SLAB_ROWS * RCH;
  static constexpr int NLD = (CHUNKS + 255) / 256;
  static constexpr int TCH = CI / 8;                       // K-chunks (8 channels) per tap
  static constexpr int KCH = CD_TAPS * TCH;                // 42 / 56
  static constexpr int KSTEPS = (KCH + 3) / 4;             // 11 / 14
  static constexpr int NJ = CO / 16;                       // 3 / 4
};

#ifdef CD_STAMP
// Diagnostic build only (python -m animal2vec_amd.build --variant cdstamp -DCD_STAMP; tools/dec_probe.py): wave 0 of the first 256
// workgroups stamps s_memtime at eight points of each of its first 24 blocks -- [0] block start, [4] the next slab requested, [5] / [6] the
// first group multiplied / stored, [7] the second group multiplied, [1] ... stored, [2] the next slab's counted wait passed, [3] behind the
// stash and the barrier.  The stamps stay in LDS until the workgroup ends (no vector-memory operation is added inside the loop).
// -DCD_NOSTORE (with it): the output stores compiled out.  The product library contains none of this.
#define CD_ST_BLK 24
__device__ unsigned long long g_cd_stamp[256][CD_ST_BLK][8];
extern "C" int a2v_diag_cd_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_cd_stamp), sizeof(g_cd_stamp)) == hipSuccess ? 0 : -1;
}
#endif
template <int CI, int CO, int GPW>
__global__ __launch_bounds__(256) void conv_dec_kernel(GemmArgs g, int nW, int front) {
  using GM = CdGeom<CI, CO, GPW>;
  constexpr int CD_GW = GM::GW, CD_PITCH = GM::PITCH, CD_SLAB_BYTES = GM::SLAB_BYTES, CD_CHUNKS = GM::CHUNKS, CD_NLD = GM::NLD;
  constexpr int CD_KSTEPS = GM::KSTEPS, TCH = GM::TCH, KCH = GM::KCH, NJ = GM::NJ, RCH = GM::RCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const ConvA& a = g.a;
  const int T = a.T_out;
  // Work order (round 6): the (clip, 64-row block) pairs form ONE sequence over the whole activation tensor and workgroup wl of the
  // nW that share a channel slice takes blocks wl, wl + nW, wl + 2 nW, ...: at any moment the nW x ny resident workgroups read (and
  // write) one contiguous window of nW x 64 rows that moves through the tensor -- an in-order front, as the streaming LayerNorm
  // kernels (a2v_norm.hip) -- instead of 256 separate streams 3 MB apart (a workgroup used to own 32 consecutive blocks of one clip).
  // front = 0 keeps that older order (a2v_debug_set(38, -1)): workgroup wl owns the blocks [wl * bpw, (wl + 1) * bpw), bpw = nW here.
  const int ny = g.G / CD_GW;
  const int by = (int)(blockIdx.x % (unsigned)ny);                // the workgroups that share a row block are dispatched together
  const int wl = (int)(blockIdx.x / (unsigned)ny);
  const int nbc = (T + CD_ROWS - 1) / CD_ROWS;                    // blocks per clip; the last one starts at T - 64 (see block_pos)
  const int total = (g.M / T) * nbc;
  const int gb0 = front ? wl : wl * nW, gstep = front ? nW : 1;
  const int nblk = front ? (total - wl + nW - 1) / nW : min(nW, total - gb0);
  const int g0 = by * CD_GW;                                      // first group of this workgroup

  // ---- weights of this wave's two groups as MFMA fragments.  K-chunk q = 4 s + fq of step s is channels [8 (q % 6), +8)
  // of tap q / 6 (chunks 42, 43 of the last step: zero); fragment row fr of column block j is output channel
  // 8 (fr >> 2) + 4 j + (fr & 3) (j < 2) or 32 + fr (j = 2): rows 4 fq + r of D are then channels 8 fq + 4 j + r / 32 + 4 fq + r
  // (CO = 64: two column-block pairs, channels 32 p + 8 fq + 4 (j & 1) + r; CO = 48: one pair and the single block above)
  bf16x8 wf[GPW][CD_KSTEPS][NJ];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const bf16* wb = (const bf16*)g.B + (long long)(g0 + GPW * wid + gi) * g.b_gstride;
#pragma unroll
    for (int s = 0; s < CD_KSTEPS; ++s) {
      const int q = 4 * s + fq;
      const int tap = q / TCH, c0 = (q - TCH * tap) * 8;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = (NJ == 3 && j == 2) ? 32 + fr : 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3);
        bf16x8 w;
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (bf16)0.f;
        if (q < KCH) w = *(const bf16x8*)(wb + (long long)n * g.ldb + tap * 64 + c0);
        wf[gi][s][j] = w;
      }
    }
  }
  // per-lane K-chunk geometry of the X fragments (same q as above; the zero weight chunks read a valid address): formed where it is
  // used, from an opaque copy of the lane's K-chunk index -- eleven offsets held across the block loop were eleven registers the
  // counted fetch does not have (three spills at 512)
  auto xoff_of = [&](int s_, int fq_, int fr_) {
    int q = 4 * s_ + fq_;
    q = q < KCH ? q : KCH - 1;
    const int tap = q / TCH, c0 = (q - TCH * tap) * 8;
    return (fr_ + tap) * CD_PITCH + c0 * 2;
  };
  // Every block is a full 64 rows: the last one of a clip starts at T - 64 and recomputes rows its predecessor also writes -- same
  // operands, same order, same bits (the launcher refuses C overlapping R).  (uniform: scalar registers)
  auto block_pos = [&](int blk, int& clip_, int& t0_) {
    const int gb = gb0 + blk * gstep;
    clip_ = gb / nbc;
    t0_ = min((gb - clip_ * nbc) * CD_ROWS, T - CD_ROWS);
  };
  // The next slab is fetched by raw-asm loads from clamped addresses into AGPRs (slab rows block_t0 - pad + [0, 70); rows outside the
  // clip and chunks past the slab are zeroed when the registers are stashed) and waited for by hand: vmcnt(N), N = the vector-memory
  // operations certainly issued after the fetch -- the block's unconditional stores and, with a residual, its residual loads.  As
  // compiler-visible loads the wait in front of the stash was vmcnt(0) (the block's output stores are pending beside them), and the 56
  // staging registers + 64-bit per-lane addresses left no room for the fragment ring below.  (The wait itself turned out not to be what
  // bounded the kernel -- profiles/r05_dec_bench.txt; tests/test_abi.py checks that nothing names the AGPRs before the wait.)
  typedef unsigned cd_u32x4 __attribute__((ext_vector_type(4)));
  cd_u32x4 rga[CD_NLD];
  const unsigned xpitch_b = (unsigned)a.pitch * 2u;              // (a slab row of a 10 s clip is < 4 MiB from the clip's base)
  auto fetch_asm = [&](int blk) {
    int clip_, t0_;
    block_pos(blk, clip_, t0_);
    const int t0 = t0_ - a.pad;
    // the clip's base, made uniform explicitly (the asm takes it as an SGPR pair; s_nop 4: a VMEM instruction reading an SGPR that
    // v_readfirstlane has just written needs five wait states, and hipcc's hazard pass does not look inside asm)
    const unsigned long long xb64_ = (unsigned long long)((const bf16*)a.ptr + (long long)clip_ * a.bstride + (long long)g0 * a.goff);
    const bf16* const xb_u = (const bf16*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xb64_ >> 32)) << 32) |
                                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb64_));
    asm volatile("s_nop 4" :: "s"(xb_u));
#pragma unroll
    for (int u = 0; u < CD_NLD; ++u) {
      int idx = tid + 256 * u;
      idx = idx < CD_CHUNKS ? idx : CD_CHUNKS - 1;
      const int row = idx / RCH, ch = idx - row * RCH;
      int rs = t0 + row;
      rs = rs < 0 ? 0 : (rs < a.T_in ? rs : a.T_in - 1);
      // (uniform 64-bit base in SGPRs + a 32-bit byte offset per lane: half the address registers of 64-bit per-lane pointers)
      const unsigned off_ = (unsigned)rs * xpitch_b + (unsigned)ch * 16u;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(rga[u]) : "v"(off_), "s"(xb_u) : "memory");
    }
  };
  auto stash_asm = [&](char* buf, int blk) {
    int clip_, t0_;
    block_pos(blk, clip_, t0_);
    const int t0 = t0_ - a.pad;
#pragma unroll
    for (int u = 0; u < CD_NLD; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / RCH, ch = idx - row * RCH;
      const bool ok = (unsigned)(t0 + row) < (unsigned)a.T_in;
      const cd_u32x4 z = {0u, 0u, 0u, 0u};
      if (idx < CD_CHUNKS) *(cd_u32x4*)(buf + row * CD_PITCH + ch * 16) = ok ? rga[u] : z;
    }
  };
  // operations of one block behind its fetch: per group 4 row blocks x (one 16-B store per column-block pair + one 8-B store at CO = 48),
  // and as many residual loads
  constexpr int CD_ST = GPW * 4 * (NJ / 2 + (NJ == 3 ? 1 : 0));
  // bias of the workgroup's GW x CO output channels in LDS (zeros without F_BIAS): held in registers it cost 24 of the 256
  // arch VGPRs next to 264 weight registers (AGPRs), 56 of slab staging and 48 accumulators
  float* const sBias = (float*)(smem + 2 * CD_SLAB_BYTES);
  for (int i = tid; i < CD_GW * CO; i += 256) sBias[i] = (g.flags & F_BIAS) ? g.bias[(long long)g0 * g.N + i] : 0.f;

  fetch_asm(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < CD_NLD; ++u) asm volatile("" : "+a"(rga[u]));
  stash_asm(smem, 0);
  __syncthreads();
  const bool res_ = (g.flags & F_RES) != 0;
  bf16* const C = (bf16*)g.C;
  const bf16* const R = (const bf16*)g.R;
  // One block = fetch the next slab (registers) -> per group: residual rows (registers), MFMAs, stores -> stash the next slab.
  // Loads, stores and their waits share ONE in-order counter (vmcnt): the residual rows of a group are all requested before
  // its first store (a load issued behind a store cannot be waited for without that store's acknowledgement), and the stores
  // of a full 64-row block are unconditional so that the compiler can count them -- behind a branch it has to assume none was
  // issued and waits for vmcnt(0), i.e. for every store of the block, before the next slab may be written to LDS.
#ifdef CD_STAMP
  __shared__ unsigned long long s_st[CD_ST_BLK][8];
#define CD_STAMP_AT(k_) if (tid == 0 && blk < CD_ST_BLK) s_st[blk][k_] = __builtin_amdgcn_s_memtime();
#else
#define CD_STAMP_AT(k_)
#endif
  for (int blk = 0; blk < nblk; ++blk) {
    const char* const cur = smem + (blk & 1) * CD_SLAB_BYTES;
    CD_STAMP_AT(0)
    if (blk + 1 < nblk) fetch_asm(blk + 1);           // in flight under this block's MFMAs
    CD_STAMP_AT(4)
    int clip, t0c;
    block_pos(blk, clip, t0c);
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int grp = g0 + GPW * wid + gi;
      const long long m0 = (long long)clip * T + t0c + fr;       // row of i = 0
      Raw8<bf16> h8[4][NJ / 2];
      uint2 h4[4];
      if (g.flags & F_RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16* rp = R + (m0 + 16 * i) * g.ldr + (long long)grp * g.c_goff;
#pragma unroll
          for (int p = 0; p < NJ / 2; ++p) h8[i][p].load(rp + 32 * p + 8 * fq);
          if (NJ == 3) h4[i] = *(const uint2*)(rp + 32 + 4 * fq);
        }
      }
      f32x4 acc[4][NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // X fragments by raw-asm ds_read_b128 one K-step AHEAD of the MFMAs that use them, behind counted lgkmcnt waits.  Left to
      // hipcc, every fragment read sat directly in front of its three MFMAs behind `s_waitcnt lgkmcnt(0)` (one wave per SIMD: nothing
      // else to run): ~120 cycles of LDS latency per 48 cycles of matrix work, MFMA-busy 0.22 -- that, not HBM, bounded this kernel.
      // Ring of two K-steps (4 row blocks each); LDS returns in order, so lgkmcnt(4) = "the older K-step's four reads are back".
      const int xs_lds = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)(const_cast<char*>(cur)) + (GPW * wid + gi) * (CI * 2);
      int fq_ = fq, fr_ = fr;
      asm volatile("" : "+v"(fq_), "+v"(fr_));
      // ring of five fragments, four reads in flight: fragment f = 4 s + i (row block i of K-step s) is requested while f - 4 is
      // multiplied; LDS returns in order, so lgkmcnt(n) with n = the reads issued behind f says f is back
      bf16x8 xr[5];
      constexpr int NF = 4 * CD_KSTEPS;
      int xa_ = 0;
#define CD_XREAD(f_)                                                                                                   \
  {                                                                                                                    \
    if (((f_) & 3) == 0) xa_ = xs_lds + xoff_of((f_) >> 2, fq_, fr_);                                                  \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xr[(f_) % 5]) : "v"(xa_), "n"(((f_) & 3) * 16 * CD_PITCH));   \
  }
      static_assert(48 * CD_PITCH < 65536, "ds_read offset field");
#pragma unroll
      for (int f = 0; f < 4; ++f) CD_XREAD(f)
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + 4 < NF) CD_XREAD(f + 4)
        const int behind = NF - 1 - f < 4 ? NF - 1 - f : 4;
        if (behind == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xr[f % 5]));
        else if (behind == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xr[f % 5]));
        else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xr[f % 5]));
        else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xr[f % 5]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[f % 5]));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[f & 3][j] = A2V_MFMA_16x16x32(wf[gi][f >> 2][j], xr[f % 5], acc[f & 3][j], 0, 0, 0);
      }
#undef CD_XREAD
      if (gi == 0) { CD_STAMP_AT(5) } else { CD_STAMP_AT(7) }
      // ---- epilogue of (this block, group): lane (fq, fr) holds row 16 i + fr, channels 32 p + 8 fq + [0, 8) per column-block
      // pair p (and 32 + 4 fq + [0, 4) of the single block at CO = 48)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16* const cp = C + (m0 + 16 * i) * g.ldc + (long long)grp * g.c_goff;
#pragma unroll
        for (int p = 0; p < NJ / 2; ++p) {
          float v8[8];
          const float* bp = sBias + (GPW * wid + gi) * CO + 32 * p + 8 * fq;
          const f32x4 ba = *(const f32x4*)bp, bb = *(const f32x4*)(bp + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v8[r] = acc[i][2 * p][r] + ba[r];
            v8[4 + r] = acc[i][2 * p + 1][r] + bb[r];
          }
          if (g.flags & F_RES) {
            float hh[8];
            h8[i][p].unpack(hh);
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] += hh[e];
          }
#ifdef CD_NOSTORE      /* timing builds only: what the block costs without its output stores (results kept alive, not written) */
          asm volatile("" :: "v"(v8[0]), "v"(v8[1]), "v"(v8[2]), "v"(v8[3]), "v"(v8[4]), "v"(v8[5]), "v"(v8[6]), "v"(v8[7]));
#else
          Vec8<bf16>::store(cp + 32 * p + 8 * fq, v8);
#endif
        }
        if (NJ == 3) {
          f32x4 o4 = acc[i][2] + *(const f32x4*)(sBias + (GPW * wid + gi) * CO + 32 + 4 * fq);
          if (g.flags & F_RES) {
            o4[0] += __uint_as_float(h4[i].x << 16); o4[1] += __uint_as_float(h4[i].x & 0xffff0000u);
            o4[2] += __uint_as_float(h4[i].y << 16); o4[3] += __uint_as_float(h4[i].y & 0xffff0000u);
          }
#ifdef CD_NOSTORE
          asm volatile("" :: "v"(o4));
#else
          Vec4<bf16>::store(cp + 32 + 4 * fq, o4);
#endif
        }
      }
      if (gi == 0) { CD_STAMP_AT(6) }
    }
    CD_STAMP_AT(1)
    if (blk + 1 < nblk) {                             // its readers (block blk - 1) are behind the last barrier
      // (volatile asm statements keep their order: the wait, then one empty statement per register that re-defines it -- every use of
      //  the loaded values sits behind the wait)
      if (res_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * CD_ST) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CD_ST) : "memory");
#pragma unroll
      for (int u = 0; u < CD_NLD; ++u) asm volatile("" : "+a"(rga[u]));
      CD_STAMP_AT(2)
      stash_asm(smem + ((blk + 1) & 1) * CD_SLAB_BYTES, blk + 1);
    }
    __syncthreads();
    CD_STAMP_AT(3)
  }
#ifdef CD_STAMP
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 256)
    for (int b = 0; b < CD_ST_BLK; ++b)
      for (int k_ = 0; k_ < 8; ++k_) g_cd_stamp[blockIdx.x][b][k_] = b < nblk ? s_st[b][k_] : 0ull;
#endif
#undef CD_STAMP_AT
}

int g_a2v_conv_dec = 1;     // a2v_debug_set(13, v): decoder grouped conv through this kernel on / off
int g_a2v_dec_bpw = 0;      // a2v_debug_set(38, v): 0 in-order front, one workgroup per CU; v > 0: v workgroups per channel slice; -1: the older order
template <int CI, int CO, int GPW>
static void conv_dec_go(const GemmArgs& g, hipStream_t stream) {
  using GM = CdGeom<CI, CO, GPW>;
  const ConvA& a = g.a;
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_dec_kernel<CI, CO, GPW>, hipFuncAttributeMaxDynamicSharedMemorySize, GM::LDS);
    attr = true;
  }
  const int nclips = g.M / a.T_out;
  const int ny = g.G / GM::GW;
  const int total = nclips * cdiv(a.T_out, CD_ROWS);                 // 64-row blocks of the tensor
  static int cus_dev[A2V_MAX_DEVICES]; int& cus = cus_dev[a2v_device()];
  if (!cus) {
    hipDeviceProp_t p;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&p, dev);
    cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  if (g_a2v_dec_bpw >= 0) {
    // in-order front: one resident workgroup per CU (512 registers per lane), nW of them per channel slice; few blocks -> fewer
    // workgroups, each with at least four blocks behind the weight fragments it loads (66 KB per wave)
    int nW = cus / ny > 0 ? cus / ny : 1;
    if (g_a2v_dec_bpw > 0) nW = g_a2v_dec_bpw;
    while (nW > 1 && total < 4 * nW) nW >>= 1;
    if (nW > total) nW = total;
    hipLaunchKernelGGL((conv_dec_kernel<CI, CO, GPW>), dim3((unsigned)(nW * ny)), dim3(256), GM::LDS, stream, g, nW, 1);
    return;
  }
  // the older order (a2v_debug_set(38, -1)): a workgroup owns bpw consecutive blocks (round 5 sweep, 768 clones x 2000 frames:
  // 8 / 16 / 32 blocks = 1405 / 1331 / 1277 us)
  int bpw = 32;
  while (bpw > 4 && (long long)cdiv(total, bpw) * ny < 512) bpw >>= 1;
  hipLaunchKernelGGL((conv_dec_kernel<CI, CO, GPW>), dim3((unsigned)(cdiv(total, bpw) * ny)), dim3(256), GM::LDS, stream, g, bpw, 0);
}
bool a2v_launch_conv_dec48(const GemmArgs& g, hipStream_t stream) {
  const ConvA& a = g.a;
  if (!g_a2v_conv_dec) return false;
  if (a.Cg != 64 || a.goff != a.Cv || g.c_goff != g.N || a.taps != CD_TAPS || a.stride != 1 || a.T_in != a.T_out ||
      g.K != CD_TAPS * 64 || g.M % a.T_out != 0 || a.pad >= a.taps || a.T_out < CD_ROWS || a.rowmask || a.rowidx || a.bdiv != 1 ||
      (g.flags & ~(F_BIAS | F_RES)) || a.pitch % 8 || g.ldc % 8 || ((g.flags & F_RES) && g.ldr % 8))
    return false;
  if (g.flags & F_RES) {                       // overlapping blocks write identical bits only if the residual is not the output
    const char *c0 = (const char*)g.C, *c1 = c0 + ((long long)(g.M - 1) * g.ldc + (long long)g.G * g.c_goff) * 2;
    const char *r0 = (const char*)g.R, *r1 = r0 + ((long long)(g.M - 1) * g.ldr + (long long)g.G * g.c_goff) * 2;
    if (r0 < c1 && c0 < r1) return false;
  }
  if (a.Cv == 48 && g.N == 48 && g.G % 8 == 0) { conv_dec_go<48, 48, 2>(g, stream); return true; }
  if (a.Cv == 64 && g.N == 48 && g.G % 4 == 0) { conv_dec_go<64, 48, 1>(g, stream); return true; }
  if (a.Cv == 48 && g.N == 64 && g.G % 4 == 0) { conv_dec_go<48, 64, 1>(g, stream); return true; }
  return false;
}
