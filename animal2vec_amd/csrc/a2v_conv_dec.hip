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
#include "a2v_gemm.h"

#define CD_ROWS 64
#define CD_TAPS 7
#define CD_CG 48
#define CD_GW 8                                   // groups per workgroup
#define CD_SLAB_ROWS (CD_ROWS + CD_TAPS - 1)      // 70
#define CD_PITCH (CD_GW * CD_CG * 2 + 16)         // 784 B: 16 consecutive rows start 4 banks apart (conflict-free b128 reads)
#define CD_SLAB_BYTES (CD_SLAB_ROWS * CD_PITCH)
#define CD_LDS (2 * CD_SLAB_BYTES)                // 109760
#define CD_CHUNKS (CD_SLAB_ROWS * CD_GW * CD_CG / 8)   // 16-B chunks of a slab: 3360
#define CD_NLD ((CD_CHUNKS + 255) / 256)          // per thread: 14
#define CD_KSTEPS 11                              // 7 taps x 6 chunks of 8 channels = 42 chunks = 10.5 K-steps of 4 chunks

__global__ __launch_bounds__(256) void conv_dec48_kernel(GemmArgs g, int blocks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const ConvA& a = g.a;
  const int T = a.T_out;
  const int rows_per_wg = blocks_per_wg * CD_ROWS;
  const int cpc = (T + rows_per_wg - 1) / rows_per_wg;           // row chunks per clip
  const int clip = blockIdx.x / cpc;
  const int t_begin = (blockIdx.x - clip * cpc) * rows_per_wg;
  const int t_end = min(T, t_begin + rows_per_wg);
  const int nblk = (t_end - t_begin + CD_ROWS - 1) / CD_ROWS;
  const int g0 = blockIdx.y * CD_GW;                              // first group of this workgroup

  // ---- weights of this wave's two groups as MFMA fragments.  K-chunk q = 4 s + fq of step s is channels [8 (q % 6), +8)
  // of tap q / 6 (chunks 42, 43 of the last step: zero); fragment row fr of column block j is output channel
  // 8 (fr >> 2) + 4 j + (fr & 3) (j < 2) or 32 + fr (j = 2): rows 4 fq + r of D are then channels 8 fq + 4 j + r / 32 + 4 fq + r
  bf16x8 wf[2][CD_KSTEPS][3];
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    const bf16* wb = (const bf16*)g.B + (long long)(g0 + 2 * wid + gi) * g.b_gstride;
#pragma unroll
    for (int s = 0; s < CD_KSTEPS; ++s) {
      const int q = 4 * s + fq;
      const int tap = q / 6, c0 = (q - 6 * tap) * 8;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int n = j < 2 ? 8 * (fr >> 2) + 4 * j + (fr & 3) : 32 + fr;
        bf16x8 w;
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (bf16)0.f;
        if (q < 42) w = *(const bf16x8*)(wb + (long long)n * g.ldb + tap * 64 + c0);
        wf[gi][s][j] = w;
      }
    }
  }
  // per-lane K-chunk geometry of the X fragments (same q as above; the zero weight chunks read a valid address)
  int xoff[CD_KSTEPS];
#pragma unroll
  for (int s = 0; s < CD_KSTEPS; ++s) {
    int q = 4 * s + fq;
    q = q < 42 ? q : 41;
    const int tap = q / 6, c0 = (q - 6 * tap) * 8;
    xoff[s] = (fr + tap) * CD_PITCH + c0 * 2;
  }

  const bf16* const xb = (const bf16*)a.ptr + (long long)(clip / a.bdiv) * a.bstride + (long long)g0 * a.goff;
  uint4 rg[CD_NLD];
  auto fetch = [&](int blk) {                        // slab rows t_begin + 64 blk - pad + [0, 70): zero outside the clip
    const int t0 = t_begin + blk * CD_ROWS - a.pad;
#pragma unroll
    for (int u = 0; u < CD_NLD; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (CD_GW * CD_CG / 8), ch = idx - row * (CD_GW * CD_CG / 8);
      const int rs = t0 + row;
      rg[u] = make_uint4(0u, 0u, 0u, 0u);
      if (idx < CD_CHUNKS && (unsigned)rs < (unsigned)a.T_in) rg[u] = *(const uint4*)(xb + (long long)rs * a.pitch + ch * 8);
    }
  };
  auto stash = [&](char* buf) {
#pragma unroll
    for (int u = 0; u < CD_NLD; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (CD_GW * CD_CG / 8), ch = idx - row * (CD_GW * CD_CG / 8);
      if (idx < CD_CHUNKS) *(uint4*)(buf + row * CD_PITCH + ch * 16) = rg[u];
    }
  };

  // bias of this lane's output channels (8 + 4 per group)
  float b8[2][8], b4[2][4];
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[gi][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[gi][e] = 0.f;
    if (g.flags & F_BIAS) {
      const float* bp = g.bias + (long long)(g0 + 2 * wid + gi) * g.N;
      Vec8<float>::load(bp + 8 * fq, b8[gi]);
      const f32x4 t4 = *(const f32x4*)(bp + 32 + 4 * fq);
#pragma unroll
      for (int e = 0; e < 4; ++e) b4[gi][e] = t4[e];
    }
  }

  fetch(0);
  stash(smem);
  __syncthreads();
  bf16* const C = (bf16*)g.C;
  const bf16* const R = (const bf16*)g.R;
  for (int blk = 0; blk < nblk; ++blk) {
    const char* const cur = smem + (blk & 1) * CD_SLAB_BYTES;
    if (blk + 1 < nblk) fetch(blk + 1);               // in flight under this block's MFMAs
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      f32x4 acc[4][3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const char* const xs = cur + (2 * wid + gi) * (CD_CG * 2);
#pragma unroll
      for (int s = 0; s < CD_KSTEPS; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 xf = *(const bf16x8*)(xs + xoff[s] + i * 16 * CD_PITCH);
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[gi][s][j], xf, acc[i][j], 0, 0, 0);
        }
      }
      // ---- epilogue of (this block, group): lane (fq, fr) holds row 16 i + fr, channels 8 fq + [0, 8) and 32 + 4 fq + [0, 4)
      const int grp = g0 + 2 * wid + gi;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = t_begin + blk * CD_ROWS + i * 16 + fr;
        if (t < t_end) {
          const long long m = (long long)clip * T + t;
          float v8[8], v4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v8[r] = acc[i][0][r] + b8[gi][r];
            v8[4 + r] = acc[i][1][r] + b8[gi][4 + r];
            v4[r] = acc[i][2][r] + b4[gi][r];
          }
          const long long co = m * g.ldc + (long long)grp * g.c_goff;
          if (g.flags & F_RES) {
            const long long ro = m * g.ldr + (long long)grp * g.c_goff;
            float h8[8];
            Vec8<bf16>::load(R + ro + 8 * fq, h8);
            const f32x4 h4 = Vec4<bf16>::load(R + ro + 32 + 4 * fq);
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] += h8[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] += h4[e];
          }
          Vec8<bf16>::store(C + co + 8 * fq, v8);
          const f32x4 o4 = {v4[0], v4[1], v4[2], v4[3]};
          Vec4<bf16>::store(C + co + 32 + 4 * fq, o4);
        }
      }
    }
    if (blk + 1 < nblk) stash(smem + ((blk + 1) & 1) * CD_SLAB_BYTES);   // its readers (block blk - 1) are behind the last barrier
    __syncthreads();
  }
}

int g_a2v_conv_dec = 1;     // a2v_debug_set(13, v): decoder grouped conv through this kernel on / off
bool a2v_launch_conv_dec48(const GemmArgs& g, hipStream_t stream) {
  const ConvA& a = g.a;
  if (!g_a2v_conv_dec) return false;
  if (a.Cg != 64 || a.Cv != CD_CG || a.goff != CD_CG || g.N != CD_CG || g.c_goff != CD_CG || a.taps != CD_TAPS || a.stride != 1 ||
      a.T_in != a.T_out || g.K != CD_TAPS * 64 || g.G % CD_GW || g.M % a.T_out != 0 || a.pad >= a.taps || a.T_out < CD_ROWS ||
      a.rowmask || a.rowidx || a.bdiv != 1 || (g.flags & ~(F_BIAS | F_RES)) || a.pitch % 8 || g.ldc % 8 || (g.flags & F_RES && g.ldr % 8))
    return false;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_dec48_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS);
    attr = true;
  }
  const int nclips = g.M / a.T_out;
  // rows per workgroup: long enough that the 2 x 33 KB of weights a wave loads are a small share of its traffic, short
  // enough for a few rounds over the CUs
  int bpw = 16;
  while (bpw > 4 && (long long)nclips * cdiv(a.T_out, bpw * CD_ROWS) * (g.G / CD_GW) < 512) bpw >>= 1;
  dim3 grid(nclips * cdiv(a.T_out, bpw * CD_ROWS), g.G / CD_GW, 1);
  hipLaunchKernelGGL(conv_dec48_kernel, grid, dim3(256), CD_LDS, stream, g, bpw);
  return true;
}
