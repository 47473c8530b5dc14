// Grouped 1-D convolution (pos-conv k19 g16, decoder k7 g16: forward and input gradient), bf16, gfx950.
//
//   out[m][g*N + n] = epi( sum_tap sum_ci  X[m + tap - pad][g*Cin + ci] * W[g][n][tap*64 + ci] )     (stride 1, Cg = 64)
//
// The implicit-GEMM kernel (a2v_gemm_fast.hip, 256 x 64 tiles) re-stages the A tile for every tap (19 times the
// same rows, shifted by one) and reads 16 KB of LDS fragments per 32 MFMAs.  Here a workgroup keeps ONE slab of X
// (1024 + taps - 1 rows x 64 channels, 133 KB) in LDS for all taps and only streams the 8 KB weight tile of each
// tap (LDS-DMA, double buffered); the A fragment of tap j is the slab read j rows further down.  8 waves x (128
// rows x 64 outputs): 24 fragment reads per 64 MFMAs, one barrier per tap.
// The slab is XOR-swizzled in 16-B chunks by (row >> 1) & 7; rows i*16 + fr + tap share that term for all i, so a
// tap costs a handful of VALU for its two chunk addresses and the 8 row tiles are immediates.
#include "a2v_gemm.h"

typedef __attribute__((address_space(3))) void cs_lds_void_t;
typedef __attribute__((address_space(1))) const void cs_gbl_cvoid_t;

#define CS_BM 1024
#define CS_SLAB_ROWS (CS_BM + 24)                    // taps <= 24
#define CS_SLAB_BYTES (CS_SLAB_ROWS * 128)           // 134144
#define CS_B_BYTES 8192
#define CS_LDS (CS_SLAB_BYTES + 2 * CS_B_BYTES)      // 150528
#define CS_NLD ((CS_SLAB_ROWS * 8 + 511) / 512)      // 16-B chunks per thread: 17

// OutT = float: bf16 operands, fp32 C / R (one of the three bf16 products of a convolution in the bf16x3 parity mode; the second
// and third accumulate through the residual: R = C)
template <typename OutT = bf16>
__global__ __launch_bounds__(512) void conv_nt_slab_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sB = smem + CS_SLAB_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvA& a = g.a;
  const int grp = blockIdx.y;
  const int T = a.T_out;
  const int bpc = (T + CS_BM - 1) / CS_BM;            // row blocks per clip
  const int clip = blockIdx.x / bpc;
  const int t_base = (blockIdx.x - clip * bpc) * CS_BM;
  const int taps = a.taps;
  const int rows_needed = min(CS_BM, T - t_base) + taps - 1;

  // ---- weight tile of tap 0 by LDS-DMA: wave w feeds rows n = 8 w .. 8 w + 7, chunk position lane & 7
  const int lr = lane >> 3;
  const int bn = wid * 8 + lr;
  const int bsrc_chunk = ((lane & 7) ^ ((bn >> 1) & 7)) & 7;
  const char* const bsrc = (const char*)g.B + ((long long)grp * g.b_gstride + (long long)(bn < g.N ? bn : g.N - 1) * g.ldb + bsrc_chunk * 8) * 2;
  __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)bsrc, (cs_lds_void_t*)(sB + wid * 1024), 16, 0, 0);

  // ---- the X slab through registers (clip edges, masked clone rows and channels >= Cv read as zero)
  {
    const bf16* xb = (const bf16*)a.ptr + (long long)(clip / a.bdiv) * a.bstride + (long long)grp * a.goff;
    const unsigned char* mk = a.rowmask ? a.rowmask + (long long)clip * a.T_in : nullptr;
    // Every load is issued unconditionally from a clamped address and masked afterwards with an opaque AND: written as
    // `if (in range && !mask[row]) load`, each chunk was its own exec region -- mask byte load, s_waitcnt vmcnt(0), data load --
    // i.e. CS_NLD global round trips one after the other at the head of every workgroup of a masked layer (and a select
    // with a load behind it is turned back into exactly that branch by the code generator).
    uint4 rg[CS_NLD];
    unsigned keep[CS_NLD], mraw[CS_NLD];
    const int chmax = a.Cv / 8 - 1;
    // (no row mask: the byte is read from the activations themselves and ANDed away -- a branch around the byte load brought the wait back)
    const unsigned char* const mkp = mk ? mk : (const unsigned char*)xb;
    unsigned mkm = mk ? 0xFFu : 0u;
    asm volatile("" : "+s"(mkm));
#pragma unroll
    for (int u = 0; u < CS_NLD; ++u) {
      const int idx = tid + 512 * u;
      const int row = idx >> 3, ch = idx & 7;
      const int rs = t_base - a.pad + row;
      const int rsc = min(max(rs, 0), a.T_in - 1);
      rg[u] = *(const uint4*)(xb + (long long)rsc * a.pitch + min(ch, chmax) * 8);
      mraw[u] = mkp[rsc];                                                        // (raw: nothing touches it before the data is due)
      keep[u] = (row < rows_needed && ch <= chmax && rs == rsc) ? 0xFFFFFFFFu : 0u;
    }
#pragma unroll
    for (int u = 0; u < CS_NLD; ++u) {
      unsigned m = keep[u] & (0u - (unsigned)((mraw[u] & mkm) == 0u));
      asm volatile("" : "+v"(m));
      rg[u].x &= m; rg[u].y &= m; rg[u].z &= m; rg[u].w &= m;
    }
#pragma unroll
    for (int u = 0; u < CS_NLD; ++u) {
      const int idx = tid + 512 * u;
      const int row = idx >> 3, ch = idx & 7;
      if (row < CS_SLAB_ROWS) *(uint4*)(smem + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = rg[u];
    }
  }
  __syncthreads();                                    // also drains the tap-0 weight DMA (vmcnt(0))

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int b_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // + j * 2048, ks flips chunk bit 2 (^ 64 B)
  const int a_row0 = wid * 128 + fr;                                // + i * 16 + tap
  for (int tap = 0; tap < taps; ++tap) {
    const int buf = tap & 1;
    if (tap + 1 < taps)
      __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)(bsrc + (tap + 1) * 128),
                                       (cs_lds_void_t*)(sB + (buf ^ 1) * CS_B_BYTES + wid * 1024), 16, 0, 0);
    const char* sb_ = sB + buf * CS_B_BYTES;
    const int ar = a_row0 + tap;
    const int a_off = ar * 128 + ((fq ^ ((ar >> 1) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(sb_ + (b_off ^ (ks << 6)) + j * 2048);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16x8 fa = *(const bf16x8*)(smem + (a_off ^ (ks << 6)) + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = A2V_MFMA_16x16x32(fa, fb[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                                  // weight tile tap+1 landed (vmcnt(0)), tile `buf` free
  }

  // ---- epilogue: two 64 x 64 stages per wave through the (now dead) slab
  float* stage = (float*)smem + wid * (64 * 68);
  const int m_limit = (clip + 1) * T;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(i * 16 + fq * 4 + r) * 68 + j * 16 + fr] = acc[h * 4 + i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    wave_epilogue<OutT>(g, stage, lane, clip * T + t_base + wid * 128 + h * 64, 0, grp, 0, m_limit);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form (64 outputs per group, bias / residual epilogues): 384-row slabs, TWO workgroups per CU.
// The 1024-row form holds one 150 KB workgroup per CU, whose slab load (through registers, latency bound), 19 taps of
// MFMAs and LDS-staged epilogue run strictly one after the other: the matrix pipe idles through the first and the last.
// With 52 KB slabs two workgroups share a CU and one's load / store phases run under the other's MFMAs.  What makes the
// small slab possible: no epilogue staging (the staged epilogue needs 17 KB per wave) -- swapped MFMA operands
// (D = W_frag x X_frag) and a permuted weight-row order give each lane 2 x 8 consecutive output channels of one row, stored
// straight from the accumulators -- and 48 accumulator registers per wave (3 x 4 tiles) instead of 128.
#define CS2_BM 384
#define CS2_SLAB_ROWS (CS2_BM + 24)
#define CS2_SLAB_BYTES (CS2_SLAB_ROWS * 128)         // 52224
#define CS2_LDS (CS2_SLAB_BYTES + 2 * CS_B_BYTES)    // 68608: two per CU
#define CS2_NLD ((CS2_SLAB_ROWS * 8 + 511) / 512)    // 7

__global__ __launch_bounds__(512, 2) void conv_nt_slab2_kernel(GemmArgs g, int group_fastest) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sB = smem + CS2_SLAB_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvA& a = g.a;
  // launch order (a2v_debug_set(35)): 0 = one group after the other (grid (row blocks, G): every pass over the tensor touches one
  // 128-B segment of every 2-KB row); 1 = 1-D grid, group fastest: the G workgroups of one row block are dispatched together, so a
  // row's segments are read -- and its output segments written -- close in time (whole DRAM pages)
  const int grp = group_fastest ? (int)(blockIdx.x % (unsigned)g.G) : (int)blockIdx.y;
  const int bx = group_fastest ? (int)(blockIdx.x / (unsigned)g.G) : (int)blockIdx.x;
  const int T = a.T_out;
  const int bpc = (T + CS2_BM - 1) / CS2_BM;
  const int clip = bx / bpc;
  const int t_base = (bx - clip * bpc) * CS2_BM;
  const int taps = a.taps;
  const int rows_here = min(CS2_BM, T - t_base);
  const int rows_needed = rows_here + taps - 1;

  // ---- weight tile of tap 0 by LDS-DMA: LDS row p = 16 j + f holds output channel 32 (j >> 1) + 8 (f >> 2) + 4 (j & 1) + (f & 3)
  const int lr = lane >> 3;
  const int bp = wid * 8 + lr;                              // LDS row this lane feeds
  const int bn = 32 * (bp >> 5) + 8 * ((bp & 15) >> 2) + 4 * ((bp >> 4) & 1) + (bp & 3);
  const int bsrc_chunk = ((lane & 7) ^ ((bp >> 1) & 7)) & 7;
  const char* const bsrc = (const char*)g.B + ((long long)grp * g.b_gstride + (long long)bn * g.ldb + bsrc_chunk * 8) * 2;
  __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)bsrc, (cs_lds_void_t*)(sB + wid * 1024), 16, 0, 0);

  {
    const bf16* xb = (const bf16*)a.ptr + (long long)(clip / a.bdiv) * a.bstride + (long long)grp * a.goff;
    const unsigned char* mk = a.rowmask ? a.rowmask + (long long)clip * a.T_in : nullptr;
    // Every load is issued unconditionally from a clamped address and masked afterwards with an opaque AND: written as
    // `if (in range && !mask[row]) load`, each chunk was its own exec region -- mask byte load, s_waitcnt vmcnt(0), data load --
    // i.e. CS2_NLD global round trips one after the other at the head of every workgroup of a masked layer (and a select
    // with a load behind it is turned back into exactly that branch by the code generator).
    uint4 rg[CS2_NLD];
    unsigned keep[CS2_NLD], mraw[CS2_NLD];
    const int chmax = a.Cv / 8 - 1;
    // (no row mask: the byte is read from the activations themselves and ANDed away -- a branch around the byte load brought the wait back)
    const unsigned char* const mkp = mk ? mk : (const unsigned char*)xb;
    unsigned mkm = mk ? 0xFFu : 0u;
    asm volatile("" : "+s"(mkm));
#pragma unroll
    for (int u = 0; u < CS2_NLD; ++u) {
      const int idx = tid + 512 * u;
      const int row = idx >> 3, ch = idx & 7;
      const int rs = t_base - a.pad + row;
      const int rsc = min(max(rs, 0), a.T_in - 1);
      rg[u] = *(const uint4*)(xb + (long long)rsc * a.pitch + min(ch, chmax) * 8);
      mraw[u] = mkp[rsc];                                                        // (raw: nothing touches it before the data is due)
      keep[u] = (row < rows_needed && ch <= chmax && rs == rsc) ? 0xFFFFFFFFu : 0u;
    }
#pragma unroll
    for (int u = 0; u < CS2_NLD; ++u) {
      unsigned m = keep[u] & (0u - (unsigned)((mraw[u] & mkm) == 0u));
      asm volatile("" : "+v"(m));
      rg[u].x &= m; rg[u].y &= m; rg[u].z &= m; rg[u].w &= m;
    }
#pragma unroll
    for (int u = 0; u < CS2_NLD; ++u) {
      const int idx = tid + 512 * u;
      const int row = idx >> 3, ch = idx & 7;
      if (row < CS2_SLAB_ROWS) *(uint4*)(smem + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = rg[u];
    }
  }
  __syncthreads();                                    // also drains the tap-0 weight DMA (vmcnt(0))

  f32x4 acc[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int b_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // + j * 2048, ks flips chunk bit 2 (^ 64 B)
  const int a_row0 = wid * 48 + fr;                                 // + i * 16 + tap
  for (int tap = 0; tap < taps; ++tap) {
    const int buf = tap & 1;
    if (tap + 1 < taps)
      __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)(bsrc + (tap + 1) * 128),
                                       (cs_lds_void_t*)(sB + (buf ^ 1) * CS_B_BYTES + wid * 1024), 16, 0, 0);
    const char* sb_ = sB + buf * CS_B_BYTES;
    const int ar = a_row0 + tap;
    const int a_off = ar * 128 + ((fq ^ ((ar >> 1) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(sb_ + (b_off ^ (ks << 6)) + j * 2048);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const bf16x8 fa = *(const bf16x8*)(smem + (a_off ^ (ks << 6)) + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = A2V_MFMA_16x16x32(fb[j], fa, acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                                  // weight tile tap+1 landed (vmcnt(0)), tile `buf` free
  }

  // ---- epilogue: lane (fq, fr) holds row 48 wid + 16 i + fr, channels 32 p + 8 fq + [0, 8) for p = 0, 1
  bf16* const C = (bf16*)g.C;
  const bf16* const R = (const bf16*)g.R;
  float b8[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[p][e] = 0.f;
    if (g.flags & F_BIAS) Vec8<float>::load(g.bias + (long long)grp * g.N + 32 * p + 8 * fq, b8[p]);
  }
  Raw8<bf16> h8[3][2];
  if (g.flags & F_RES) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int rl = wid * 48 + i * 16 + fr;
      const long long m = (long long)clip * T + t_base + (rl < rows_here ? rl : 0);
#pragma unroll
      for (int p = 0; p < 2; ++p) h8[i][p].load(R + m * g.ldr + (long long)grp * g.c_goff + 32 * p + 8 * fq);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int rl = wid * 48 + i * 16 + fr;
    if (rl < rows_here) {
      const long long m = (long long)clip * T + t_base + rl;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float v8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v8[r] = acc[i][2 * p][r] + b8[p][r];
          v8[4 + r] = acc[i][2 * p + 1][r] + b8[p][4 + r];
        }
        if (g.flags & F_RES) {
          float hh[8];
          h8[i][p].unpack(hh);
#pragma unroll
          for (int e = 0; e < 8; ++e) v8[e] += hh[e];
        }
        Vec8<bf16>::store(C + m * g.ldc + (long long)grp * g.c_goff + 32 * p + 8 * fq, v8);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3 (dtype A2V_BF16X3S): the three products of a split convolution -- X_hi W_hi + X_lo W_hi + X_hi W_lo -- in ONE pass.
// Both operands arrive as [hi ; lo] stacked images (a2v_split_bf16 mode 2): the workgroup keeps a 384-row slab of X_hi AND of
// X_lo in LDS (2 x 52 KB), streams the hi and the lo weight tile of every tap (2 x 8 KB, double buffered) and issues three
// MFMAs where the bf16 kernel issues one; fp32 bias / residual / output straight from the accumulators.  As three launches of
// conv_nt_slab_kernel<float> the second and third re-read and re-write the fp32 output through R = C (20 B per element
// instead of 4) and the X_hi slab is staged twice.  64 outputs per group (the pos-conv); anything else keeps the three launches.
#define CS3_LDS (2 * CS2_SLAB_BYTES + 4 * CS_B_BYTES)    // 137216

__global__ __launch_bounds__(512) void conv_nt_slab3_kernel(GemmArgs g, long long a_lo, long long b_lo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sXl = smem + CS2_SLAB_BYTES;                 // the X_lo slab
  char* const sB = smem + 2 * CS2_SLAB_BYTES;              // [buf][hi, lo] weight tiles
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvA& a = g.a;
  const int grp = blockIdx.y;
  const int T = a.T_out;
  const int bpc = (T + CS2_BM - 1) / CS2_BM;
  const int clip = blockIdx.x / bpc;
  const int t_base = (blockIdx.x - clip * bpc) * CS2_BM;
  const int taps = a.taps;
  const int rows_here = min(CS2_BM, T - t_base);
  const int rows_needed = rows_here + taps - 1;

  // weight tiles of tap 0 (hi and lo) by LDS-DMA; LDS row p = 16 j + f holds output channel 32 (j >> 1) + 8 (f >> 2) + 4 (j & 1) + (f & 3)
  const int lr = lane >> 3;
  const int bp = wid * 8 + lr;
  const int bn = 32 * (bp >> 5) + 8 * ((bp & 15) >> 2) + 4 * ((bp >> 4) & 1) + (bp & 3);
  const int bsrc_chunk = ((lane & 7) ^ ((bp >> 1) & 7)) & 7;
  const char* const bsrc = (const char*)g.B + ((long long)grp * g.b_gstride + (long long)bn * g.ldb + bsrc_chunk * 8) * 2;
  const char* const bsrc_lo = bsrc + b_lo * 2;
  __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)bsrc, (cs_lds_void_t*)(sB + wid * 1024), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)bsrc_lo, (cs_lds_void_t*)(sB + CS_B_BYTES + wid * 1024), 16, 0, 0);

  {   // the two X slabs through registers (clip edges, masked clone rows and channels >= Cv read as zero): as in conv_nt_slab2_kernel
    const bf16* xb = (const bf16*)a.ptr + (long long)(clip / a.bdiv) * a.bstride + (long long)grp * a.goff;
    const unsigned char* mk = a.rowmask ? a.rowmask + (long long)clip * a.T_in : nullptr;
    const unsigned char* const mkp = mk ? mk : (const unsigned char*)xb;
    unsigned mkm = mk ? 0xFFu : 0u;
    asm volatile("" : "+s"(mkm));
    const int chmax = a.Cv / 8 - 1;
    uint4 rh[CS2_NLD], rl[CS2_NLD];
    unsigned keep[CS2_NLD], mraw[CS2_NLD];
#pragma unroll
    for (int u = 0; u < CS2_NLD; ++u) {
      const int idx = tid + 512 * u;
      const int row = idx >> 3, ch = idx & 7;
      const int rs = t_base - a.pad + row;
      const int rsc = min(max(rs, 0), a.T_in - 1);
      const bf16* src = xb + (long long)rsc * a.pitch + min(ch, chmax) * 8;
      rh[u] = *(const uint4*)src;
      rl[u] = *(const uint4*)(src + a_lo);
      mraw[u] = mkp[rsc];
      keep[u] = (row < rows_needed && ch <= chmax && rs == rsc) ? 0xFFFFFFFFu : 0u;
    }
#pragma unroll
    for (int u = 0; u < CS2_NLD; ++u) {
      unsigned m = keep[u] & (0u - (unsigned)((mraw[u] & mkm) == 0u));
      asm volatile("" : "+v"(m));
      rh[u].x &= m; rh[u].y &= m; rh[u].z &= m; rh[u].w &= m;
      rl[u].x &= m; rl[u].y &= m; rl[u].z &= m; rl[u].w &= m;
      const int idx = tid + 512 * u;
      const int row = idx >> 3, ch = idx & 7;
      if (row < CS2_SLAB_ROWS) {
        const int off = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
        *(uint4*)(smem + off) = rh[u];
        *(uint4*)(sXl + off) = rl[u];
      }
    }
  }
  __syncthreads();                                    // also drains the tap-0 weight DMA (vmcnt(0))

  f32x4 acc[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int b_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // + j * 2048, ks flips chunk bit 2 (^ 64 B)
  const int a_row0 = wid * 48 + fr;                                 // + i * 16 + tap
  for (int tap = 0; tap < taps; ++tap) {
    const int buf = tap & 1;
    if (tap + 1 < taps) {
      char* const d = sB + (buf ^ 1) * 2 * CS_B_BYTES + wid * 1024;
      __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)(bsrc + (tap + 1) * 128), (cs_lds_void_t*)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((cs_gbl_cvoid_t*)(bsrc_lo + (tap + 1) * 128), (cs_lds_void_t*)(d + CS_B_BYTES), 16, 0, 0);
    }
    const char* sbh = sB + buf * 2 * CS_B_BYTES;
    const char* sbl = sbh + CS_B_BYTES;
    const int ar = a_row0 + tap;
    const int a_off = ar * 128 + ((fq ^ ((ar >> 1) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fbh[4], fbl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fbh[j] = *(const bf16x8*)(sbh + (b_off ^ (ks << 6)) + j * 2048);
        fbl[j] = *(const bf16x8*)(sbl + (b_off ^ (ks << 6)) + j * 2048);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const bf16x8 fah = *(const bf16x8*)(smem + (a_off ^ (ks << 6)) + i * 2048);
        const bf16x8 fal = *(const bf16x8*)(sXl + (a_off ^ (ks << 6)) + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = A2V_MFMA_16x16x32(fbh[j], fal, acc[i][j], 0, 0, 0);     // small terms first
          acc[i][j] = A2V_MFMA_16x16x32(fbl[j], fah, acc[i][j], 0, 0, 0);
          acc[i][j] = A2V_MFMA_16x16x32(fbh[j], fah, acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                  // weight tiles of tap + 1 landed (vmcnt(0)), tiles `buf` free
  }

  // epilogue: lane (fq, fr) holds row 48 wid + 16 i + fr, channels 32 p + 8 fq + [0, 8) for p = 0, 1 (fp32: two 16-B stores each)
  float* const C = (float*)g.C;
  const float* const R = (const float*)g.R;
  float b8[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[p][e] = 0.f;
    if (g.flags & F_BIAS) Vec8<float>::load(g.bias + (long long)grp * g.N + 32 * p + 8 * fq, b8[p]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int rl_ = wid * 48 + i * 16 + fr;
    if (rl_ < rows_here) {
      const long long m = (long long)clip * T + t_base + rl_;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float v8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v8[r] = acc[i][2 * p][r] + b8[p][r];
          v8[4 + r] = acc[i][2 * p + 1][r] + b8[p][4 + r];
        }
        if (g.flags & F_RES) {
          float hh[8];
          Vec8<float>::load(R + m * g.ldr + (long long)grp * g.c_goff + 32 * p + 8 * fq, hh);
#pragma unroll
          for (int e = 0; e < 8; ++e) v8[e] += hh[e];
        }
        Vec8<float>::store(C + m * g.ldc + (long long)grp * g.c_goff + 32 * p + 8 * fq, v8);
      }
    }
  }
}

// dtype A2V_BF16X3S: [hi ; lo] stacked operand images, the three products in one pass (64 outputs per group, bias / residual)
bool a2v_launch_conv_slab3(const GemmArgs& g, long long a_lo, long long b_lo, hipStream_t stream) {
  const ConvA& a = g.a;
  if (a.Cg != 64 || a.stride != 1 || a.T_in != a.T_out || a.taps < 2 || a.taps > 24 || g.N != 64 || a.Cv % 8 || g.K != a.taps * 64 ||
      g.M % a.T_out != 0 || a.pad >= a.taps || a.T_out < 256 || a.rowidx || (g.flags & ~(F_BIAS | F_RES)) || g.ldc % 4 || g.c_goff % 4 ||
      ((g.flags & F_RES) && g.ldr % 4) || (a_lo * 2) % 16 || (b_lo * 2) % 16)
    return false;
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_nt_slab3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CS3_LDS);
    attr = true;
  }
  dim3 grid((g.M / a.T_out) * cdiv(a.T_out, CS2_BM), g.G, 1);
  hipLaunchKernelGGL(conv_nt_slab3_kernel, grid, dim3(512), CS3_LDS, stream, g, a_lo, b_lo);
  return true;
}

int g_a2v_slab_order = 0;    // a2v_debug_set(35, v): bit 0: conv_nt_slab2 launches group-fastest, bit 1: conv_dec launches group-half-fastest
int g_a2v_conv_slab2 = 1;    // a2v_debug_set(14, v): 384-row two-per-CU form of the slab kernel on / off
bool a2v_launch_conv_slab(const GemmArgs& g, hipStream_t stream) {
  const ConvA& a = g.a;
  if (a.Cg != 64 || a.stride != 1 || a.T_in != a.T_out || a.taps < 2 || a.taps > 24 || g.N > 64 || g.N % 4 ||
      a.Cv % 8 || g.K != a.taps * 64 || g.M % a.T_out != 0 || a.pad >= a.taps || a.T_out < 256)
    return false;
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_nt_slab_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS);
    (void)hipFuncSetAttribute((const void*)conv_nt_slab_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS);
    (void)hipFuncSetAttribute((const void*)conv_nt_slab2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CS2_LDS);
    attr = true;
  }
  const int nclips = g.M / a.T_out;
  if (g_a2v_conv_slab2 && g.N == 64 && !(g.flags & ~(F_BIAS | F_RES)) && g.ldc % 8 == 0 && g.c_goff % 8 == 0 &&
      (!(g.flags & F_RES) || g.ldr % 8 == 0)) {
    const long long nb2 = (long long)nclips * cdiv(a.T_out, CS2_BM);
    const int gf = (g_a2v_slab_order & 1) && nb2 * g.G < (1ll << 31) ? 1 : 0;
    dim3 grid2(gf ? (unsigned)(nb2 * g.G) : (unsigned)nb2, gf ? 1 : g.G, 1);
    hipLaunchKernelGGL(conv_nt_slab2_kernel, grid2, dim3(512), CS2_LDS, stream, g, gf);
    return true;
  }
  dim3 grid(nclips * cdiv(a.T_out, CS_BM), g.G, 1);
  hipLaunchKernelGGL(conv_nt_slab_kernel<bf16>, grid, dim3(512), CS_LDS, stream, g);
  return true;
}

// bf16 operands, fp32 outputs (dtype A2V_BF16X3 with 64-wide taps: one bf16 product of a split convolution)
bool a2v_launch_conv_slab_o32(const GemmArgs& g, hipStream_t stream) {
  const ConvA& a = g.a;
  if (a.Cg != 64 || a.stride != 1 || a.T_in != a.T_out || a.taps < 2 || a.taps > 24 || g.N > 64 || g.N % 4 ||
      a.Cv % 8 || g.K != a.taps * 64 || g.M % a.T_out != 0 || a.pad >= a.taps || a.T_out < 256 || a.rowidx)
    return false;
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_nt_slab_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS);
    attr = true;
  }
  dim3 grid((g.M / a.T_out) * cdiv(a.T_out, CS_BM), g.G, 1);
  hipLaunchKernelGGL(conv_nt_slab_kernel<float>, grid, dim3(512), CS_LDS, stream, g);
  return true;
}
