// Weight gradient of the grouped 1-D convolutions (pos-conv k19 g16, decoder k7 g16), bf16, gfx950.
//
//   dW[g][n][tap*64 + ci] += sum_m dY[m][g*N + n] * X[m + tap - pad][g*Cin + ci]        (stride 1, Cg = 64)
//
// The general TN kernel (a2v_gemm_fast.hip, 64 x 128 tiles) treats the im2col view as an ordinary operand: every
// 128-wide kk tile (two taps) re-reads the same 64 rows of dY and of X, 24 KB of operands per MFLOP, and it sits on
// the L2 bandwidth at 340 TFLOP/s.  Here one workgroup owns ALL taps of a (group, clip range): per 64-row step it
// stages the dY tile (64 x 64) and ONE slab of X (64 + taps - 1 rows x 64 channels) and forms the B fragment of
// tap j by reading the slab j rows further down (ds_read_b64_tr_b16 on both operands, reduction over rows).
// 18.5 KB staged per 10 MFLOP.  8 waves; wave w owns the kk tiles [w*CPW, (w+1)*CPW) x all 64 n: 4*CPW
// accumulator tiles (CPW = 10 for 19 taps: 160 VGPRs).  Register-prefetched double buffer, one barrier per step;
// the result leaves with fp32 atomics (256 units x 311 KB for the pos-conv).
#include "a2v_gemm.h"

typedef __attribute__((address_space(3))) s16x4 ws_lds_s16x4;

// LDS images are [rows][64] bf16 with a 160-B row pitch (128 B of data + 32 B of padding): the four rows a
// 16-lane group of a tr read touches land on four different 8-bank groups, and -- unlike an XOR swizzle -- the
// address of (row + tap, column tile) is the lane's base address plus a compile-time constant, so the 28 fragment
// reads of a K step need two base registers instead of 56 hoisted (and spilled) addresses.
#define WS_PITCH 160
__device__ __forceinline__ bf16x8 ws_frag(const char* p) {      // rows r .. r+3 (lo) and r+4 .. r+7 (hi) of a 16 x 32 block
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ws_lds_s16x4*)p);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ws_lds_s16x4*)(p + 4 * WS_PITCH));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

#define WS_SLAB_ROWS 84                      // 64 + 20 taps
#define WS_P_BYTES (64 * WS_PITCH)
#define WS_STAGE (WS_P_BYTES + WS_SLAB_ROWS * WS_PITCH)

// CPW = 5 (round 6 experiment: builds with -DA2V_WGRAD_SPLIT + a2v_debug_set(40, 1); not in the product library): the kk tiles of a 19-20-tap convolution split over TWO workgroups
// (blockIdx.z = 0 / 1: tiles [z * 40, z * 40 + 40)), 80 accumulator registers per lane instead of 160, two workgroups per CU (4 waves
// per SIMD), so that one's staging / barrier / restart runs under the other's MFMAs.  Measured: SLOWER -- 488 -> 543 us per launch at
// 96 x 2000 rows, whole step -0.5 % (tools/gpu/r06_wgs.sh): each A fragment now feeds 5 kk tiles instead of 10 (LDS reads per MFMA
// +29 %), both workgroups stage the same tiles, and 128 registers cost a spilled pointer reloaded in the step loop.
template <int CPW, bool MASKED>
__global__ __launch_bounds__(512, CPW == 5 ? 4 : 1) void conv_wgrad_slab_kernel(WgradArgs g, int clips_per_unit) {
  __shared__ __attribute__((aligned(16))) char smem[2 * WS_STAGE];
  if (g.skip_if && *g.skip_if) return;                   // (launched behind conv_wgrad_sparse_kernel: only runs if that one could not)
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvA& a = g.a;
  const int grp = blockIdx.y;
  const int T = a.T_out;
  const int nclips = g.M / T;
  const int c0 = blockIdx.x * clips_per_unit;
  const int c1 = min(nclips, c0 + clips_per_unit);
  if (c0 >= c1) return;
  const int spc = (T + 63) / 64;                       // steps per clip
  const int nsteps = (c1 - c0) * spc;
  const int taps = a.taps;
  const int CT = taps * 4;                             // 16-wide kk tiles

  // staging slots: dY chunk (row tid>>3, chunk tid&7); X slab rows tid>>3 and 64 + (tid>>3)
  const int srow = tid >> 3, sch = tid & 7;
  const bool p_ok = sch * 8 < g.N;
  const bool x_ok = sch * 8 < a.Cv;
  const bool x2 = srow < taps - 1;                     // this thread also stages slab row 64 + srow
  uint4 rp = {0u, 0u, 0u, 0u}, rx0 = rp, rx1 = rp;
  unsigned mkm = a.rowmask ? 0xFFu : 0u;
  asm volatile("" : "+s"(mkm));
  unsigned m0 = 0u, m1 = 0u;                           // row-mask bytes of the two slab rows (in flight with the data)
  bool edge = true;
  unsigned kp, k0, k1;                                 // all-ones where the staged chunk is real data, zero where it is padding / masked
// (unconditional loads from clamped addresses, masked afterwards with an opaque AND: as `if (in range && !mask[row]) load` each
// of the three was its own exec region, and with a row mask -- the first pos-conv layer -- the mask byte's round trip was waited
// for (vmcnt(0), the dY load included) before the data load was even issued: two exposed global round trips per 64-row step)
#define WS_LOAD_M(s_)                                                                                     \
  {                                                                                                      \
    const int cl = c0 + (s_) / spc;                                                                      \
    const int t0 = ((s_) - ((s_) / spc) * spc) * 64;                                                     \
    const int tp = min(t0 + srow, T - 1);                                                                \
    if (p_ok) rp = *(const uint4*)((const bf16*)g.P + ((long long)cl * T + tp) * g.ldp + (long long)grp * g.p_goff + sch * 8); \
    const bf16* xb = (const bf16*)a.ptr + (long long)(cl / a.bdiv) * a.bstride + (long long)grp * a.goff + sch * 8; \
    const unsigned char* mk = a.rowmask ? a.rowmask + (long long)cl * a.T_in : nullptr;                  \
    const int ra = t0 - a.pad + srow, rac = min(max(ra, 0), a.T_in - 1);                                 \
    const int rb = ra + 64, rbc = min(max(rb, 0), a.T_in - 1);                                           \
    /* (thread-constant conditions only -- narrow groups (decoder: 24 of 64 channels) and the taps - 1 extra slab rows */ \
    /* are not worth a load per lane -- nothing that depends on loaded data) */                          \
    if (x_ok) rx0 = *(const uint4*)(xb + (long long)rac * a.pitch);                                      \
    if (x2 && x_ok) rx1 = *(const uint4*)(xb + (long long)rbc * a.pitch);                                \
    if (mk) { m0 = mk[rac]; m1 = mk[x2 ? rbc : rac]; }    /* (raw; consumed in WS_STORE, like the data) */ \
    kp = (p_ok && t0 + srow < T) ? 0xFFFFFFFFu : 0u;                                                     \
    k0 = (x_ok && ra == rac) ? 0xFFFFFFFFu : 0u;                                                         \
    k1 = (x2 && x_ok && rb == rbc) ? 0xFFFFFFFFu : 0u;                                                   \
    /* lanes that never load keep the zeros they were initialised with; the rest needs masking only where a clip edge or a */ \
    /* row mask is in play (uniform): in the steps between, the 12 ANDs per lane are issue time a 32-MFMA step (k = 7) feels */ \
    edge = mk != nullptr || t0 - a.pad < 0 || t0 - a.pad + 63 + taps > a.T_in || t0 + 64 > T;            \
  }
#define WS_STORE_M(buf_)                                                                                  \
  {                                                                                                      \
    char* sp_ = smem + (buf_) * WS_STAGE;                                                                \
    if (edge) {                                                                                          \
      k0 &= 0u - (unsigned)((m0 & mkm) == 0u); k1 &= 0u - (unsigned)((m1 & mkm) == 0u);                   \
      asm volatile("" : "+v"(kp), "+v"(k0), "+v"(k1));                                                   \
      rp.x &= kp; rp.y &= kp; rp.z &= kp; rp.w &= kp;                                                    \
      rx0.x &= k0; rx0.y &= k0; rx0.z &= k0; rx0.w &= k0;                                                \
      rx1.x &= k1; rx1.y &= k1; rx1.z &= k1; rx1.w &= k1;                                                \
    }                                                                                                    \
    *(uint4*)(sp_ + srow * WS_PITCH + sch * 16) = rp;                                                    \
    *(uint4*)(sp_ + WS_P_BYTES + srow * WS_PITCH + sch * 16) = rx0;                                      \
    if (x2) *(uint4*)(sp_ + WS_P_BYTES + (64 + srow) * WS_PITCH + sch * 16) = rx1;                       \
  }

// no row mask (every layer but the first pos-conv one): the plain form -- conditional loads, no masking arithmetic; the
// branch-free form above costs a k = 7 step (32 MFMAs) 4-9 % (same-box A/B)
  const uint4 zero4 = {0u, 0u, 0u, 0u};
#define WS_LOAD_P(s_)                                                                                      \
  {                                                                                                      \
    const int cl = c0 + (s_) / spc;                                                                      \
    const int t0 = ((s_) - ((s_) / spc) * spc) * 64;                                                     \
    rp = zero4; rx0 = zero4; rx1 = zero4;                                                                \
    if (p_ok && t0 + srow < T)                                                                           \
      rp = *(const uint4*)((const bf16*)g.P + ((long long)cl * T + t0 + srow) * g.ldp + (long long)grp * g.p_goff + sch * 8); \
    const bf16* xb = (const bf16*)a.ptr + (long long)(cl / a.bdiv) * a.bstride + (long long)grp * a.goff + sch * 8; \
    const unsigned char* mk = a.rowmask ? a.rowmask + (long long)cl * a.T_in : nullptr;                  \
    const int ra = t0 - a.pad + srow;                                                                    \
    if (x_ok && (unsigned)ra < (unsigned)a.T_in && !(mk && mk[ra])) rx0 = *(const uint4*)(xb + (long long)ra * a.pitch); \
    const int rb = ra + 64;                                                                              \
    if (x2 && x_ok && (unsigned)rb < (unsigned)a.T_in && !(mk && mk[rb])) rx1 = *(const uint4*)(xb + (long long)rb * a.pitch); \
  }
#define WS_STORE_P(buf_)                                                                                   \
  {                                                                                                      \
    char* sp_ = smem + (buf_) * WS_STAGE;                                                                \
    *(uint4*)(sp_ + srow * WS_PITCH + sch * 16) = rp;                                                    \
    *(uint4*)(sp_ + WS_P_BYTES + srow * WS_PITCH + sch * 16) = rx0;                                      \
    if (x2) *(uint4*)(sp_ + WS_P_BYTES + (64 + srow) * WS_PITCH + sch * 16) = rx1;                       \
  }

  f32x4 acc[4][CPW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if constexpr (MASKED) { WS_LOAD_M(0); WS_STORE_M(0); } else { WS_LOAD_P(0); WS_STORE_P(0); }
  __syncthreads();

  const int gq = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_base = (8 * gq + q) * WS_PITCH + 8 * p;
  const int ct0 = wid * CPW + (int)blockIdx.z * 8 * CPW;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    // (unconditional, the last step re-fetches itself: with the loads and their consuming stores behind `if (st + 1 < nsteps)` the
    // wait-count pass saw a path on which a load stays pending across the back edge and put s_waitcnt vmcnt(0) in front of the
    // NEXT load into the same registers -- dY, slab row, second slab row fetched one round trip after the other)
    if constexpr (MASKED) {
      WS_LOAD_M(min(st + 1, nsteps - 1));
      __builtin_amdgcn_sched_barrier(0);    // (the fragment reads / MFMAs below are scheduled on their own, not threaded through the address arithmetic above)
    } else {
      if (st + 1 < nsteps) WS_LOAD_P(st + 1);
    }
    const char* sp_ = smem + buf * WS_STAGE + lane_base;          // this lane's (row 8 gq + q, column 4 p) of the dY tile
    const char* sx_ = sp_ + WS_P_BYTES;                           // ... and of the X slab
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = ws_frag(sp_ + ks * 32 * WS_PITCH + i * 32);
      // kk tile c = wid*CPW + j: tap c >> 2, channel block c & 3; wid is wave-uniform, so the tile's offset from the
      // lane base is a scalar (one v_add per fragment)
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const int c = ct0 + j;
        const bf16x8 fb = ws_frag(sx_ + ks * 32 * WS_PITCH + ((c >> 2) * WS_PITCH + (c & 3) * 32));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = A2V_MFMA_16x16x32(fa[i], fb, acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (MASKED) {
      __builtin_amdgcn_sched_barrier(0);    // (the masking ANDs wait for the global loads: after the step's last MFMA, not among them)
      WS_STORE_M(buf ^ 1);
    } else {
      if (st + 1 < nsteps) WS_STORE_P(buf ^ 1);
    }
    __syncthreads();
  }
#undef WS_LOAD_M
#undef WS_STORE_M
#undef WS_LOAD_P
#undef WS_STORE_P

  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int c = wid * CPW + (int)blockIdx.z * 8 * CPW + j;
    if (c >= CT) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = i * 16 + fq * 4 + r;
        if (n < g.N) atomicAdd(g.W + (long long)grp * g.w_gstride + (long long)n * g.ldw + c * 16 + fr, acc[i][j][r]);
      }
  }
}

int g_a2v_wgrad_slab_split = 0;     // a2v_debug_set(40, v), -DA2V_WGRAD_SPLIT builds only: 19-20 taps: kk tiles split over two workgroups per CU
bool a2v_launch_wgrad_slab(const WgradArgs& g, hipStream_t stream) {
  const ConvA& a = g.a;
  if (a.Cg != 64 || a.stride != 1 || a.T_in != a.T_out || a.taps < 2 || a.taps > 20 || g.N > 64 || g.N % 8 || a.Cv % 8 ||
      g.K != a.taps * 64 || g.M % a.T_out != 0 || a.pad >= a.taps)
    return false;
  const int nclips = g.M / a.T_out;
  int units = 512 / g.G;                               // ~2 units per CU
  if (units < 1) units = 1;
  if (units > nclips) units = nclips;
  const int cpu = cdiv(nclips, units);
  units = cdiv(nclips, cpu);
  dim3 grid(units, g.G, 1);
  if (a.taps * 4 <= 32) {
    if (a.rowmask) hipLaunchKernelGGL((conv_wgrad_slab_kernel<4, true>), grid, dim3(512), 0, stream, g, cpu);
    else hipLaunchKernelGGL((conv_wgrad_slab_kernel<4, false>), grid, dim3(512), 0, stream, g, cpu);
#ifdef A2V_WGRAD_SPLIT     /* experiment builds only (python -m animal2vec_amd.build --variant wgs -DA2V_WGRAD_SPLIT): the 128-register form spills */
  } else if (g_a2v_wgrad_slab_split) {
    const dim3 grid2(units, g.G, 2);
    if (a.rowmask) hipLaunchKernelGGL((conv_wgrad_slab_kernel<5, true>), grid2, dim3(512), 0, stream, g, cpu);
    else hipLaunchKernelGGL((conv_wgrad_slab_kernel<5, false>), grid2, dim3(512), 0, stream, g, cpu);
#endif
  } else {
    if (a.rowmask) hipLaunchKernelGGL((conv_wgrad_slab_kernel<10, true>), grid, dim3(512), 0, stream, g, cpu);
    else hipLaunchKernelGGL((conv_wgrad_slab_kernel<10, false>), grid, dim3(512), 0, stream, g, cpu);
  }
  return true;
}
