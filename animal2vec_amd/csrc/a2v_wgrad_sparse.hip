// Weight gradient of the positional encoder's grouped convolutions where ONE operand is row-sparse (round 5), bf16, gfx950.
//
//   dW[g][n][tap*64 + ci] += sum_m dY[m][g*64 + n] * X[m + tap - pad][g*64 + ci]        (stride 1, 64 channels per group)
//
// The student's positional encoder runs on the masked clones (reference: audio.py:90-113 on x * (1 - mask), base.py:292-314):
//   * first layer  -- X is zero at the masked frames: 152 of 2000 rows are non-zero.  conv_wgrad_slab_kernel<10, true> multiplied
//     the zeros (3.56 ms per 64-clip pass);
//   * last layer   -- only the kept frames of the output are used, so dY exists at 152 of 2000 rows (compact).  The row-gathered
//     TN kernel re-gathered the 19-row im2col window of every kept row from L2 (gemm_tn_bf16_fast_kernel<64, 128>, 2.2 ms).
// With masks of span length 2 the kept frames are scattered singles and pairs: every kept row's +-9 neighbourhood together
// covers the whole sequence, so the DENSE operand has to be read once whatever happens (3.1 GB: the floor), but the reduction
// only has to run over the kept rows.  Per step a workgroup takes 256 consecutive frames of one clone and one group:
//   compact operand  C: the kept rows of the window, <= 128 slots x 64 channels (gathered; slots in k-steps of 32)
//   dense operand    D: the window's rows +- the taps' reach, 256 + taps - 1 rows x 64 channels (a time-contiguous slab)
//   tile (a, tap', b) += sum_slot C[slot][a] * D[pos[slot] + tap'][b]
// i.e. conv_wgrad_slab_kernel's scheme (a fixed operand, a slab read `tap` rows further down, ds_read_b64_tr_b16 on both,
// 160-B row pitch so that a tap is an immediate offset) with the reduction index running over the kept slots: the slab rows of
// a fragment are GATHERED -- a transposing read takes its row from the lane's address, so that costs nothing -- and a window
// with 20 kept rows issues one 32-deep k-step instead of eight.  (256-frame windows, not 128: one workgroup per CU (232
// registers) has ONE step of loads in flight, and at 128 frames that was 19 KB per CU -- 1.82 / 1.34 ms against 1.30 / 0.84 (0.89 / 0.84 with the mask form's epilogue atomics along ci).)
//   mode A (first layer): C = X at the kept frames, D = dY, D row = pos + pad - tap      -> tap' = taps - 1 - tap, a = ci, b = n
//   mode B (last layer):  C = dY (compact rows),    D = X,  D row = pos + tap - pad      -> tap' = tap,            a = n,  b = ci
// The kept positions come from a table built by a small kernel in front (per (clone, window): count, first compact row, the
// window-relative positions), from the row mask (A) or from the ascending row list (B).  The builder checks what this kernel
// relies on -- at most 128 kept frames per 256-frame window; B: an ascending list -- and clears a flag otherwise: this kernel
// then returns at once and the kernel it replaces, launched behind it with the flag as its skip condition, does the work.
#include "a2v_gemm.h"

typedef __attribute__((address_space(3))) s16x4 wsp_lds_s16x4;

#define WSP_PITCH 160
#define WSP_WIN 256                                       // frames per window (the dense slab of a step)
#define WSP_CAP 128                                       // kept-row slots per step: a window that keeps more clears the table's ok flag
                                                          // (the whole call then runs on the kernels this one replaces)
#define WSP_SLAB_ROWS (WSP_WIN + 20)                      // taps <= 20 (276 >= 256 + taps - 1)
#define WSP_SLAB_U ((WSP_SLAB_ROWS + 63) / 64)            // 16-B chunks per thread (512 threads: 64 rows x 8 chunks per pass)
#define WSP_C_BYTES (WSP_CAP * WSP_PITCH)                 // 20480
#define WSP_STAGE (WSP_C_BYTES + WSP_SLAB_ROWS * WSP_PITCH + WSP_CAP)   // + the position list: 64768
#define WSP_ENTRY 144                                     // table entry: int cnt, int base, 8 B pad, u8 pos[128] (window-relative frame)
#define WSP_TAB_HEAD 16                                   // int ok (mode B: the row list is strictly ascending)

__device__ __forceinline__ bf16x8 wsp_frag2(const char* plo, const char* phi) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wsp_lds_s16x4*)plo);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wsp_lds_s16x4*)phi);
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// ---- table builders -----------------------------------------------------------------------------------------------------
// mode A: one workgroup per (clone, window), thread j = frame t0 + j: kept = rowmask == 0
__global__ __launch_bounds__(256) void wsp_table_from_mask_kernel(const unsigned char* __restrict__ mask, char* __restrict__ tab, int nclones,
                                                                  int T, int NW) {
  __shared__ int sN[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int e = blockIdx.x;
  const int c = e / NW, w = e - c * NW, t0 = w * WSP_WIN;                 // (the header is set to 1 by wsp_table_ok_kernel in front)
  const bool kept = t0 + tid < T && mask[(long long)c * T + min(t0 + tid, T - 1)] == 0;
  const unsigned long long bal = __ballot(kept);
  if (lane == 0) sN[wv] = __popcll(bal);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int v = 0; v < 4; ++v) { before += v < wv ? sN[v] : 0; total += sN[v]; }
  char* ent = tab + WSP_TAB_HEAD + (long long)e * WSP_ENTRY;
  if (total > WSP_CAP) { if (tid == 0) *(int*)tab = 0; return; }
  if (kept) ((unsigned char*)ent)[16 + before + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned char)tid;
  if (tid == 0) { ((int*)ent)[0] = total; ((int*)ent)[1] = 0; }
}
// mode B: one workgroup per clone; rows[c*K + i] = c*T + t, strictly ascending (checked -> tab header)
__global__ __launch_bounds__(256) void wsp_table_from_rows_kernel(const int* __restrict__ rows, char* __restrict__ tab, int K, int T, int NW) {
  __shared__ int sFirst[64], sCnt[64];                    // NW <= 64 (T <= 16384)
  const int c = blockIdx.x, tid = threadIdx.x;
  for (int w = tid; w < NW; w += 256) { sFirst[w] = 0; sCnt[w] = 0; }
  __syncthreads();
  bool bad = false;
  for (int i = tid; i < K; i += 256) {
    const int r = rows[(long long)c * K + i], t = r - c * T;
    const int rp = i > 0 ? rows[(long long)c * K + i - 1] : c * T - 1;
    if (t < 0 || t >= T || r <= rp) { bad = true; continue; }
    const int w = t >> 8, wp = (rp - c * T) >> 8;
    if (i == 0 || wp != w) sFirst[w] = i;
    atomicAdd(&sCnt[w], 1);
  }
  __syncthreads();
  for (int w = tid; w < NW; w += 256) bad |= sCnt[w] > WSP_CAP;
  if (__syncthreads_or(bad)) { if (tid == 0) *(int*)tab = 0; return; }     // (the header is set to 1 by wsp_table_ok_kernel in front)
  char* ent0 = tab + WSP_TAB_HEAD + (long long)c * NW * WSP_ENTRY;
  for (int i = tid; i < K; i += 256) {
    const int t = rows[(long long)c * K + i] - c * T, w = t >> 8;
    ((unsigned char*)(ent0 + (long long)w * WSP_ENTRY))[16 + i - sFirst[w]] = (unsigned char)(t & 255);
  }
  for (int w = tid; w < NW; w += 256) {
    int* e = (int*)(ent0 + (long long)w * WSP_ENTRY);
    e[0] = sCnt[w]; e[1] = c * K + sFirst[w];
  }
}
__global__ void wsp_table_ok_kernel(char* tab) { *(int*)tab = 1; }

// ---- the product ---------------------------------------------------------------------------------------------------------
template <bool MODE_A>
__global__ __launch_bounds__(512) void conv_wgrad_sparse_kernel(WgradArgs g, const char* __restrict__ tab, int NW, int clones_per_unit) {
  constexpr int CPW = 10;
  extern __shared__ __attribute__((aligned(16))) char smem[];            // 2 * WSP_STAGE
  if (*(const int*)tab == 0) return;                                    // (mode B: the row list was not ascending)
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvA& a = g.a;
  const int grp = blockIdx.y;
  const int T = a.T_out;
  const int taps = a.taps;
  const int c0 = blockIdx.x * clones_per_unit;
  const int c1 = min(g.m_chunk /* = number of clones (set by the launcher) */, c0 + clones_per_unit);
  if (c0 >= c1) return;
  const int NE = NW;                                                     // table entries (steps) per clone
  const int nsteps = (c1 - c0) * NE;
  const int org = MODE_A ? a.pad - (taps - 1) : -a.pad;                  // slab row 0 = frame t0 + org
  const int srows = WSP_WIN + taps - 1;

  // staging slots of this thread: compact (slot, chunk) x 2, slab (row, chunk) x WSP_SLAB_U
  const int srow = tid >> 3, sch = tid & 7;
  const char* const tab0 = tab + WSP_TAB_HEAD;
  // dense / compact operand bases for (clone, group): D rows have pitch dpitch, C rows cpitch (bytes)
  const long long dpitch = MODE_A ? g.ldp * 2 : a.pitch * 2;
  const long long cpitch = MODE_A ? a.pitch * 2 : g.ldp * 2;
  const char* const Dall = MODE_A ? (const char*)g.P + ((long long)grp * g.p_goff + sch * 8) * 2
                                  : (const char*)a.ptr + ((long long)grp * a.goff + sch * 8) * 2;
  const char* const Call = MODE_A ? (const char*)a.ptr + ((long long)grp * a.goff + sch * 8) * 2
                                  : (const char*)g.P + ((long long)grp * g.p_goff + sch * 8) * 2;

  const uint4 zero4 = {0u, 0u, 0u, 0u};
  uint4 rc[2], rd[WSP_SLAB_U];
  int t_cnt = 0, t_base = 0, t_pos[2] = {0, 0};                          // table entry of the step being FETCHED
  int n_cnt = 0, n_base = 0, n_pos[2] = {0, 0};                          // ... of the step after it (in flight)
  int cur_cnt = 0;                                                       // kept rows of the step being multiplied
#define WSP_TLOAD(s_, cnt_, base_, pos_)                                                                 \
  {                                                                                                      \
    const int s__ = min((s_), nsteps - 1);                                                               \
    const char* e_ = tab0 + ((long long)c0 * NE + s__) * WSP_ENTRY;                                      \
    cnt_ = ((const int*)e_)[0]; base_ = ((const int*)e_)[1];                                             \
    pos_[0] = ((const unsigned char*)e_)[16 + srow]; pos_[1] = ((const unsigned char*)e_)[16 + 64 + srow]; \
  }
// (positions of slots >= cnt are whatever the table holds: those slots are staged as zeros and their slab rows are clamped)
#define WSP_DLOAD(s_)                                                                                     \
  {                                                                                                      \
    const int cl = c0 + (s_) / NE, t0 = ((s_) - ((s_) / NE) * NE) * WSP_WIN;                              \
    const char* db = Dall + (MODE_A ? (long long)cl * T * dpitch : (long long)(cl / a.bdiv) * a.bstride * 2); \
    const char* cb = Call + (MODE_A ? (long long)(cl / a.bdiv) * a.bstride * 2 : 0ll);                   \
    _Pragma("unroll") for (int u = 0; u < WSP_SLAB_U; ++u) {                                             \
      const int row = srow + 64 * u, tt = t0 + org + row;                                                \
      rd[u] = zero4;                                                                                     \
      if (row < srows && (unsigned)tt < (unsigned)T && t_cnt > 0) rd[u] = *(const uint4*)(db + (long long)tt * dpitch); \
    }                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                      \
      const int slot = srow + 64 * u;                                                                    \
      rc[u] = zero4;                                                                                     \
      if (slot < t_cnt) rc[u] = *(const uint4*)(cb + (MODE_A ? (long long)(t0 + t_pos[u]) : (long long)(t_base + slot)) * cpitch); \
    }                                                                                                    \
  }
#define WSP_STORE(buf_)                                                                                   \
  {                                                                                                      \
    char* sp_ = smem + (buf_) * WSP_STAGE;                                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) *(uint4*)(sp_ + (srow + 64 * u) * WSP_PITCH + sch * 16) = rc[u]; \
    _Pragma("unroll") for (int u = 0; u < WSP_SLAB_U; ++u)                                               \
      if (srow + 64 * u < WSP_SLAB_ROWS) *(uint4*)(sp_ + WSP_C_BYTES + (srow + 64 * u) * WSP_PITCH + sch * 16) = rd[u]; \
    if (sch == 0) {                                                                                      \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                      \
        ((unsigned char*)sp_)[WSP_C_BYTES + WSP_SLAB_ROWS * WSP_PITCH + srow + 64 * u] = (unsigned char)(srow + 64 * u < t_cnt ? t_pos[u] : 0); \
    }                                                                                                    \
  }

  f32x4 acc[4][CPW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  WSP_TLOAD(0, t_cnt, t_base, t_pos);
  WSP_TLOAD(1, n_cnt, n_base, n_pos);
  WSP_DLOAD(0);
  WSP_STORE(0);
  cur_cnt = t_cnt;
  __syncthreads();

  const int gq = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_base = (8 * gq + q) * WSP_PITCH + 8 * p;
  const int ct0 = wid * CPW;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    // entry of step st + 1 becomes the one being fetched, the entry of st + 2 goes in flight
    t_cnt = n_cnt; t_base = n_base; t_pos[0] = n_pos[0]; t_pos[1] = n_pos[1];
    if (st + 1 >= nsteps) t_cnt = 0;
    WSP_DLOAD(min(st + 1, nsteps - 1));
    WSP_TLOAD(st + 2, n_cnt, n_base, n_pos);
    const char* const sc_ = smem + buf * WSP_STAGE;
    const char* const sd_ = sc_ + WSP_C_BYTES;
    const unsigned char* const spos = (const unsigned char*)sd_ + WSP_SLAB_ROWS * WSP_PITCH;
    const int nks = (cur_cnt + 31) >> 5;
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = wsp_frag2(sc_ + lane_base + ks * 32 * WSP_PITCH + i * 32, sc_ + lane_base + (ks * 32 + 4) * WSP_PITCH + i * 32);
      const int rlo = spos[ks * 32 + 8 * gq + q], rhi = spos[ks * 32 + 8 * gq + q + 4];
      const char* const dlo = sd_ + rlo * WSP_PITCH + 8 * p;
      const char* const dhi = sd_ + rhi * WSP_PITCH + 8 * p;
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const int c = ct0 + j;                           // kk tile: tap' = c >> 2, dense channel block c & 3
        const int off = (c >> 2) * WSP_PITCH + (c & 3) * 32;
        const bf16x8 fb = wsp_frag2(dlo + off, dhi + off);
#pragma unroll
        for (int i = 0; i < 4; ++i)                       // (mode A: operands exchanged, so that the LANES of a tile run along ci -- the contiguous
                                                          //  index of dW -- in the atomics of the epilogue: 637 -> 159 MB of write traffic)
          acc[i][j] = MODE_A ? A2V_MFMA_16x16x32(fb, fa[i], acc[i][j], 0, 0, 0)
                             : A2V_MFMA_16x16x32(fa[i], fb, acc[i][j], 0, 0, 0);
      }
    }
    WSP_STORE(buf ^ 1);
    cur_cnt = t_cnt;
    __syncthreads();
  }
#undef WSP_TLOAD
#undef WSP_DLOAD
#undef WSP_STORE

  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int c = wid * CPW + j;
    if (c >= taps * 4) continue;
    const int tp = c >> 2;
    const int tap = MODE_A ? taps - 1 - tp : tp;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // accumulator rows 4 fq + r belong to the MFMA's first operand, the lane's column fr to its second: mode B (compact, dense) =
        // (n, ci), mode A (dense, compact) = (n, ci) as well -- ci on the lanes either way
        const int n = MODE_A ? (c & 3) * 16 + fq * 4 + r : i * 16 + fq * 4 + r;
        const int ci = MODE_A ? i * 16 + fr : (c & 3) * 16 + fr;
        atomicAdd(g.W + (long long)grp * g.w_gstride + (long long)n * g.ldw + tap * 64 + ci, acc[i][j][r]);
      }
  }
}

// returns 0: not eligible, 1: launched (mode A), 2: launched (mode B) -- the caller then also launches the kernel this one replaces
// with WgradArgs::skip_if = *ok_flag: it runs only if the table builder cleared the flag (mode B: row list not ascending; either
// mode: a 256-frame window that keeps more than 128 frames).  n_seq: number of sequences (clones) the rows
// belong to (mode B; M / n_seq rows each).
int a2v_launch_wgrad_sparse(const WgradArgs& g_in, int n_seq, hipStream_t stream, const int** ok_flag) {
  const ConvA& a = g_in.a;
  const bool modeA = a.rowmask != nullptr && a.rowidx == nullptr;
  const bool modeB = a.rowmask == nullptr && a.rowidx != nullptr;
  if (!modeA && !modeB) return 0;
  if (a.Cg != 64 || a.Cv != 64 || g_in.N != 64 || a.stride != 1 || a.T_in != a.T_out || a.taps < 2 || a.taps > 20 || a.pad >= a.taps ||
      g_in.K != a.taps * 64 || a.T_out > 16384)
    return 0;
  const int T = a.T_out;
  const int NW = cdiv(T, WSP_WIN);
  long long nclones;
  int K = 0;
  if (modeA) {
    if (g_in.M % T) return 0;
    nclones = g_in.M / T;
  } else {
    if (n_seq <= 0 || g_in.M % n_seq) return 0;
    nclones = n_seq;
    K = g_in.M / n_seq;
  }
  if (nclones * NW * WSP_ENTRY + WSP_TAB_HEAD > A2V_WS_SPTAB_BYTES) return 0;
  char* const ws = a2v_stream_ws(stream);
  if (!ws) return 0;
  char* const tab = ws + A2V_WS_SPTAB;
  hipLaunchKernelGGL(wsp_table_ok_kernel, dim3(1), dim3(1), 0, stream, tab);
  if (modeA) {
    hipLaunchKernelGGL(wsp_table_from_mask_kernel, dim3((int)nclones * NW), dim3(256), 0, stream, a.rowmask, tab, (int)nclones, T, NW);
  } else {
    hipLaunchKernelGGL(wsp_table_from_rows_kernel, dim3((int)nclones), dim3(256), 0, stream, a.rowidx, tab, K, T, NW);
  }
  WgradArgs g = g_in;
  g.m_chunk = (int)nclones;
  int units = 512 / g.G;
  if (units < 1) units = 1;
  if (units > nclones) units = (int)nclones;
  const int cpu = cdiv((int)nclones, units);
  units = cdiv((int)nclones, cpu);
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_sparse_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSP_STAGE);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_sparse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSP_STAGE);
    attr = true;
  }
  dim3 grid(units, g.G, 1);
  if (modeA) hipLaunchKernelGGL((conv_wgrad_sparse_kernel<true>), grid, dim3(512), 2 * WSP_STAGE, stream, g, (const char*)tab, NW, cpu);
  else hipLaunchKernelGGL((conv_wgrad_sparse_kernel<false>), grid, dim3(512), 2 * WSP_STAGE, stream, g, (const char*)tab, NW, cpu);
  if (ok_flag) *ok_flag = (const int*)tab;
  return modeA ? 1 : 2;
}
