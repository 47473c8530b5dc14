// Grouped 1-D convolution of a ROW-SPARSE input with a dense output (round 5), bf16, 64 channels per group, stride 1, gfx950.
//
//   Y[c][m][g*64 + n] = bias[g*64 + n] + sum_tap sum_ci W[g][n][tap*64 + ci] * X[c][m + tap - pad][g*64 + ci]
//
// where X is non-zero at few rows only.  Two products of the student's positional encoder have this shape (reference audio.py:90-113
// on x * (1 - mask), base.py:292-314):
//   * the first layer's forward   -- X = clone * (1 - mask): 152 of 2000 frames kept            (rows given by the row mask)
//   * the last layer's input gradient -- dY only exists at the kept frames (compact rows), W = the transposed / flipped pack
//                                                                                                  (rows given by the row list)
// Both ran on the dense slab kernel (conv_nt_slab2_kernel: 3.4 ms per 64-clip pass each, multiplying zeros; the second one behind
// a 3.1 GB scatter of the compact gradient into a zero tensor).  Every output row is within 9 frames of some kept row, so the
// output is dense and its 3.1 GB have to be written (the floor: 0.6 ms), but the products only exist per kept row:
//   D_tap[slot][n] = sum_ci X[r_slot][ci] W[n][tap][ci]          (16 slots x 16 n per MFMA pair, K = 64)
//   Y[r_slot - tap + pad][n] += D_tap[slot][n]                     (a scatter-add: 19 output rows per kept row)
// One workgroup (4 waves) walks the 160-frame windows of its clones for ONE group: wave w owns output channels 16 w .. 16 w + 15 of
// the group and keeps ALL its weight fragments in registers (19 taps x 2 k-steps x 4 registers); the window's output lives as an
// fp32 tile in LDS (160 + 2 x 18 rows, the halo rows are never written out), receives the scatter-adds as plain read-modify-write
// -- a tile column belongs to one wave, so no atomics are needed and the order of the sums is the program's: deterministic --
// and leaves as bf16 rows with the bias added.  The kept rows of a window (+- the taps' reach) are found by the workgroup itself (ordered compaction of the
// mask bytes / of the clone's row list: any list, ascending or not) three steps ahead, their data gathered two steps ahead.
// Two workgroups per CU.
#include "a2v_gemm.h"

#define CSP_WIN 160
#define CSP_TAPS 19
#define CSP_TROWS (CSP_WIN + 2 * (CSP_TAPS - 1))          // 196 tile rows
#define CSP_TPITCH 68                                     // floats: a lane's four rows (slots 4 apart: frames >= 4 apart) land on different banks
#define CSP_XPITCH 144                                    // bytes per staged kept row (128 + 16: conflict-free 16-B fragment reads)
#define CSP_CAP 32                                        // kept rows staged per chunk (a window with more takes further chunks, synchronously)
#define CSP_LIST 256                                      // list capacity per window (range = 178 frames)
#define CSP_TILE_BYTES ((CSP_TROWS + 1) * CSP_TPITCH * 4)   // 53584: + one row, so that the slots past the list have 19 rows of their own
#define CSP_XBUF_BYTES (CSP_CAP * CSP_XPITCH)             // 4608
#define CSP_LDS (CSP_TILE_BYTES + 2 * CSP_XBUF_BYTES + 3 * (CSP_LIST * 4 + CSP_LIST) + 16 + 64)

typedef __attribute__((address_space(3))) float csp_lds_float;

struct SpInArgs {
  const bf16* X; long long x_pitch, x_bstride; int x_goff;
  const unsigned char* rowmask; int bdiv;                 // mode mask: X = sequence c / bdiv, kept where rowmask[c][t] == 0
  const int* rowidx; int K;                               // mode rows: X compact, row c*K + i holds frame rowidx[c*K + i] - c*T
  int T, pad, nseq;
  const bf16* W; long long ldb, b_gstride;
  const float* bias;
  bf16* Y; long long ldc; int c_goff;
};

template <bool ROWS>
__global__ __launch_bounds__(256, 2) void conv_sparse_in_kernel(SpInArgs a, int clones_per_unit) {
  constexpr int TAPS = CSP_TAPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const tile = (float*)smem;
  char* const xbuf = smem + CSP_TILE_BYTES;                               // [2][CSP_CAP][CSP_XPITCH]
  int* const lsrc = (int*)(xbuf + 2 * CSP_XBUF_BYTES);                    // [3][CSP_LIST]: a ring -- the list of step s lives in buffer s % 3
  unsigned char* const lrel = (unsigned char*)(lsrc + 3 * CSP_LIST);      // [3][CSP_LIST]
  int* const lcnt = (int*)(lrel + 3 * CSP_LIST);                          // [3] (+ 1 pad)
  int* const swc = lcnt + 4;                                              // [4 passes][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.y, T = a.T, pad = a.pad;
  const int NW = (T + CSP_WIN - 1) / CSP_WIN;
  const int c0 = blockIdx.x * clones_per_unit, c1 = min(a.nseq, c0 + clones_per_unit);
  if (c0 >= c1) return;
  const int nsteps = (c1 - c0) * NW;
  const int fr = lane & 15, fq = lane >> 4;

  // this wave's weights: B fragment (tap, ks): n = 16 wid + fr, ci = 32 ks + 8 fq .. + 8
  bf16x8 wf[TAPS][2];
  {
    const bf16* wrow = a.W + (long long)grp * a.b_gstride + (long long)(16 * wid + fr) * a.ldb + 8 * fq;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *(const bf16x8*)(wrow + t * 64 + ks * 32);
  }
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = a.bias ? a.bias[grp * a.c_goff + (tid & 7) * 8 + e] : 0.f;
  for (int i = tid; i < (CSP_TROWS + 1) * CSP_TPITCH / 4; i += 256) ((f32x4*)tile)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- the kept rows of step s: frames lo .. lo + WIN + TAPS - 2 of clone c, lo = t0 - pad; entry = (source row, frame - lo)
  constexpr int NP = ROWS ? 4 : 1;                         // scan passes (rows mode: K <= 1024 list entries per sequence)
  // scan_load only LOADS (the mask byte / the row-list entries of this thread, raw): anything that looks at the loaded value right
  // away makes hipcc wait vmcnt(0) behind the load -- the full global round trip at the top of every step, and with it every
  // other load and store in flight.  The values are interpreted in list_count, at the end of the step's math.
  int sraw[NP];
  int sfl = 0, srl = 0;                                    // per pass p: bit p = entry in range; byte p = its frame - lo
  int s_lo = 0, s_ok = 0;
  auto scan_load = [&](int s) {
#pragma unroll
    for (int p = 0; p < NP; ++p) sraw[p] = ROWS ? -(1 << 30) : 1;
    s_ok = s < nsteps;
    if (!s_ok) return;
    const int c = c0 + s / NW, lo = (s - (s / NW) * NW) * CSP_WIN - pad;
    s_lo = lo;
    if constexpr (ROWS) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i = tid + 256 * p;
        if (i < a.K) sraw[p] = a.rowidx[(long long)c * a.K + i] - c * T;
      }
    } else {
      const int f = lo + tid;
      if (tid < CSP_WIN + TAPS - 1 && f >= 0 && f < T) sraw[0] = a.rowmask[(long long)c * T + f];
    }
  };
  unsigned long long sbal[NP];
  auto list_count = [&]() {                                // before a barrier: every wave's number of kept entries, per pass
    sfl = 0; srl = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(sraw[p]));      // (opaque: the comparison below is not formed next to the load)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if constexpr (ROWS) {
        const int f = sraw[p];
        if (s_ok && f >= max(s_lo, 0) && f < min(s_lo + CSP_WIN + TAPS - 1, T)) { sfl |= 1 << p; srl |= (f - s_lo) << (8 * p); }
      } else {
        if (s_ok && sraw[0] == 0) { sfl = 1; srl = tid; }
      }
      sbal[p] = __ballot((sfl >> p) & 1);
      if (lane == 0) swc[p * 4 + wid] = __popcll(sbal[p]);
    }
  };
  auto list_write = [&](int buf, int s) {                  // behind that barrier: ordered compaction (pass, wave, lane) of step s's scan
    const int c = c0 + s / NW, lo = (s - (s / NW) * NW) * CSP_WIN - pad;
    int before = 0, total = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int v = 0; v < 4; ++v) total += swc[p * 4 + v];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      int b_ = before;
#pragma unroll
      for (int v = 0; v < 4; ++v) { b_ += v < wid ? swc[p * 4 + v] : 0; before += swc[p * 4 + v]; }
      if ((sfl >> p) & 1) {
        const int slot = b_ + __popcll(sbal[p] & ((1ull << lane) - 1ull));
        const int rel = (srl >> (8 * p)) & 0xFF;
        if (slot < CSP_LIST) {
          lsrc[buf * CSP_LIST + slot] = ROWS ? c * a.K + tid + 256 * p : lo + rel;      // compact row / frame of the sequence
          lrel[buf * CSP_LIST + slot] = (unsigned char)rel;
        }
      }
    }
    if (tid == 0) lcnt[buf] = min(total, CSP_LIST);
  };
  // ---- the kept rows' data: chunk ch (slots 32 ch ..) of the list in buffer lb -> a register -> xbuf[xb]
  auto gather_load = [&](int s, int lb, int ch, uint4& dst) {
    dst = uint4{0u, 0u, 0u, 0u};
    if (s >= nsteps) return;
    const int c = c0 + s / NW;
    const int cnt = lcnt[lb];
    const bf16* xb = a.X + (long long)grp * a.x_goff + (tid & 7) * 8 + (ROWS ? 0ll : (long long)(c / a.bdiv) * a.x_bstride);
    const int slot = ch * CSP_CAP + (tid >> 3);
    if (slot < cnt) dst = *(const uint4*)(xb + (long long)lsrc[lb * CSP_LIST + slot] * a.x_pitch);
  };
  auto gather_store = [&](int xb, const uint4& src) {
    *(uint4*)(xbuf + xb * CSP_XBUF_BYTES + (tid >> 3) * CSP_XPITCH + (tid & 7) * 16) = src;
  };
  // ---- the products of one chunk (slots 64 ch .. of list lb, data in xbuf[xb]) into the tile
  auto chunk_math = [&](int lb, int xb, int ch) {
    const int cnt = lcnt[lb];
    const int n_here = min(cnt - ch * CSP_CAP, CSP_CAP);
    const int nmb = (n_here + 15) >> 4;
    const char* const xs = xbuf + xb * CSP_XBUF_BYTES;
    for (int mb = 0; mb < nmb; ++mb) {
      const bf16x8 a0 = *(const bf16x8*)(xs + (16 * mb + fr) * CSP_XPITCH + fq * 16);
      const bf16x8 a1 = *(const bf16x8*)(xs + (16 * mb + fr) * CSP_XPITCH + 64 + fq * 16);
      const int s0 = ch * CSP_CAP + 16 * mb + 4 * fq;                      // this lane's four slots (accumulator rows)
      const unsigned rel4 = *(const unsigned*)(lrel + lb * CSP_LIST + s0);
      csp_lds_float* trow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (slots past the list multiply zero rows, but must not share a cell with a real slot in the same instruction: their 19 rows
        // are the bottom halo + the extra row, which nobody reads)
        const int rel = s0 + r < cnt ? (int)((rel4 >> (8 * r)) & 0xFFu) : CSP_WIN + TAPS - 1;
        trow[r] = (csp_lds_float*)tile + (rel + TAPS - 1) * CSP_TPITCH + 16 * wid + fr;
      }
      // Plain read-modify-write, one tap after the other: this wave is the only one that touches its 16 tile columns, the four
      // slots of a lane are four different rows, and LDS operations of a wave execute in order -- but tap t of one slot and tap t + 1
      // of its neighbour CAN be the same cell (kept frames come in adjacent pairs), so the taps are not overlapped.  (As
      // ds_add_f32 the scatter ran at about one lane per three clocks per CU: 19.8 ms per call.)
      f32x4 accn = {0.f, 0.f, 0.f, 0.f};                  // the products of tap t + 1 are formed under the read-modify-write of tap t
      accn = A2V_MFMA_16x16x32(a0, wf[0][0], accn, 0, 0, 0);
      accn = A2V_MFMA_16x16x32(a1, wf[0][1], accn, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const f32x4 acc = accn;
        float old[4];
        // (LDS-address-space pointers, not volatile generic ones: those became flat_load / flat_store, each behind its own vmcnt(0).
        // The order that matters is kept by the compiler itself: the four rows of a lane have unrelated bases -- may alias.)
#pragma unroll
        for (int r = 0; r < 4; ++r) old[r] = *(trow[r] - t * CSP_TPITCH);                      // tile row rel - tap + TAPS - 1
        if (t + 1 < TAPS) {
          accn = (f32x4){0.f, 0.f, 0.f, 0.f};
          accn = A2V_MFMA_16x16x32(a0, wf[t + 1 < TAPS ? t + 1 : t][0], accn, 0, 0, 0);
          accn = A2V_MFMA_16x16x32(a1, wf[t + 1 < TAPS ? t + 1 : t][1], accn, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) *(trow[r] - t * CSP_TPITCH) = old[r] + acc[r];
      }
    }
  };

  // ---- prologue: lists of steps 0, 1 and 2, data of step 0 in LDS, data of step 1 in flight
  uint4 xnext, xnew;                                      // data of step st + 1 (loaded a step ago) / of step st + 2 (just requested)
  for (int s0_ = 0; s0_ < 3; ++s0_) {
    scan_load(s0_);
    list_count();
    __syncthreads();
    list_write(s0_, s0_);
    __syncthreads();
  }
  gather_load(0, 0, 0, xnext);
  gather_store(0, xnext);
  gather_load(1, 1, 0, xnext);
  __syncthreads();

  // one step; xn: the data of step st + 1 (requested a step ago), xw: receives the data of step st + 2.  The loop below calls it with the
  // two registers exchanged from step to step (no copy of a register whose load is in flight)
  auto step = [&](int st, uint4& xn, uint4& xw) {
    const int lb = st % 3, xb = st & 1;
    // two steps of lead for the gathered rows (one step was not enough: the step is about as long as a global round trip, and
    // with two workgroups per CU every wave sat in front of its gather_store), three for the scan of the mask / row list
    scan_load(st + 3);
    gather_load(st + 2, (st + 2) % 3, 0, xw);
    const int cnt = lcnt[lb];
    chunk_math(lb, xb, 0);
    for (int ch = 1; ch * CSP_CAP < cnt; ++ch) {          // a window that keeps more than 32 rows (rare with the pretraining masks): synchronous chunks
      __syncthreads();
      uint4 xc;
      gather_load(st, lb, ch, xc);
      gather_store(xb, xc);
      __syncthreads();
      chunk_math(lb, xb, ch);
    }
    list_count();
    __syncthreads();                                      // every scatter-add of the window is in the tile
    // (before the write-out: the wait for the gathered rows is a vmcnt(0) -- the value crosses the loop's back edge -- and behind the
    // write-out it would also wait for the stores just issued)
    gather_store(xb ^ 1, xn);                             // data of step st + 1
    {
      const int c = c0 + st / NW, t0 = (st - (st / NW) * NW) * CSP_WIN;
      bf16* yb = a.Y + ((long long)c * T + t0) * a.ldc + (long long)grp * a.c_goff + (tid & 7) * 8;
#pragma unroll
      for (int u = 0; u < CSP_WIN * 8 / 256; ++u) {
        const int row = (tid >> 3) + 32 * u;
        float* tp = tile + (row + TAPS - 1) * CSP_TPITCH + (tid & 7) * 8;
        const f32x4 v0 = *(const f32x4*)tp, v1 = *(const f32x4*)(tp + 4);
        *(f32x4*)tp = (f32x4){0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(tp + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t0 + row < T) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = (bf16)(v0[e] + bias8[e]); o[4 + e] = (bf16)(v1[e] + bias8[4 + e]); }
          *(bf16x8*)(yb + (long long)row * a.ldc) = o;
        }
      }
    }
    list_write(lb, st + 3);                               // list of step st + 3 (the list of step st has been consumed)
    __syncthreads();
  };
  for (int st = 0; st < nsteps; st += 2) {
    step(st, xnext, xnew);
    if (st + 1 < nsteps) step(st + 1, xnew, xnext);
  }
}

static bool a2v_launch_conv_sparse_in(const SpInArgs& a, int G, hipStream_t stream) {
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_sparse_in_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CSP_LDS);
    (void)hipFuncSetAttribute((const void*)conv_sparse_in_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CSP_LDS);
    attr = true;
  }
  int units = 1024 / G;                                    // ~ 4 units per CU (two resident)
  if (units < 1) units = 1;
  if (units > a.nseq) units = a.nseq;
  const int cpu = cdiv(a.nseq, units);
  units = cdiv(a.nseq, cpu);
  dim3 grid(units, G, 1);
  if (a.rowidx) hipLaunchKernelGGL((conv_sparse_in_kernel<true>), grid, dim3(256), CSP_LDS, stream, a, cpu);
  else hipLaunchKernelGGL((conv_sparse_in_kernel<false>), grid, dim3(256), CSP_LDS, stream, a, cpu);
  return true;
}

extern "C" int a2v_conv_sparse_in(int dtype, const void* X, long long x_pitch, long long x_bstride, int x_goff, const unsigned char* rowmask,
                                  int bdiv, const int* rowidx, int rows_per_seq, int n_seq, int T, int pad, int taps, const void* W,
                                  long long ldb, long long b_gstride, const float* bias, void* Y, long long ldc, int c_goff, int G,
                                  hipStream_t stream) {
  if (dtype != A2V_BF16) return A2V_ERR_DTYPE;
  if (!X || !W || !Y || n_seq <= 0 || T <= 0 || G <= 0 || (!rowmask) == (!rowidx)) return A2V_ERR_ARG;
  if (taps != CSP_TAPS || pad < 0 || pad >= taps || bdiv < 1 || (rowidx && (rows_per_seq <= 0 || rows_per_seq > 1024))) return A2V_ERR_ARG;
  if ((x_pitch | ldb | ldc | b_gstride) % 8 || x_goff % 8 || c_goff % 8 || ((((uintptr_t)X) | ((uintptr_t)W) | ((uintptr_t)Y)) & 15)) return A2V_ERR_ARG;
  SpInArgs a{(const bf16*)X, x_pitch, x_bstride, x_goff, rowmask, bdiv, rowidx, rows_per_seq, T, pad, n_seq, (const bf16*)W, ldb, b_gstride,
             bias, (bf16*)Y, ldc, c_goff};
  a2v_launch_conv_sparse_in(a, G, stream);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}
