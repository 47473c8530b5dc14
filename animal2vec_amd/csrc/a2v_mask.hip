// Span-mask sampling + mask bookkeeping on the device: replaces, per forward, the reference's host-side
// fairseq `compute_mask_indices` call (nn/modalities/base.py:401-413: a numpy loop over the B*clone_batch rows,
// then `torch.from_numpy(mask).to(device)`) and `make_maskinfo`'s two argsorts (base.py:427-455).
//
// Distribution (what fairseq@920a548 `compute_mask_indices` draws for the arguments the reference passes:
// static span length, no_overlap False, min_masks 1, require_same_masks True, mask_dropout, no padding mask):
//   per row  num = max(min_masks, int(mask_prob * T / mask_length + U)),  U ~ U[0,1)
//            min_len = mask_length (or T - num - 1 when T - mask_length <= num)
//            `num` span starts WITHOUT replacement from [0, T - min_len), every start covers mask_length frames,
//            overlapping spans merge (unique), frames >= T are dropped
//   batch    every row is cut down to the SMALLEST masked count of the batch by a uniform draw without
//            replacement from its masked frames; then rint(count * mask_dropout) more frames are released the same way.
// The RNG *stream* of fairseq/numpy is not reproducible (third party, absent from the reference tree: SURVEY 8c,
// parity unpinned); this file defines its own counter-based stream -- a pure function of (seed, row, frame) --
// restated bit for bit in oracle/a2v_oracle.py::span_mask (the parity check of this kernel).
//
// Sampling n of M without replacement = taking the n smallest of M iid keys.  Keys are 46-bit and unique by
// construction ((hash32 << 14) | frame), so "the n smallest" is a threshold: the n-th smallest key is found by a
// bitwise descent with one block-wide count per bit (46 counts over keys held in LDS); no sort, no tie rule.
//
// Two launches (the batch minimum is a grid-wide dependency):
//   span_mask_rows_kernel    one workgroup per row: spans -> provisional mask (global, u8) + masked count
//   span_mask_finish_kernel  one workgroup per row: batch minimum, equalise / dropout, final mask, then
//                            ids_keep (kept frames ascending, pitch K), ids_restore (kept rank | K + masked rank):
//                            the inverse of the reference's `mask.argsort()` with ties in ascending frame order
//                            (the convention of masking.make_maskinfo / DESIGN 2), and K = T - masked count.
#include "a2v_common.h"

#define MASK_THREADS 256
#define MASK_MAX_T 16384        // 14 index bits of the 46-bit keys; LDS: 4 B + 1 B per frame

__device__ __forceinline__ unsigned mix32(unsigned u) {
  u ^= u >> 16; u *= 0x7FEB352Du; u ^= u >> 15; u *= 0x846CA68Bu; u ^= u >> 16;
  return u;
}
// stream `purpose` of (seed, row): two finalizer rounds, the row and the frame enter in different rounds
__device__ __forceinline__ unsigned mask_hash(unsigned seed_lo, unsigned seed_hi, unsigned purpose, unsigned row, unsigned i) {
  const unsigned a = mix32(row * 0x9E3779B1u + seed_lo + purpose * 0x85EBCA6Bu);
  return mix32((a ^ seed_hi) + i * 0xC2B2AE35u + 0x27D4EB2Fu);
}

__device__ __forceinline__ int block_sum_i(int v, int* red) {
  // 256 threads = 4 waves; result in every thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// threshold v = the n-th smallest (1-based) of the unique 46-bit keys (key_hi[i] << 14 | i) over the frames i < M with
// valid[i] != 0 (valid == nullptr: all).  Requires 1 <= n <= number of valid frames.
__device__ __forceinline__ unsigned long long nth_smallest_key(const unsigned* key_hi, const unsigned char* valid, int M, int n,
                                                                int* red) {
  unsigned long long prefix = 0;
  for (int bit = 45; bit >= 0; --bit) {
    const unsigned long long test = prefix | (1ull << bit);
    int c = 0;
    for (int i = threadIdx.x; i < M; i += MASK_THREADS) {
      const unsigned long long k = ((unsigned long long)key_hi[i] << 14) | (unsigned)i;
      c += (k < test && (valid == nullptr || valid[i])) ? 1 : 0;
    }
    c = block_sum_i(c, red);
    if (c < n) prefix = test;       // fewer than n keys below `test`: the n-th smallest is >= test
  }
  return prefix;
}

__global__ __launch_bounds__(MASK_THREADS) void span_mask_rows_kernel(unsigned char* __restrict__ mask, int* __restrict__ counts,
                                                                      int T, double spans_f, int mask_length, int min_masks,
                                                                      unsigned seed_lo, unsigned seed_hi) {
  extern __shared__ unsigned char smem[];
  unsigned* key = (unsigned*)smem;                 // [T]
  unsigned char* sel = smem + (size_t)T * 4;       // [T]
  __shared__ int red[4];
  const unsigned row = blockIdx.x;
  const double u = (double)(mask_hash(seed_lo, seed_hi, 1u, row, 0u) >> 8) * (1.0 / 16777216.0);
  int num = (int)(spans_f + u);
  if (num < min_masks) num = min_masks;
  int min_len = mask_length;
  if (T - min_len <= num) min_len = T - num - 1;
  int M = T - min_len;
  if (M > T) M = T;
  if (M < 1) M = 1;
  if (num > M) num = M;
  for (int i = threadIdx.x; i < M; i += MASK_THREADS) key[i] = mask_hash(seed_lo, seed_hi, 2u, row, (unsigned)i);
  __syncthreads();
  const unsigned long long v = num > 0 ? nth_smallest_key(key, nullptr, M, num, red) : 0ull;
  for (int i = threadIdx.x; i < T; i += MASK_THREADS) {
    const unsigned long long k = ((unsigned long long)(i < M ? key[i] : 0u) << 14) | (unsigned)i;
    sel[i] = (num > 0 && i < M && k <= v) ? 1 : 0;
  }
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < T; i += MASK_THREADS) {
    int m = 0;
    for (int off = 0; off < mask_length; ++off) {
      const int s = i - off;
      if (s >= 0) m |= sel[s];
    }
    mask[(size_t)row * T + i] = (unsigned char)m;
    c += m;
  }
  c = block_sum_i(c, red);
  if (threadIdx.x == 0) counts[row] = c;
}

__global__ __launch_bounds__(MASK_THREADS) void span_mask_finish_kernel(unsigned char* __restrict__ mask, const int* __restrict__ counts,
                                                                        int* __restrict__ ids_keep, int* __restrict__ ids_restore,
                                                                        int* __restrict__ keep_out, int N, int T, double mask_dropout,
                                                                        unsigned seed_lo, unsigned seed_hi) {
  extern __shared__ unsigned char smem[];
  unsigned* key = (unsigned*)smem;                 // [T]
  unsigned char* mk = smem + (size_t)T * 4;        // [T]
  __shared__ int red[4];
  __shared__ int wsum[4];
  const unsigned row = blockIdx.x;
  // ---- smallest masked count of the batch (require_same_masks), then mask_dropout
  int mn = 0x7fffffff;
  for (int r = threadIdx.x; r < N; r += MASK_THREADS) mn = min(mn, counts[r]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
  __syncthreads();
  mn = min(min(red[0], red[1]), min(red[2], red[3]));
  const int cnt = counts[row];
  const int n_final = mn - (int)rint((double)mn * mask_dropout);     // np.rint: half to even
  for (int i = threadIdx.x; i < T; i += MASK_THREADS) {
    const unsigned char m = mask[(size_t)row * T + i];
    mk[i] = m;
    key[i] = mask_hash(seed_lo, seed_hi, 3u, row, (unsigned)i);
  }
  __syncthreads();
  if (n_final < cnt) {                                               // uniform subset of this row's masked frames
    const unsigned long long v = n_final > 0 ? nth_smallest_key(key, mk, T, n_final, red) : 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += MASK_THREADS) {
      const unsigned long long k = ((unsigned long long)key[i] << 14) | (unsigned)i;
      const unsigned char m = (mk[i] && n_final > 0 && k <= v) ? 1 : 0;
      mk[i] = m;
      mask[(size_t)row * T + i] = m;
    }
    __syncthreads();
  }
  // ---- ranks: each thread owns a contiguous run of frames; exclusive scan of the kept flags
  const int K = T - n_final;
  const int per = (T + MASK_THREADS - 1) / MASK_THREADS;
  const int lo = threadIdx.x * per, hi = min(T, lo + per);
  int kept = 0;
  for (int i = lo; i < hi; ++i) kept += mk[i] ? 0 : 1;
  int incl = kept;                                                   // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if ((int)(threadIdx.x & 63) >= o) incl += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int base = incl - kept;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
  int kr = base;                                                     // kept frames before `lo`
  for (int i = lo; i < hi; ++i) {
    if (mk[i]) {
      ids_restore[(size_t)row * T + i] = K + (i - kr);               // masked rank = frames before i that are masked
    } else {
      ids_restore[(size_t)row * T + i] = kr;
      ids_keep[(size_t)row * K + kr] = i;
      ++kr;
    }
  }
  if (row == 0 && threadIdx.x == 0) keep_out[0] = K;
}

// mask (N, T) u8 out (1 = masked); counts: N ints of caller-owned scratch; ids_keep: room for N * T ints, written as
// (N, K) with K = keep_out[0] (known only after the launch: read it back, or size for the worst case);
// ids_restore (N, T); keep_out: one int on the device.  seed: any 64-bit value; the draw is a pure function of
// (seed, row, frame) -- the same call returns the same masks.
extern "C" int a2v_span_mask(unsigned char* mask, int* counts, int* ids_keep, int* ids_restore, int* keep_out, int N, int T,
                             float mask_prob, int mask_length, int min_masks, float mask_dropout, unsigned long long seed,
                             hipStream_t st) {
  if (N <= 0 || T < 2 || T > MASK_MAX_T || mask_length < 1 || mask_length >= T || mask_prob <= 0.f || min_masks < 0 ||
      mask_dropout < 0.f || mask_dropout >= 1.f)
    return A2V_ERR_ARG;
  const double spans_f = (double)mask_prob * (double)T / (double)mask_length;      // + U, truncated, in the kernel
  const size_t lds = (size_t)T * 5;
  const unsigned lo = (unsigned)seed, hi = (unsigned)(seed >> 32);
  static bool attr_dev[A2V_MAX_DEVICES]; bool& attr = attr_dev[a2v_device()];   // (function attributes are per device)
  if (!attr) {      // T > 13107 needs more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)span_mask_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MASK_MAX_T * 5);
    (void)hipFuncSetAttribute((const void*)span_mask_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MASK_MAX_T * 5);
    attr = true;
  }
  hipLaunchKernelGGL(span_mask_rows_kernel, dim3(N), dim3(MASK_THREADS), lds, st, mask, counts, T, spans_f, mask_length, min_masks,
                     lo, hi);
  A2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(span_mask_finish_kernel, dim3(N), dim3(MASK_THREADS), lds, st, mask, counts, ids_keep, ids_restore, keep_out,
                     N, T, (double)mask_dropout, lo, hi);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}

// mask bookkeeping alone (make_maskinfo, base.py:427-455) for a given mask with equal counts per row
// (precomputed_mask): ids_keep (N, K) / ids_restore (N, T) as above; K = T - (masked count of row 0), the reference's
// len_keep (base.py:437).  Rows whose count differs from row 0's make the call meaningless exactly as they do there;
// rows_bad (one int, zeroed by the caller, may be NULL) receives their number.
__global__ __launch_bounds__(MASK_THREADS) void maskinfo_kernel(const unsigned char* __restrict__ mask, int* __restrict__ ids_keep,
                                                                int* __restrict__ ids_restore, int* __restrict__ rows_bad, int T, int K) {
  __shared__ int wsum[4];
  const unsigned row = blockIdx.x;
  const unsigned char* mk = mask + (size_t)row * T;
  const int per = (T + MASK_THREADS - 1) / MASK_THREADS;
  const int lo = threadIdx.x * per, hi = min(T, lo + per);
  int kept = 0;
  for (int i = lo; i < hi; ++i) kept += mk[i] ? 0 : 1;
  int incl = kept;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if ((int)(threadIdx.x & 63) >= o) incl += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int base = incl - kept;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
  const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (total != K) {                      // unequal rows: flag, write nothing out of range
    if (threadIdx.x == 0 && rows_bad) atomicAdd(rows_bad, 1);
    return;
  }
  int kr = base;
  for (int i = lo; i < hi; ++i) {
    if (mk[i]) {
      ids_restore[(size_t)row * T + i] = K + (i - kr);
    } else {
      ids_restore[(size_t)row * T + i] = kr;
      ids_keep[(size_t)row * K + kr] = i;
      ++kr;
    }
  }
}

extern "C" int a2v_maskinfo(const unsigned char* mask, int* ids_keep, int* ids_restore, int* rows_bad, int N, int T, int K,
                            hipStream_t st) {
  if (N <= 0 || T < 1 || K < 0 || K > T) return A2V_ERR_ARG;
  hipLaunchKernelGGL(maskinfo_kernel, dim3(N), dim3(MASK_THREADS), 0, st, mask, ids_keep, ids_restore, rows_bad, T, K);
  A2V_CHECK_LAUNCH();
  return A2V_OK;
}
