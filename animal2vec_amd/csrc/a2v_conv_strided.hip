// Dense strided Conv1d of the feature extractor (nn/utils.py:1085-1091: Conv1d(cin, 512, k, stride, bias=False,
// padding=ceil(stride / 2)), layers 1-4 of the large recipe: k 10 / s 5, then 3 x (k 3 / s 2)) on the persistent
// 256 x 256 8-phase GEMMs, forward (NT) and weight gradient (TN).
//
// With channels-last input the im2col row of output frame t is the k*C CONTIGUOUS elements that start at frame
// t*s - pad, and because T_in = T_out * s (every shipped layer) consecutive output frames are s*C elements apart ACROSS
// clip boundaries too: the whole batch is one row-major matrix with overlapping rows -- A[m][kk] = X_flat[(m*s - pad)*C + kk],
// pitch s*C -- which is exactly what gemm_nt_bf16_p8_kernel / gemm_tn_bf16_p8_kernel take (arbitrary pitch, LDS-DMA
// straight from global).  The implicit-im2col loaders of the 128 x 128 kernels (a2v_gemm_fast.hip) ran these at
// 0.20-0.25 of the MFMA peak.
//
// What differs from the convolution: the first r0 = ceil(pad / s) and the last r1 output frames of every clip have taps
// outside [0, T_in) -- zeros for the convolution, the neighbouring clip's frames in the flat view (and memory outside the
// tensor for the first / last rows of the batch, which is why the GEMM runs on rows [r0, M - r1) only).  <= 2 rows per clip:
//   forward          conv_edge_nt_kernel recomputes those rows with zero padding and overwrites them;
//   weight gradient  conv_edge_tn_kernel adds dY[m] x (correct - used) for those rows: minus the aliased taps of rows
//                    the GEMM saw, plus the valid taps of the rows it skipped.
#include "a2v_gemm.h"

int g_a2v_conv_strided = 1;      // a2v_debug_set(21, v): strided dense convs on the 8-phase GEMMs (1) / the implicit-im2col kernels (0)

struct StridedGeom {
  int C, T_in, T_out, stride, pad, taps, nclips, r0, r1;
};

static bool strided_geom(const ConvA& a, int M, int K, int G, StridedGeom& s) {
  if (G != 1 || a.rowmask || a.rowidx || a.bdiv != 1 || a.goff != 0) return false;
  if (a.taps < 2 || a.stride < 2 || a.Cv != a.Cg || a.pitch != a.Cg) return false;
  if (a.T_in != a.T_out * a.stride || a.bstride != (long long)a.T_in * a.pitch) return false;
  if (M % a.T_out || K != a.taps * a.Cg || a.pad < 0 || a.pad >= a.taps) return false;
  s.C = a.Cg; s.T_in = a.T_in; s.T_out = a.T_out; s.stride = a.stride; s.pad = a.pad; s.taps = a.taps;
  s.nclips = M / a.T_out;
  s.r0 = (a.pad + a.stride - 1) / a.stride;                          // frames t with t*s - pad < 0
  const int first_bad = (a.T_in - a.taps + a.pad) / a.stride + 1;    // first t with t*s - pad + taps - 1 >= T_in
  s.r1 = first_bad < a.T_out ? a.T_out - first_bad : 0;
  return s.r0 + s.r1 >= 1 && s.r0 + s.r1 <= 8 && s.r0 + s.r1 < a.T_out && (long long)M - s.r0 - s.r1 >= 2048;
}

// one workgroup per (clip, edge frame): the zero-padded im2col row into LDS, then every output channel's dot product
__global__ __launch_bounds__(256) void conv_edge_nt_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W, bf16* __restrict__ Cout,
                                                           StridedGeom s, long long ldb, long long ldc, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* const xs = (bf16*)smem;
  const int ne = s.r0 + s.r1;
  const int e = blockIdx.x % ne, b = blockIdx.x / ne;
  const int t = e < s.r0 ? e : s.T_out - s.r1 + (e - s.r0);
  for (int c8 = threadIdx.x; c8 < K / 8; c8 += 256) {
    const int kk = c8 * 8, j = kk / s.C, ci = kk - j * s.C;
    const int f = t * s.stride - s.pad + j;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (f >= 0 && f < s.T_in) v = *(const uint4*)(X + ((long long)b * s.T_in + f) * s.C + ci);
    *(uint4*)(xs + kk) = v;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) {
    const bf16* w = W + (long long)n * ldb;
    float acc = 0.f;
    for (int kk = 0; kk < K; kk += 8) {
      const bf16x8 wv = *(const bf16x8*)(w + kk), xv = *(const bf16x8*)(xs + kk);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf((float)xv[i], (float)wv[i], acc);
    }
    Cout[((long long)b * s.T_out + t) * ldc + n] = (bf16)acc;
  }
}

// thread (n, kk): sum over the edge frames of dY[m][n] * (correct - used)[m][kk]; unique owner of W[n][kk] (plain update)
__global__ __launch_bounds__(256) void conv_edge_tn_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, float* __restrict__ Wg,
                                                           StridedGeom s, long long ldp, long long ldw, int K, int M) {
  const int kk = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (kk >= K) return;
  const int j = kk / s.C;
  const int ne = s.r0 + s.r1;
  float acc = 0.f;
  for (int b = 0; b < s.nclips; ++b)
    for (int e = 0; e < ne; ++e) {
      const int t = e < s.r0 ? e : s.T_out - s.r1 + (e - s.r0);
      const long long m = (long long)b * s.T_out + t;
      const int f = t * s.stride - s.pad + j;
      const bool in_range = f >= 0 && f < s.T_in;
      const bool in_gemm = m >= s.r0 && m < (long long)M - s.r1;
      if (in_range == in_gemm) continue;          // both: the GEMM used the right value; neither: zero either way
      const float x = (float)X[(m * s.stride - s.pad) * s.C + kk];      // flat view: the frame's own value or its alias
      const float d = (float)dY[m * ldp + n];
      acc = fmaf(in_range ? d : -d, x, acc);
    }
  if (acc != 0.f) Wg[(long long)n * ldw + kk] += acc;
}

bool a2v_launch_conv_strided_nt(const GemmArgs& g, hipStream_t st) {
  StridedGeom s;
  if (!g_a2v_conv_strided || g.flags != 0 || !strided_geom(g.a, g.M, g.K, g.G, s)) return false;
  if (g.K % 128 || g.K < 256 || g.N % 256 || g.K * 2 > 60000 || g.ldc % 8) return false;
  GemmArgs h = g;
  const long long ps = (long long)s.stride * g.a.pitch;
  h.a.ptr = (const bf16*)g.a.ptr + ((long long)s.r0 * s.stride - s.pad) * g.a.pitch;
  h.M = g.M - s.r0 - s.r1;
  h.a.pitch = ps; h.a.bstride = 0; h.a.T_in = h.M; h.a.T_out = h.M; h.a.stride = 1; h.a.pad = 0; h.a.taps = 1;
  h.a.Cg = g.K; h.a.Cv = g.K;
  h.C = (bf16*)g.C + (long long)s.r0 * g.ldc;
  if (!a2v_launch_nt_bf16_p8(h, st)) return false;
  hipLaunchKernelGGL(conv_edge_nt_kernel, dim3(s.nclips * (s.r0 + s.r1)), dim3(256), g.K * 2, st, (const bf16*)g.a.ptr,
                     (const bf16*)g.B, (bf16*)g.C, s, g.ldb, g.ldc, g.N, g.K);
  return true;
}

bool a2v_launch_conv_strided_tn(const WgradArgs& g, hipStream_t st) {
  StridedGeom s;
  if (!g_a2v_conv_strided || !strided_geom(g.a, g.M, g.K, g.G, s)) return false;
  WgradArgs h = g;
  h.a.ptr = (const bf16*)g.a.ptr + ((long long)s.r0 * s.stride - s.pad) * g.a.pitch;
  h.M = g.M - s.r0 - s.r1;
  h.a.pitch = (long long)s.stride * g.a.pitch; h.a.bstride = 0; h.a.T_in = h.M; h.a.T_out = h.M; h.a.stride = 1; h.a.pad = 0;
  h.a.taps = 1; h.a.Cg = g.K; h.a.Cv = g.K;
  h.P = (const bf16*)g.P + (long long)s.r0 * g.ldp;
  if (!a2v_launch_tn_bf16_p8(h, st)) return false;
  hipLaunchKernelGGL(conv_edge_tn_kernel, dim3(cdiv(g.K, 256), g.N), dim3(256), 0, st, (const bf16*)g.a.ptr, (const bf16*)g.P, g.W, s,
                     g.ldp, g.ldw, g.K, g.M);
  return true;
}
