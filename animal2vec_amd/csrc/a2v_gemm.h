// Shared definitions of the implicit-GEMM family (a2v_gemm.hip: general + fp32 kernels and the C ABI,
// a2v_gemm_fast.hip: the bf16 throughput kernels).
#pragma once
#include "a2v_common.h"

struct ConvA {
  const void* ptr;
  long long pitch;       // elements between consecutive source rows
  long long bstride;     // elements between batch items (T_in * pitch normally)
  int goff;              // channel offset per group
  int T_in, T_out;       // rows per batch item: source / GEMM-M
  int stride, pad, taps, Cg;   // Cg = tap width in K (packed weights), K = taps * Cg
  int Cv;                      // valid source channels per tap (<= Cg); channels [Cv, Cg) read as zero
  // clone-batch view (reference base.py:244,464): GEMM batch item b reads source item b / bdiv and
  // rows with rowmask[b*T_in + row] != 0 read as zero -- x.repeat_interleave(cb) * (1 - mask) is
  // never materialised.
  const unsigned char* rowmask;
  int bdiv;
  // row gather: GEMM row m reads the im2col row of flat index rowidx[m] (= b*T_out + t) instead of m itself -- the
  // output (NT) / dY (TN) stay compact.  Used where only the kept frames of a sequence are needed (the last pos-conv
  // layer, the first layer's input gradient): 140 of 2000 rows.
  const int* rowidx;
};

struct GemmArgs {
  ConvA a;
  const void* B; long long ldb; long long b_gstride;
  void* C; long long ldc; int c_goff;
  void* C2;                      // optional: pre-activation copy (same layout as C)
  const void* R; long long ldr;  // residual (RES) or pre-activation for GELU_BWD
  const float* bias;             // [G*N] fp32
  float* colsum;                 // [G*N] fp32, atomically accumulated column sums of the output
  int M, N, K, G;
  int flags;
  Drop drop;                     // F_DROPOUT: applied to (acc + bias [gelu]) before the residual add
  long long drop_base;           // added to the dropout element index m * ldc + col: a launch on a row range of a larger output
                                 // (the tail rows the persistent kernel hands to the 128 x 128 kernel) keeps the whole tensor's mask
};

enum { F_BIAS = 1, F_GELU = 2, F_RES = 4, F_GELU_BWD = 8, F_STORE_PRE = 16, F_COLSUM = 32, F_DROPOUT = 64, F_DACT = 128, F_X3IMG = 256,
       F_DACT8 = 512, F_QKNORM = 1024 };
// F_DACT8 (bf16, with F_DACT): the derivative tensor (C2 when stored, R when multiplied) is uint8, a2v_common.h dact8_*
// F_QKNORM (bf16, with F_BIAS only, the persistent kernel's fixed flavour or nothing): `colsum` is instead a [M / T_out][N / 32] fp32 table
// (zeroed by the caller) that receives, by atomic maximum, max over the T_out rows of each clip of the squared norm of every 32-column
// segment of the output -- what the long-sequence attention forward's ALiBi band limit needs of q and k (a2v_attn.hip), taken from the
// accumulators instead of a second pass over the qkv tensor.
// F_X3IMG (dtype A2V_BF16X3 on the persistent kernel only): C is written as the bf16 [hi | lo | hi] split image of the result (row pitch
// ldc >= 3 N bf16 elements) instead of fp32 -- for a result that only further products read (the Mlp's hidden activation)
// F_DACT (with F_GELU | F_STORE_PRE): C2 receives gelu'(acc + bias) instead of the pre-activation -- both come out of one
// evaluation of the erf parts -- and (with F_GELU_BWD) R is that derivative: the backward epilogue multiplies, no transcendental.

struct WgradArgs {
  ConvA a;                       // implicit A (M x K per group)
  const void* P; long long ldp; int p_goff;   // dY: P[m][g*p_goff + n]
  float* W; long long ldw; long long w_gstride;  // fp32 [G][N][ldw], accumulated with atomics
  int M, N, K, G;
  int m_chunk;                   // rows of M handled by one split (multiple of the M step)
  const int* skip_if = nullptr;  // gemm_tn_bf16_fast_kernel: return at once if *skip_if != 0 (the fallback behind conv_wgrad_sparse_kernel)
};

// ----------------------------------------------------------------------------------------------
// Per-thread description of one 16-byte chunk this thread stages from the implicit A operand.
struct ARow {
  const char* base;   // points at X[b / bdiv][0][g*goff] (bytes)
  const unsigned char* mrow;   // rowmask + b*T_in, or nullptr
  int t0;             // t*stride - pad
  bool valid;
};

template <typename T>
__device__ __forceinline__ ARow make_arow(const ConvA& a, int m, int M, int g) {
  ARow r;
  r.valid = m < M;
  int mm = r.valid ? m : 0;
  if (a.rowidx) mm = a.rowidx[mm];
  const int b = mm / a.T_out, t = mm - b * a.T_out;
  r.base = (const char*)a.ptr + ((long long)(b / a.bdiv) * a.bstride + (long long)g * a.goff) * (long long)sizeof(T);
  r.mrow = a.rowmask ? a.rowmask + (long long)b * a.T_in : nullptr;
  r.t0 = t * a.stride - a.pad;
  return r;
}

// load the 16-B chunk holding A[m][kk .. kk+16/sizeof(T)) or zeros
template <typename T>
__device__ __forceinline__ uint4 load_a_chunk(const ConvA& a, const ARow& r, int kk, int K) {
  uint4 z = {0u, 0u, 0u, 0u};
  if (!r.valid || kk >= K) return z;
  const int j = kk / a.Cg, ci = kk - j * a.Cg;
  const int row = r.t0 + j;
  if (row < 0 || row >= a.T_in || ci >= a.Cv) return z;
  if (r.mrow && r.mrow[row]) return z;
  return *(const uint4*)(r.base + ((long long)row * a.pitch + ci) * (long long)sizeof(T));
}

// ----------------------------------------------------------------------------------------------
// Shared epilogue: one wave owns a 64x64 fp32 sub-tile staged in LDS (pitch 68 floats).
template <typename T>
__device__ __forceinline__ void wave_epilogue(const GemmArgs& g, float* stage, int lane, int m_base, int n_base,
                                              int grp, int jump32 = 0, int m_limit = -1) {
  // stage[r][c] r,c in [0,64); jump32: stage columns 32..63 map to output columns n_base + jump32 + 32.. (the
  // 256x256 kernel's wave owns two 32-column strips 128 columns apart)
  const int cq = (lane & 15) * 4;
  const int n = n_base + cq + (cq >= 32 ? jump32 : 0);                 // column inside the group
  const bool ncol_ok = n < g.N;              // N % 4 == 0 is required
  const long long ccol = (long long)grp * g.c_goff + n;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if ((g.flags & F_BIAS) && ncol_ok) bias4 = *(const f32x4*)(g.bias + (long long)grp * g.N + n);
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  T* C = (T*)g.C;
  T* C2 = (T*)g.C2;
  const T* R = (const T*)g.R;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int r = it * 4 + (lane >> 4);
    const int m = m_base + r;
    f32x4 v = *(const f32x4*)(stage + r * 68 + cq);
    if (m < (m_limit < 0 ? g.M : m_limit) && ncol_ok) {
      v += bias4;
      if ((g.flags & (F_DACT | F_GELU | F_STORE_PRE)) == (F_DACT | F_GELU | F_STORE_PRE)) {
        f32x4 gr;
#pragma unroll
        for (int i = 0; i < 4; ++i) { float d_; v[i] = gelu_with_grad(v[i], d_); gr[i] = d_; }
        if (g.flags & F_DACT8) *(unsigned*)((unsigned char*)g.C2 + (long long)m * g.ldc + ccol) = dact8_pack4(gr[0], gr[1], gr[2], gr[3]);
        else Vec4<T>::store(C2 + (long long)m * g.ldc + ccol, gr);
      } else {
        if (g.flags & F_STORE_PRE) Vec4<T>::store(C2 + (long long)m * g.ldc + ccol, v);
        if (g.flags & F_GELU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = gelu_f(v[i]);
        }
      }
      if (g.flags & F_DROPOUT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = drop_apply(g.drop, (unsigned long long)m * g.ldc + ccol + i + (unsigned long long)g.drop_base, v[i]);
      }
      if (g.flags & F_RES) v += Vec4<T>::load(R + (long long)m * g.ldr + ccol);
      if (g.flags & F_GELU_BWD) {
        f32x4 h;
        if (g.flags & F_DACT8) {
          float h4[4];
          dact8_unpack4(*(const unsigned*)((const unsigned char*)g.R + (long long)m * g.ldr + ccol), h4);
          h = (f32x4){h4[0], h4[1], h4[2], h4[3]};
        } else {
          h = Vec4<T>::load(R + (long long)m * g.ldr + ccol);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= (g.flags & F_DACT) ? h[i] : gelu_grad_f(h[i]);
      }
      Vec4<T>::store(C + (long long)m * g.ldc + ccol, v);
      if (g.flags & F_COLSUM) {
        // sum what was actually written (rounded to T), so bias grads match the stored gradient
#pragma unroll
        for (int i = 0; i < 4; ++i) cs[i] += Cvt<T>::to(Cvt<T>::from(v[i]));
      }
    }
  }
  if (g.flags & F_COLSUM) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = cs[i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      cs[i] = s;
    }
    if (lane < 16 && ncol_ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(g.colsum + (long long)grp * g.N + n + i, cs[i]);
    }
  }
}


// byte offset of 16-B chunk `c` (0..7) of row `r` inside a [rows][64] bf16 tile (128-B rows)
__device__ __forceinline__ int swz128(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// TN tiles: [64 rows m][128 cols] bf16 (256-B rows); 32-B column blocks XOR-swizzled by row so the
// ds_read_b64_tr_b16 lane groups (rows 8g+q, q=0..3) land on distinct banks.
__device__ __forceinline__ int tn_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ int tn_off(int r, int col) {   // byte offset of element (r, col)
  return r * 256 + ((((col >> 4) ^ tn_f(r)) & 7) << 5) + ((col & 15) << 1);
}

#define NT_THREADS 256
#define NT_LDS_BYTES 69632   // 4 waves * 64*68*4 B epilogue staging (>= any pipeline stage pair used)

// launchers implemented in a2v_gemm_fast.hip; return false when the shape is not eligible
bool a2v_launch_nt_bf16_fast(const GemmArgs& g, hipStream_t stream);
bool a2v_launch_nt_bf16_p8(const GemmArgs& g, hipStream_t stream);
bool a2v_launch_nt_bf16_w2(const GemmArgs& g, hipStream_t stream);   // a2v_gemm_w2.hip: 256 x 128 tiles, two workgroups per CU
bool a2v_launch_conv_slab3(const GemmArgs& g, long long a_lo, long long b_lo, hipStream_t stream);   // A2V_BF16X3S
bool a2v_launch_nt_x3_p8(const GemmArgs& g, hipStream_t stream);      // bf16 split operands, fp32 C / C2 / R (A2V_BF16X3)
bool a2v_launch_nt_x3_fast(const GemmArgs& g, hipStream_t stream);    // the same for convolutions / shapes the 8-phase kernel refuses
bool a2v_launch_conv_slab_o32(const GemmArgs& g, hipStream_t stream); // grouped convs with 64-wide taps: slab kernel, fp32 outputs
bool a2v_launch_tn_bf16_p8(const WgradArgs& g, hipStream_t stream);   // a2v_gemm_tn8.hip: 256x256 weight gradients of the Linears
bool a2v_launch_tn_bf16_p8_pair(const WgradArgs* ga, const WgradArgs* gb, hipStream_t stream);   // two of them in one launch
bool a2v_launch_wgrad_slab(const WgradArgs& g, hipStream_t stream);
int a2v_launch_wgrad_sparse(const WgradArgs& g, int n_seq, hipStream_t stream, const int** ok_flag);   // a2v_wgrad_sparse.hip
bool a2v_launch_conv_dec48(const GemmArgs& g, hipStream_t stream);     // a2v_conv_dec.hip: decoder grouped conv (48 channels per group, HBM-bound)
bool a2v_launch_conv_slab(const GemmArgs& g, hipStream_t stream);      // a2v_conv_slab.hip: grouped-conv forward / input grads    // a2v_wgrad_slab.hip: grouped-conv weight grads     // a2v_gemm_p8.hip: 256x256 8-phase kernel
bool a2v_launch_tn_bf16_fast(const WgradArgs& g, int splits, hipStream_t stream);
bool a2v_launch_conv_strided_nt(const GemmArgs& g, hipStream_t stream);   // a2v_conv_strided.hip: dense strided convs on the 8-phase GEMMs
bool a2v_launch_conv_strided_tn(const WgradArgs& g, hipStream_t stream);
int a2v_tn_bf16_tiles(int N, int K);
