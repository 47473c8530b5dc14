/* animal2vec MI355X hot path -- C ABI of liba2v_hip.so (gfx950).
 *
 * Drop-in boundary.  The reference (livingingroups/animal2vec) has no native code at all: every
 * device op of its pretraining path is a stock torch op issued from Python.  Each entry point below
 * therefore replaces a *Python call site* of the reference (cited as file:line, paths relative to the
 * reference root) rather than an existing FFI symbol.  The reference-side binding a maintainer would
 * add is the ctypes stub shown in INTEGRATION.md (animal2vec_amd/_lib.py is that stub).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers.
 *  - dtype: 0 = fp32 storage + v_mfma_f32_16x16x4_f32 (parity mode), 1 = bf16 storage +
 *    v_mfma_f32_16x16x32_bf16 with fp32 accumulation (throughput mode).  Parameters, biases,
 *    statistics and gradients of parameters are always fp32.
 *  - activations are channels-last: (batch, time, channels) row-major.
 *  - ownership: the caller owns every buffer (incl. workspaces and tensors saved for backward);
 *    kernels never allocate, free or retain pointers.
 *  - every call is asynchronous on `stream` and graph-capturable (no sync, no alloc) and re-entrant across streams:
 *    the device scratch the kernels need besides their arguments (tile counters of the persistent GEMM, column-sum partial
 *    rows, d alibi_scale partial sums, |q.k| bounds) is ONE workspace per stream.  ABI 10: the caller owns it --
 *    a2v_query_workspace() bytes of zeroed device memory handed over with a2v_bind_workspace(stream, ptr, bytes), any
 *    number of streams, the library keeps no state of its own for them (caller-owned scratch).  A stream that
 *    was never bound falls back to one of 4 library-owned static sets (first use binds one, a2v_stream_release returns it;
 *    a fifth unbound stream still works, through the kernels that need no scratch) -- the ABI <= 9 behaviour, kept so that
 *    existing hosts run unchanged.  Calls on ONE stream are ordered by the stream as usual.
 *  - return value: 0 on success, <0 on error (-1 bad argument, -2 bad dtype, -(1000+hipError_t) on a
 *    launch failure).  No exceptions cross the ABI.
 */
#ifndef A2V_H
#define A2V_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* a2v_stream_t; /* == hipStream_t */

#define A2V_F32 0
#define A2V_BF16 1
#define A2V_BF16X3 2 /* a2v_conv_gemm only: A and Bw are bf16 SPLIT images (a2v_split_bf16) with K = 3 x the real reduction length,
                      * C / C2 / R are fp32 -- the bf16x3 parity mode: fp32-grade results on the bf16 matrix cores */
#define A2V_BF16X3S 3 /* a2v_conv_gemm only (ABI 9): grouped stride-1 convolutions with 64-wide taps and 64 outputs per group
                       * (taps 2..24, T_out >= 256, flags within BIAS | RES).  A and Bw are [hi ; lo] STACKED images
                       * (a2v_split_bf16 mode 2): the lo image of A starts ceil(M / T_out / bdiv) * a_bstride elements after A,
                       * the lo pack of Bw starts G * b_gstride (G > 1; else N * ldb) elements after Bw; Cg / K / pitches are
                       * the bf16 convolution's.  One pass forms X_hi W_hi + X_lo W_hi + X_hi W_lo; C / R are fp32.
                       * Anything else: A2V_ERR_UNSUPPORTED (run the three products as three A2V_BF16X3 calls with R = C). */

/* epilogue flags of a2v_conv_gemm */
#define A2V_EPI_BIAS 1      /* + bias[g*N+n] */
#define A2V_EPI_GELU 2      /* exact erf GELU (torch.nn.GELU) */
#define A2V_EPI_RES 4       /* + R[m][.] (residual / gradient accumulation) */
#define A2V_EPI_GELU_BWD 8  /* * gelu'(R[m][.]) : R holds the saved pre-activation */
#define A2V_EPI_STORE_PRE 16 /* also store the pre-activation (acc + bias) to C2 */
#define A2V_EPI_COLSUM 32   /* atomically add column sums of the output to colsum (bias gradient) */
#define A2V_EPI_DROPOUT 64  /* counter-based dropout(drop_p, drop_seed) of (acc+bias[gelu]) before the residual */
#define A2V_EPI_X3IMG 256   /* dtype A2V_BF16X3 only (ABI 9), plain row-major products the persistent kernel takes, no RES / DROPOUT (GELU_BWD since round 4):
                             * C is written as the bf16 [hi | lo | hi] split image of the result (row pitch ldc >= 3 N bf16 elements; a
                             * pre-activation / derivative tensor C2 stays fp32 with row pitch N) instead of fp32 -- for a result that only further A2V_BF16X3 products read; else A2V_ERR_UNSUPPORTED */
#define A2V_EPI_DACT 128    /* modifier.  With GELU | STORE_PRE: C2 receives gelu'(acc + bias) instead of the pre-activation (one
                             * evaluation of the erf parts gives both); with GELU_BWD: R holds that derivative and is multiplied in
                             * as it is.  The Mlp's fc1 forward / fc2 input gradient pair (timm Mlp, modules.py:328-337) uses it:
                             * the backward epilogue then has no transcendental. */

#define A2V_EPI_DACT8 512   /* modifier of A2V_EPI_DACT (ABI 11), dtype A2V_BF16, plain Linear products (G = 1, taps = 1, no row gather / mask): the
                             * derivative tensor -- C2 with GELU | STORE_PRE, R with GELU_BWD -- holds ONE BYTE per element (row pitch ldc / ldr
                             * in elements = bytes, % 8): q = round(200 gelu') + 26, gelu' ~ (q - 26) / 200, a grid of 1/200 on which 0 and 1
                             * are exact (|error| <= 0.0025 -- what bf16 rounding does to a derivative near 1/2).  A quarter of the fc1 launch's
                             * output bytes and a third of the fc2 input-gradient launch's extra input bytes less; the bf16 throughput mode's
                             * default (A2V_DACT8=0 keeps bf16), never used by the fp32 / bf16x3 parity modes.  Other uses: A2V_ERR_UNSUPPORTED */

#define A2V_EPI_QKNORM 1024 /* (ABI 14) dtype A2V_BF16, flags == BIAS | QKNORM, plain Linear products over whole clips (G = 1, taps = 1, T_in = T_out = the
                             * clip's rows >= 256, M % T_out == 0, a_bstride = T_out * a_pitch, N % 256 == 0): `colsum` is a caller-zeroed
                             * (M / T_out, N / 32) fp32 table that receives, per clip and 32-column segment of the output, the maximum over the
                             * clip's rows of the segment's squared norm -- for a qkv product, what a2v_attention_fwd(qk_norms) needs.  Only the
                             * persistent 256-row kernel forms it: any call it does not take returns A2V_ERR_UNSUPPORTED and has done nothing
                             * (repeat it without the flag and pass qk_norms = NULL). */

/* ---------------------------------------------------------------- implicit-GEMM (MFMA) family
 * C[m][g*c_goff + n] = epi( sum_kk A_g[m][kk] * Bw[g][n][kk] ),  kk = tap*Cg + ci,
 * A_g[m=(b,t)][kk] = X[b / bdiv][t*stride - pad + tap][g*a_goff + ci]   (0 outside [0,T_in), 0 for
 * ci >= Cv (tap padding of the packed weights), and 0 where rowmask[b*T_in + row] != 0).
 * Replaces: F.linear in AltAttention/Mlp/project_features/decoder.proj (nn/modalities/modules.py:371,
 * 408; timm Mlp; nn/modalities/audio.py:87; modules.py:191), nn.Conv1d of the feature extractor
 * (nn/utils.py:1085-1090), the grouped positional/decoder convs (audio.py:97-103, modules.py:143-149)
 * and their autograd input gradients; with rowmask/bdiv also x.repeat_interleave(cb)*(1-mask)
 * (nn/modalities/base.py:244,464). */
int a2v_conv_gemm(int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in, int T_out,
                  int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv, const void* Bw,
                  long long ldb, long long b_gstride, void* C, long long ldc, int c_goff, void* C2, const void* R,
                  long long ldr, const float* bias, float* colsum, int M, int N, int K, int G, int flags, float drop_p,
                  unsigned drop_seed, a2v_stream_t stream);

/* W[g][n][kk] += sum_m P[m][g*p_goff + n] * A_g[m][kk]   (fp32 atomics; zero W first)
 * Replaces autograd's weight gradients of the ops above. */
int a2v_conv_wgrad(int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in, int T_out,
                   int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv, const void* P,
                   long long ldp, int p_goff, float* W, long long ldw, long long w_gstride, int M, int N, int K, int G,
                   a2v_stream_t stream);
/* Row-gathered forms: GEMM row m reads the im2col row of flat index rowidx[m] (= b*T_out + t, int32 on the device)
 * instead of row m; C (a2v_conv_gemm_rows) / P = dY (a2v_conv_wgrad_rows) stay compact with M = number of gathered
 * rows.  The reference computes the whole (N, T, D) positional-encoder output and then gathers the kept frames
 * (base.py:300-314); the last layer only needs those rows, and the first layer's input gradient only matters at
 * them, so the engine evaluates 140 of 2000 rows there.  rowidx = NULL is the dense form. */
int a2v_conv_gemm_rows(const int* rowidx, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                       int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                       const void* Bw, long long ldb, long long b_gstride, void* C, long long ldc, int c_goff, void* C2,
                       const void* R, long long ldr, const float* bias, float* colsum, int M, int N, int K, int G, int flags,
                       float drop_p, unsigned drop_seed, a2v_stream_t stream);
int a2v_conv_wgrad_rows(const int* rowidx, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                        int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                        const void* P, long long ldp, int p_goff, float* W, long long ldw, long long w_gstride, int M, int N,
                        int K, int G, a2v_stream_t stream);
/* Grouped convolution of a ROW-SPARSE input with a dense output (ABI 12; bf16, 64 channels per group in and out, stride 1,
 * taps = 19: the positional encoder, audio.py:90-113):
 *   Y[c][m][g*c_goff + n] = bias[g*c_goff + n] + sum_tap sum_ci W[g][n][tap*64 + ci] * X_c[m + tap - pad][g*x_goff + ci]
 * for n_seq sequences of T frames, where X_c is zero outside a few rows given EITHER by rowmask (n_seq x T bytes; X_c = sequence
 * c / bdiv of X (rows x_pitch apart, sequences x_bstride apart) at the frames whose mask byte is 0: the first layer's forward on
 * clone * (1 - mask), base.py:292-314) OR by rowidx (n_seq x rows_per_seq flat indices c*T + t, any order; X compact: row i of X is
 * frame rowidx[i]: the last layer's input gradient, whose output gradient only exists at the kept frames -- W is then the
 * transposed / flipped pack and pad = taps - 1 - pad_forward).  Same result as a2v_conv_gemm on the dense (zero-filled) input up to
 * the order of the fp32 sums; the products are formed per kept row and scatter-added into an fp32 tile of the output window
 * (a2v_conv_sparse.hip).  W: (G, 64, taps*64) bf16, rows ldb apart, groups b_gstride apart; Y: (n_seq, T, ldc). */
int a2v_conv_sparse_in(int dtype, const void* X, long long x_pitch, long long x_bstride, int x_goff, const unsigned char* rowmask,
                       int bdiv, const int* rowidx, int rows_per_seq, int n_seq, int T, int pad, int taps, const void* W,
                       long long ldb, long long b_gstride, const float* bias, void* Y, long long ldc, int c_goff, int G,
                       a2v_stream_t stream);
/* The kept-frame form of a2v_conv_wgrad_rows (ABI 12): rowidx holds the rows of n_seq sequences, M / n_seq each, strictly
 * ascending -- what the reference's x[ids_keep] gather produces when the kept frames stay in frame order (base.py:300-314;
 * animal2vec_amd/masking.py resolves the reference's unstable argsort that way).  Same result as a2v_conv_wgrad_rows; with 64
 * channels per group, stride 1 and bf16 the product runs over the kept rows against time-contiguous slabs of the dense operand
 * (a2v_wgrad_sparse.hip) instead of re-gathering a taps-row window per kept row.  The order is checked on the device: a list that
 * is not ascending takes the a2v_conv_wgrad_rows kernel.  (a2v_conv_wgrad with a row mask -- the first layer, whose INPUT is zero
 * at the masked frames -- takes the same kernel with the operands' roles exchanged; no new entry is needed there.) */
int a2v_conv_wgrad_kept(const int* rowidx, int n_seq, int dtype, const void* A, long long a_pitch, long long a_bstride, int a_goff, int T_in,
                        int T_out, int stride, int pad, int taps, int Cg, int Cv, const unsigned char* rowmask, int bdiv,
                        const void* P, long long ldp, int p_goff, float* W, long long ldw, long long w_gstride, int M, int N,
                        int K, int G, a2v_stream_t stream);

/* Two Linear weight gradients over the same M rows in one launch (gw_a[Na,Ka] += dy_a^T x_a, gw_b[Nb,Kb] += dy_b^T x_b;
 * torch.autograd's two separate addmm in the reference).  Sharing the launch lets the 256x256 TN kernel cut M into
 * fewer chunks, i.e. fewer fp32 atomic adds per gradient element; shapes it does not take run as two launches. */
int a2v_linear_wgrad_pair(int dtype, int M, const void* xa, long long lda, const void* dya, long long ldpa, float* gwa,
                          long long ldwa, int Na, int Ka, const void* xb, long long ldb, const void* dyb, long long ldpb,
                          float* gwb, long long ldwb, int Nb, int Kb, a2v_stream_t stream);

/* ---------------------------------------------------------------- attention with analytic ALiBi
 * qkv (N,S,3,H,HD); out (N,S,H*HD); lse (N,H,S) fp32; pos (N,S) int32 original frame index or NULL
 * (= arange); slopes (H) fp32; alibi_scale (H) fp32 (clamped at 0 inside).
 * Replaces AltAttention.forward's q@k^T + alibi + softmax + @v (modules.py:391-407) together with
 * get_alibi_bias / masked_alibi (base.py:292-314,553-698): no (N,H,S,S) tensor is ever built. */
/* mdist (N,H,S) fp32 or NULL (ABI 8): c_i = sum_j P_ij |pos_i - pos_j|, the mean attended distance of row i.  Calls that keep
 * lse for a2v_attention_bwd pass it too: the backward centres the distance on it when it forms d alibi_scale, which makes
 * that gradient insensitive to the rounding of the saved output (sum_j dS_ij = 0 holds only up to dO.(O_rounded - O); see
 * AttnBwdArgs::mdist in csrc/a2v_attn.hip).  NULL in both calls is the uncentred ABI-7 arithmetic. */
/* dropmask (ABI 13): where a2v_attention_dropmask_words(dtype, S, HD) is W > 0 (bf16, HD = 64, S <= 160: the student's kept
 * frames) the forward makes the dropout decision of the softmax probabilities (modules.py:405) ONCE and writes it as a bit mask of
 * W u32 words per (n, h) -- N*H*W words, caller-owned like every buffer -- which a2v_attention_bwd reads instead of re-hashing
 * every score (layout: csrc/a2v_attn_seq2.h).  NULL in either call, or W = 0: both calls regenerate the same decisions from the
 * counter hash of (drop_seed, row, key), so a mask-less backward after a masked forward (and vice versa) is still exact. */
/* qk_norms (ABI 14) or NULL: the (N, 3*H*HD/32) fp32 table the qkv product left through A2V_EPI_QKNORM (HD = 64, bf16, pos = NULL, S >= 512:
 * the EMA teacher's and the finetune path's full-length sequences).  The forward's ALiBi band limit -- key tiles whose softmax weight is
 * provably below e^-20 of the row sum are not visited -- needs max_i |q_i|^2 and max_j |k_j|^2 of every (n, h); with the table they come
 * from the product's accumulators (Cauchy-Schwarz per 32-column segment, two segments per head row: a bound as before), without it
 * from a pass over q and k.  Ignored where no band limit applies. */
int a2v_attention_dropmask_words(int dtype, int S, int HD);
int a2v_attention_fwd(int dtype, const void* qkv, void* out, float* lse, float* mdist, unsigned* dropmask, const float* qk_norms,
                      const int* pos, const float* slopes, const float* alibi_scale, int N, int S, int H, int HD, float scale,
                      float drop_p, unsigned drop_seed, a2v_stream_t stream);
/* delta (N,H,S) fp32 scratch; dqkv receives dK,dV; dalibi (H) fp32 accumulates d loss / d alibi_scale.
 * dQ: when a2v_attention_bwd_direct_dq(dtype, S, HD) is 1 (bf16 with S <= 192 -- the student's kept-token sequences --
 * or bf16 with HD = 64 at any S -- the two-pass long-sequence kernels of the finetune path)
 * the kernel owns a whole (n,h) sequence and writes dQ into the q slot of dqkv itself (dq32 may be NULL);
 * otherwise dq32 (N,S,H,HD) fp32 must be zero on entry, receives dQ through atomics, and the caller casts it
 * into dqkv (a2v_cast2d). */
int a2v_attention_bwd(int dtype, const void* qkv, const void* out, const void* d_out, const float* lse, const float* mdist,
                      const unsigned* dropmask, float* delta, void* dqkv, float* dq32, float* dalibi, float* qkv_bias_grad, const int* pos, const float* slopes,
                      const float* alibi_scale, int N, int S, int H, int HD, float scale, float drop_p, unsigned drop_seed,
                      a2v_stream_t stream);
/* qkv_bias_grad (3*H*HD fp32, accumulated) or NULL: the column sums of dqkv -- the qkv Linear's bias gradient -- formed
 * in the kernel instead of by a separate pass over dqkv; allowed only where a2v_attention_bwd_fused_bias() is 1 (the
 * whole-sequence kernel: bf16, HD = 64, S <= 192). */
int a2v_attention_bwd_fused_bias(int dtype, int S, int HD);
int a2v_attention_bwd_direct_dq(int dtype, int S, int HD);

/* ---------------------------------------------------------------- normalisation
 * y = act( LN(x [+ res]) * gamma + beta ), act 0 none / 1 GELU; optional sum_out = x + res;
 * optional mean/rstd (M) fp32 saved for backward.
 * Replaces Fp32LayerNorm / nn.LayerNorm call sites: nn/utils.py:1105-1115, audio.py:86,106,
 * modules.py:85,152,330,333. */
int a2v_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                      void* sum_out, float* mean, float* rstd, const void* post, long long M, int C, long long ldx,
                      long long ldy, float eps, int act, float drop_p, unsigned drop_seed, a2v_stream_t stream);
/* dx [+ dres]; dgamma/dbeta (C) and dxsum (C, column sums of dx) are accumulated with atomics.
 * drop_mode 1: the forward applied dropout after this LN -> dy is masked first; drop_mode 2: additionally
 * writes dx_drop = dropout(dx) (gradient of a dropped branch added to this LN's input) and dxsum sums it. */
int a2v_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, void* dx, const void* dres, float* dgamma, float* dbeta,
                      float* dxsum, long long M, int C, long long ldx, long long lddy, int act, float drop_p,
                      unsigned drop_seed, int drop_mode, void* dx_drop, a2v_stream_t stream);
/* y (B,T,D) fp32 = mean_l instance_norm_T(layers[l])   -- make_targets, nn/data2vec2.py:1023-1066.
 * layers: HOST array of L device pointers (B,T,D); mean_ws/rstd_ws (L*B*D) fp32 scratch. */
int a2v_instnorm_targets(int dtype, const void* const* layers, int L, float* mean_ws, float* rstd_ws, float* y, int B,
                         int T, int D, float eps, a2v_stream_t stream);
int a2v_colsum(int dtype, const void* x, float* out, long long M, int C, long long ld, a2v_stream_t stream);

/* ---------------------------------------------------------------- SincConv front end (nn/sinc.py)
 * filters (Cp,k) fp32 from (low_hz_, band_hz_) -- SincConv._get_sinc_filters, sinc.py:181-223;
 * n_ / window are the module buffers of sinc.py:264-276. */
int a2v_sinc_filters_fwd(const float* low_hz, const float* band_hz, const float* n_, const float* window, float* filt,
                         int C, int Cp, int k, float min_low_hz, float min_band_hz, float sample_rate,
                         a2v_stream_t stream);
int a2v_sinc_filters_bwd(const float* low_hz, const float* band_hz, const float* n_, const float* window,
                         const float* dfilt, float* dlow /* += */, float* dband /* += */, int C, int k, float min_low_hz,
                         float min_band_hz, float sample_rate, a2v_stream_t stream);
/* x (B,L) fp32 -> y (B,L,128) channels-last: reflect pad + conv1d + LayerNorm(C) + PSwish in one
 * pass -- SincConv.forward (sinc.py:107-159) + ConvFeatureExtractionModel block 0 tail
 * (nn/utils.py:1105-1117) + PSwish.forward (utils.py:1430-1431). mean/rstd (B,L) saved for backward.
 * `filt` must be what a2v_sinc_filters_fwd produces: band-pass rows are symmetric (filt[c][j] == filt[c][k-1-j],
 * sinc.py:204-218) and the single-pass kernels (C <= 128, k <= 63) fold the mirrored taps (half the fp32 MFMA work). */
int a2v_sinc_fused_fwd(int dtype, const float* x, const float* filt, const float* ln_w, const float* ln_b,
                       const float* ps_alpha, const float* ps_beta, void* y, float* mean, float* rstd, int B, int L,
                       int C, int k, float eps, a2v_stream_t stream);
/* dfilt (C,k) and dpar (4, pitch: d ln_w, d ln_b, d ps_alpha, d ps_beta) are accumulated: zero first.  dfilt is the
 * gradient with respect to the SYMMETRIC filter: what is defined is dfilt[c][j] + dfilt[c][k-1-j] for j < k/2 (and the centre
 * tap) -- exactly what a2v_sinc_filters_bwd consumes; the single-pass kernels put the whole sum into entry j.  pitch =
 * a2v_sinc_out_channels(C, k); when it is 320 (blocked kernels) `scratch` must hold B*L*320 floats (the conv-output
 * gradient, formed once and then reduced into dfilt by a second kernel); it may be NULL for the 128-pitch specs. */
int a2v_sinc_fused_bwd(int dtype, const float* x, const float* filt, const float* ln_w, const float* ln_b,
                       const float* ps_alpha, const float* ps_beta, const void* dy, const float* mean,
                       const float* rstd, float* dfilt, float* dpar, float* scratch, int B, int L, int C, int k,
                       a2v_stream_t stream);
/* Channel pitch of y / dy and row length of dpar for a (filters C, taps k) front-end spec: 128 for C <= 128, k <= 63
 * (the shipped 8 kHz spec 127 x 63), 320 for the blocked kernels (C <= 320, k <= 255: e.g. 296 x 149 at 44.1 kHz),
 * -1 if unsupported. */
int a2v_sinc_out_channels(int C, int k);

/* ---------------------------------------------------------------- masking / token plumbing
 * out[n,k,:] = a[n/cb, ids[n,k], :] (+ b[n, ids[n,k], :])  -- make_maskinfo gather + x_unmasked +
 * gather_unmasked(x_pos) (base.py:244,447,277-280). */
int a2v_gather_tokens(int dtype, const void* a, const void* b, const int* ids, void* out, int N, int K, int D, int Ta,
                      int cb, float drop_p, unsigned drop_seed, a2v_stream_t stream);
int a2v_scatter_tokens(int dtype, const void* dtok, const int* restore, void* dpos, int N, int K, int D, int T,
                       a2v_stream_t stream);
int a2v_clone_reduce(int dtype, const void* dxm, const void* dtok, const int* restore, void* dfeat, int B, int cb,
                     int K, int D, int T, a2v_stream_t stream);
/* decoder_input (base.py:162-192): out[n,t] = restore<K ? x[n,restore] : noise[n,restore-K] */
int a2v_decoder_input(int dtype, const void* x, const float* noise, const int* restore, void* out, int N, int K, int D,
                      int T, float drop_p, unsigned drop_seed, a2v_stream_t stream);
/* the same with the noise rows drawn inside the kernel: N(0, noise_std) per masked (n, t, channel), a counter hash of
 * (noise_seed, output element) -- the distribution base.py:172-176 draws with normal_(), not torch's Philox stream --
 * so that no (N, T-K, D) fp32 tensor is written and read back. */
int a2v_decoder_input_rng(int dtype, const void* x, const int* restore, void* out, int N, int K, int D, int T,
                          float noise_std, unsigned noise_seed, float drop_p, unsigned drop_seed, a2v_stream_t stream);

/* ---------------------------------------------------------------- loss (data2vec2.py:850-864,904-909,
 * 1005-1021) + compute_var sums (:1095-1110).  stats: double[4*D+2] accumulated (zero first):
 * sum x, sum x^2, sum y, sum y^2, loss sum, masked-row count. dx may be NULL; loss_rows (N,T,D)
 * fp32 unreduced loss (0 on kept rows) may be NULL. */
int a2v_d2v_loss(int dtype, const void* x, const float* y, const unsigned char* mask, void* dx, float* loss_rows,
                 double* stats, int N, int T, int D, int cb, float scale, float gscale, a2v_stream_t stream);

/* ---------------------------------------------------------------- BC-learning source mixup (train only, no grad)
 * compute_gain_torch (nn/data2vec2.py:453-498): gain_db (B, nframes) of A-weighted framed power spectra;
 * mixup block (:536-598, same_mixup, mixup_prob 1): out[b] = (p x[b] + (1-p) x[perm[b]]) / sqrt(p^2+(1-p)^2). */
int a2v_mixup_gain(const float* x, const float* window, const float* aweight, float* gain_db, int B, int L, int n_fft,
                   int hop, float floor_, a2v_stream_t stream);
/* mix_mask (B) uint8 or NULL: with mixup_prob < 1 only the clips whose byte is set are mixed (with x[perm[b]]), the
 * others are copied through (data2vec2.py:540-547,584-589). */
int a2v_mixup_apply(const float* x, const int* perm, const float* gain_db, float* out, int B, int L, int nframes,
                    float r, const unsigned char* mix_mask, a2v_stream_t stream);

/* ---------------------------------------------------------------- optimizer side
 * EMA teacher: fairseq EMAModule.step driven from data2vec2.py:386-410, flat fp32 buffers. */
int a2v_ema_update(float* ema, const float* p, long long n, float decay, a2v_stream_t stream);
/* fairseq Adam (configs/MeerKAT/a2v_large_pretrain_best.yaml:64-79). */
/* decay_mask: one byte per 64 parameters (0 = no weight decay: the reference's optim_overrides weight_decay_scale 0,
 * data2vec2.py:318-322) or NULL; lr_scale: one float per 64 parameters (per-group lr_scale of the composite optimizer,
 * e.g. layer_decay, wav2vec2.py:214-236) or NULL.  Any contiguous slice may be stepped on its own (own step count). */
int a2v_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float grad_scale, const unsigned char* decay_mask,
                  const float* grad_scale_dev, const float* lr_scale, a2v_stream_t stream);
int a2v_sumsq(const float* x, long long n, double* out, a2v_stream_t stream);

/* ---------------------------------------------------------------- layout helpers
 * w: reference parameter layout (G*N, Cg, k) fp32.  mode 0: [g][n][tap*Cgp+ci]; mode 1: flipped /
 * transposed [g][ci][tap*Np+n] (input-gradient-as-convolution operand, and W^T for Linear);
 * mode 2 (G==1): [tap*Cgp+ci][n]. */
int a2v_pack_weight(int dtype, const float* w, void* out, int G, int N, int Cg, int k, int Cgp, int Np, int mode,
                    a2v_stream_t stream);
/* Batched form of a2v_pack_weight: desc_dev = n x 10 int64 on the device
 * (src fp32 ptr, dst ptr, G, N, Cg, k, Cgp, Np, mode, total elements of dst); one launch rebuilds every operand
 * image of a parameter set after the optimizer / EMA step (fairseq has no counterpart: its modules read the fp32
 * parameters directly). */
int a2v_pack_weight_batch(int dtype, const long long* desc_dev, int n, a2v_stream_t stream);
/* gw[(g*N+n)][ci][j] += packed[g][n][j*Cgp + ci], then the entries read are cleared: the packed fp32 accumulator of
 * a2v_conv_wgrad is ready for the next micro-batch without a fill. */
int a2v_unpack_wgrad(float* packed, float* gw, int GN, int Cg, int k, int Cgp, a2v_stream_t stream);
int a2v_col2im(int dtype, const void* dA, void* dX, int B, int T_in, int T_out, int C, int k, int s, int p,
               long long lda, long long ldo, a2v_stream_t stream);
int a2v_cast2d(int in_dtype, int out_dtype, const void* in, void* out, long long M, int C, long long ldi,
               long long ldo, a2v_stream_t stream);

/* keep[i] = 1 if linear index i survives dropout(drop_p, drop_seed) -- the mask every kernel regenerates */
int a2v_dropout_mask(unsigned char* keep, long long n, float drop_p, unsigned drop_seed, a2v_stream_t stream);
/* The attention-probability site uses a 2-D form of the same idea (one hash per pair of adjacent keys of a query row,
 * 16 bits per element): keep-mask of rows x cols with rows = (n*H + h)*S + q, cols = key index. */
int a2v_dropout_mask2d(unsigned char* keep, long long rows, int cols, float drop_p, unsigned drop_seed, a2v_stream_t stream);

/* A/B knobs for kernel variants (tests and tools/ only; defaults are the production paths):
 *   0 bf16 NT GEMM staging (0 register-staged, 1 LDS-DMA)      1 persistent 256x256 NT kernel on / off
 *   2 grouped-conv weight-gradient slab kernel on / off         3 grouped-conv slab kernel on / off
 *   4 persistent 256x256 TN kernel on / off                     5 force the NT tile height (2 / 3 / 4 = 192 / 224 / 256 rows)
 *   6 wave-group stagger (0 off, 1 by shape, 2 always)          7 tile hand-out (0 static, 1 dynamic after the first tile, 2 all dynamic)
 *   8 long-sequence attention backward (0 tiled kernel + dQ atomics, 1 two-pass, 2 two-pass for every S)
 *   10 persistent NT kernel epilogue (1 direct stores, 0 staged through LDS)     13 decoder grouped-conv kernel on / off
 *   14 grouped-conv slab kernel form (1: 384-row slabs, two workgroups per CU; 0: 1024-row slabs)
 *   15 whole-sequence attention backward for S <= 160 (1: 48-row tiles on three waves; 0: 64-row tiles on four)
 *   16 attention forward for S <= 160 (1: whole-sequence kernel, 48 query rows per wave; 0: the tiled kernel)
 *   17 attention forward for S >= 512: 1-D launch that keeps the query blocks of one (n, h) on one XCD (1) / 3-D grid (0)
 *   18 forward-only attention for S >= 512: row sums of P from the bf16 P on the matrix cores (16 x 16 x 32 kernel: 1, default since
 *      round 3; 32 x 32 x 16 kernel of round 5: only with 2 -- there the fp32 sums are faster) / fp32 sums (0)
 *   19 attention forward for S >= 512 with positions = arange: skip key tiles beyond the ALiBi band limit (a bound from the row
 *      norms of q and k: weights below e^-20 of the row sum in the bf16 kernels, e^-32 in the bf16x3 kernel) (1, default) / visit
 *      every tile (0) / v >= 8: on, with e^-v in the bf16 kernels (experiments)
 *   20 fixed-flavour instantiations of the NT kernel's 256-row tile for the eight hottest epilogue flag sets (1, default) / the generic epilogue (0)
 *   21 dense strided convs of the feature extractor on the 8-phase GEMMs + edge-frame kernels (1, default) / implicit-im2col 128 x 128 kernels (0)
 *   22 persistent NT kernel: the rows of a last round of tiles that fills at most v quarters of the CUs go to the 128 x 128 kernel
 *      (1, default; 2 / 3 measured in round 4: no gain) / stay (0)
 *   23 persistent NT kernel tile order: 0 (default) 8 m-tiles x all n-tiles per group; 1 "W-stationary" groups of four n-tiles
 *      (round-4 experiment: +8.6 % fabric reads, 0.6 - 4 % slower -- profiles/r04_nt_tile_ab.txt)
 *   25 the 256 x 128 NT kernel with two workgroups per CU (a2v_gemm_w2.hip, round-4 experiment, profiles/r04_w2_experiment.txt):
 *      0 off (default), 1 eligible Linears with K <= 1024, 2 every eligible Linear
 *   30-34 streaming LayerNorm kernels (round 5; a2v_norm.hip): 30 bit mask of the directions that may take them (1 forward, 2
 *      backward; default 3), 31 rows per batch (0 auto), 32 waves per CU of the persistent form (0 auto), 33 / 34 forward / backward
 *      launch form (0 auto, > 0 batches per wave of the in-order front, -1 persistent)
 *   35 launch order of the grouped-conv kernels: bit 0 conv_nt_slab2_kernel group-fastest, bit 2 conv_dec_kernel halves together
 *      (round-5 experiment, no gain: profiles/r05_conv_order_ab.txt; default 0)
 *   37 row-sparse weight gradients of the positional encoder's first / last layer (a2v_wgrad_sparse.hip) on (1, default) / off (0)
 *   38 conv_dec_kernel work order (round 6): 0 (default) the 64-row blocks of the whole tensor as one in-order front, one workgroup per
 *      CU; v > 0: v workgroups per channel slice; -1 the older order (a workgroup owns 32 consecutive blocks)
 *   39 whole-sequence bf16 attention for S <= 160 (round 6, a2v_attn_seq2.h): 1 (default) the persistent ten-wave kernels for
 *      S = 145 .. 160, 2 for every S <= 160, 0 the round-5 kernels
 *   40 experiment builds with -DA2V_WGRAD_SPLIT only: pos-conv weight gradient with its kk tiles over two workgroups per CU
 *   36 bf16 attention forward for S >= 512, head dimension 64: the 32 x 32 x 16 MFMA kernel of round 5 (1, default) / the
 *      16 x 16 x 32 kernel (0)
 *   (19 also switches the bf16x3 attention forward's band limit, round 4)
 * No launch state goes through here (ABI 5's key 9, the tile-counter set, is gone: the stream argument selects it). */
int a2v_debug_set(int key, int value);

/* Gives the library-owned static scratch set an UNBOUND stream picked up back to the pool (see "re-entrant across streams"
 * above); only needed by hosts that create and destroy many streams without binding workspaces.  -1 if the stream holds no set. */
int a2v_stream_release(a2v_stream_t stream);

/* ---------------------------------------------------------------- caller-owned scratch (ABI 10)
 * a2v_query_workspace: bytes of device scratch the calls on ONE stream need (a constant of the build, about 23.5 MiB: shapes
 * whose partial results would not fit take kernels that need none).
 * a2v_bind_workspace: `ws` = at least that many bytes of device memory on the current device, 256-byte aligned, ZERO when
 * handed over (the kernels leave it zero / self-consistent between calls), alive until every call queued on `stream` has
 * finished or another workspace is bound.  From then on the library uses nothing of its own for `stream` (and no static
 * set is consumed: any number of streams).  ws = NULL unbinds (the stream falls back to the static sets).  Binding is host
 * state only -- no launch, no synchronisation; do it before the first call on the stream, not between dependent calls.
 * Returns -1 for a short / misaligned workspace or an unbind of a stream that was never bound. */
size_t a2v_query_workspace(void);
int a2v_bind_workspace(a2v_stream_t stream, void* ws, size_t bytes);

/* library version / build info */
/* ---------------------------------------------------------------- finetune path
 * Sum-reduced sigmoid focal loss on logits and its gradient (nn/utils.py:971-1010, called with reduction="sum" by
 * FinetuneCrossEntropyCriterion.forward, criterions.py:243-248).  logits / dlogits (M, ld) in the activation dtype
 * with C valid columns (padded columns of dlogits are written as zero; dlogits may be NULL), targets (M, C) fp32
 * (soft after target mixup), loss_sum double accumulated atomically, gscale multiplied into dlogits. */
int a2v_focal_loss(int dtype, const void* logits, const float* targets, void* dlogits, double* loss_sum, long long M, int C,
                   int ld, float alpha, float gamma, float gscale, a2v_stream_t stream);
/* out = scale * sum of L same-shaped tensors (L <= 32, n elements, n % 8 == 0): the mean of the last
 * average_top_k_layers layer results that feeds the classification head (wav2vec2.py:448-466). */
int a2v_mean_layers(int dtype, const void* const* layers, int L, void* out, long long n, float scale, a2v_stream_t stream);

/* out = a + dropout(b + c), n elements (n % 8 == 0); a and c may be NULL.  Keeps the Mlp output t (a layer result,
 * modules.py:328-337) apart from x1 + post_mlp_dropout(t), and carries gradients back through the Mlp's own output
 * dropout.  Mask = counter hash of (drop_seed, element index), as in the GEMM epilogues. */
int a2v_add_drop(int dtype, const void* a, const void* b, const void* c, void* out, long long n, float drop_p,
                 unsigned drop_seed, a2v_stream_t stream);
/* ModalitySpecificEncoder.apply_mask (base.py:457-485) on (B,T,D): frames with mask_t[b*T+t] != 0 become N(0, noise_std)
 * (noise_std 0: zero, encoder_zero_mask), then channels with mask_c[b*D+c] != 0 are zeroed on every frame; either mask
 * may be NULL.  mode 1: the backward of that map (gradient zeroed at replaced positions).  out may alias x. */
int a2v_mask_frames(int dtype, const void* x, const unsigned char* mask_t, const unsigned char* mask_c, void* out,
                    int B, int T, int D, float noise_std, unsigned seed, int mode, a2v_stream_t stream);

/* ---------------------------------------------------------------- span masks on the device
 * Replaces, per forward, fairseq `compute_mask_indices` as ModalitySpecificEncoder.compute_mask calls it
 * (nn/modalities/base.py:401-413: host numpy loop over the B*clone_batch rows + torch.from_numpy(mask).to(device)) and
 * make_maskinfo's two argsorts (base.py:427-455).  Per row max(min_masks, int(mask_prob*T/mask_length + U)) span starts
 * without replacement from [0, T - mask_length), spans of mask_length frames merged; every row cut down, by a uniform draw
 * from its masked frames, to the smallest masked count of the batch (require_same_masks=True, the reference's only
 * setting), then rint(count * mask_dropout) more frames released.  The draw is a pure function of (seed, row, frame)
 * (counter hash; fairseq's numpy stream is third party and not reproducible: SURVEY 8c).
 *   mask (N,T) u8 out, 1 = masked;  counts: N ints of caller-owned scratch;  ids_keep: room for N*T ints, written as
 *   (N,K) kept frames ascending;  ids_restore (N,T): rank among the kept frames, or K + rank among the masked ones
 *   (= argsort(argsort(mask)) with ties in frame order);  keep_out: ONE int on the device that receives K = T - masked
 *   count (data dependent: read it back before shaping ids_keep).  2 <= T <= 16384. */
int a2v_span_mask(unsigned char* mask, int* counts, int* ids_keep, int* ids_restore, int* keep_out, int N, int T,
                  float mask_prob, int mask_length, int min_masks, float mask_dropout, unsigned long long seed,
                  a2v_stream_t stream);
/* ---------------------------------------------------------------- bf16x3 parity mode (ABI 8)
 * x = hi + lo (+ r, |r| <= 2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi); a_hi b_hi + a_lo b_hi + a_hi b_lo misses a b by <= 2^-16
 * relative and is three EXACT bf16 products accumulated in fp32 -- what v_mfma_f32_16x16x32_bf16 does.  a2v_split_bf16 writes the
 * images that put those three products side by side along the reduction axis, so that the bf16 kernels compute them:
 *   mode 0: (M, 3K) [hi | lo | hi]   left operand of a2v_conv_gemm(A2V_BF16X3, ...) (activations)
 *   mode 1: (M, 3K) [hi | hi | lo]   right operand (weights: F.linear's W, or W^T for the input gradient)
 *   mode 2: (2M, K) [hi ; lo]        operands of the weight-gradient products, run as three a2v_conv_wgrad(A2V_BF16, ...) calls
 *                                    (dY_hi, X_hi), (dY_hi, X_lo), (dY_lo, X_hi) into the same fp32 accumulator
 * ldx / ldo: row pitches of x and out in elements (ldo 0 = dense; a larger ldo leaves the caller's zero padding between rows).
 * a2v_conv_gemm(A2V_BF16X3) runs plain row-major products on the persistent 8-phase kernel (taps 1, M >= 1024, N % 256 == 0,
 * K % 128 == 0) and everything else with 64-wide taps -- the convolutions, as implicit im2col of the split activation image with
 * 3 Cg channels per group and tap -- on the 128 x 128 LDS-DMA kernel; -4 (unsupported) otherwise: such shapes stay on A2V_F32.  The Linears of the step
 * (north_star: "attention QKV / out-proj and FFN GEMMs as MFMA bf16 tiles ... outputs within 1e-3") run this way in the
 * engine's bf16x3 mode; the reference has no counterpart (its fp32 is torch's). */
int a2v_split_bf16(const float* x, void* out, long long M, int K, long long ldx, long long ldo, int mode, a2v_stream_t stream);

/* make_maskinfo alone (base.py:427-455) for a caller-supplied mask (`precomputed_mask`) with K kept frames in every row
 * (the reference assumes equal counts, base.py:437): ids_keep (N,K), ids_restore (N,T) as above.  rows_bad (one int the
 * caller zeroed, may be NULL) counts the rows whose kept count is not K; such rows are left unwritten. */
int a2v_maskinfo(const unsigned char* mask, int* ids_keep, int* ids_restore, int* rows_bad, int N, int T, int K,
                 a2v_stream_t stream);

/* 14: a2v_attention_fwd takes `qk_norms` (one more pointer, after dropmask; NULL = the ABI-13 behaviour); + epilogue flag A2V_EPI_QKNORM of a2v_conv_gemm.
 * 13: a2v_attention_fwd / a2v_attention_bwd take `dropmask` (one more pointer each, after mdist; NULL = the ABI-12 arithmetic); + a2v_attention_dropmask_words;
 *     a2v_debug_set key 39 (round-6 whole-sequence attention kernels on / off).
 * 12: + a2v_conv_sparse_in, a2v_conv_wgrad_kept.
 * 11: + epilogue flag A2V_EPI_DACT8 of a2v_conv_gemm; a2v_debug_set keys 30-34 (LayerNorm launch forms); no signature changed.
 * 10: + a2v_query_workspace / a2v_bind_workspace (caller-owned scratch; every other signature unchanged); a2v_debug_set keys 23.
 * 9: + dtype A2V_BF16X3S and epilogue flag A2V_EPI_X3IMG of a2v_conv_gemm; a2v_attention_bwd takes dtype A2V_BF16X3 (signatures unchanged).
 * 8: + a2v_split_bf16 and dtype A2V_BF16X3 of a2v_conv_gemm; + a2v_span_mask, a2v_maskinfo; a2v_attention_fwd / a2v_attention_bwd take `mdist` (one more pointer each, NULL = the ABI-7
 * arithmetic).  7 added a2v_decoder_input_rng and A2V_EPI_DACT over 6. */
int a2v_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* A2V_H */
