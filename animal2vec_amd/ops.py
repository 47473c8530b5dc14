"""Torch-tensor front door of the C ABI (include/a2v.h).  PyTorch is plumbing here: it owns device
memory and the stream; every op below is a hand-written HIP kernel in liba2v_hip.so.  No op has a
fallback: without the library, or on a CPU tensor, they raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import BF16, BF16X3, BF16X3S, F32, EPI_BIAS, EPI_GELU, EPI_RES, EPI_GELU_BWD, EPI_STORE_PRE, EPI_COLSUM, EPI_DROPOUT, EPI_DACT, EPI_DACT8, EPI_X3IMG, EPI_QKNORM, ERR_UNSUPPORTED, check


_HALF = False       # inside half_library(): calls go to liba2v_hip_f16.so, whose 16-bit type is IEEE half (the teacher's forward)


def _L():
    return _lib.load_f16() if _HALF else _lib.load()


class half_library:
    """with ops.half_library(): every op of this module runs in liba2v_hip_f16.so -- the same kernels with torch.float16 as the
    16-bit storage / MFMA operand type (csrc/a2v_common.h).  Python is sequential, so a module switch is enough; tensors of the
    other 16-bit type are refused by dt()."""

    def __enter__(self):
        global _HALF
        self._was, _HALF = _HALF, True
        return self

    def __exit__(self, *a):
        global _HALF
        _HALF = self._was


def have_half_library() -> bool:
    return _lib.have_f16()


def dt(t: torch.Tensor) -> int:
    if t.dtype == (torch.float16 if _HALF else torch.bfloat16):
        return BF16                      # "this library's 16-bit type"
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported activation dtype {t.dtype}" + (" in the fp16 library" if _HALF else ""))


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.A2VError("a2v ops need device tensors (no CPU path exists)")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def per16(t: torch.Tensor) -> int:
    return 16 // t.element_size()


# ---- allocation at capacity.  The student's activations are (sequences x kept frames K) rows long, and K is drawn with the masks:
# with exact-size allocations every step whose K exceeds all earlier ones needs a whole new generation of blocks from the caching
# allocator while the old, slightly smaller ones stay cached -- 169 -> 215 -> 236 GiB reserved over ten 64-clip steps (measured), each
# growth a hipMalloc of tens of GiB inside a step.  Between set_row_cap(N, K, K_cap) and the next call, every tensor this module
# allocates that is K-proportional by SHAPE (first dimension N*K, leading (N, K), or last dimension K) gets the storage it would need at
# K_cap and is returned as a view of its first elements: allocation sizes no longer depend on the draw.
_ROWCAP = None


def set_row_cap(N, K, K_cap):
    global _ROWCAP
    _ROWCAP = (N * K, N, K, K_cap) if (K_cap and 0 < K <= K_cap) else None


def _E(*shape, dtype, device):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    rc = _ROWCAP
    if rc is not None and shape:
        rows, N, K, Kc = rc
        n = 1
        for d in shape:
            n *= d
        if n and n % rows == 0 and (shape[0] == rows or (len(shape) >= 2 and shape[0] == N and shape[1] == K) or shape[-1] == K):
            return torch.empty(n // K * Kc, dtype=dtype, device=device)[:n].view(shape)
    return torch.empty(shape, dtype=dtype, device=device)


def _EL(x):
    return _E(tuple(x.shape), dtype=x.dtype, device=x.device)


def _drop(drop):
    """(p, seed) or None -> C arguments"""
    if drop is None or drop[0] <= 0:
        return 0.0, 0
    return float(drop[0]), int(drop[1]) & 0xFFFFFFFF


# ----------------------------------------------------------------------------------------------
#                                         GEMM family
# ----------------------------------------------------------------------------------------------
def conv_gemm(x, w, out, *, M, N, K, G=1, T_in, T_out, stride=1, pad=0, taps=1, Cg, Cv=0, a_pitch, a_bstride,
              a_goff=0, rowmask=None, bdiv=1, ldb, b_gstride=0, ldc, c_goff=0, bias=None, res=None, ldr=0,
              gelu=False, gelu_bwd=False, pre_out=None, colsum=None, drop=None, rowidx=None, dact=False, x3=None, x3img=False,
              qknorm=None):
    """drop = (p, seed) applies dropout to (acc + bias [gelu]) before the residual add.
    rowidx (int32, M entries): GEMM row m reads the im2col row of flat index rowidx[m]; out stays compact.
    x3=True: x and w are bf16 split images (split_bf16), out / res / pre_out fp32 (include/a2v.h: A2V_BF16X3);
    x3="stacked": x and w are [hi ; lo] stacked images, the three products in one pass (A2V_BF16X3S)."""
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
    if gelu:
        flags |= EPI_GELU
    if res is not None and not gelu_bwd:
        flags |= EPI_RES
    if gelu_bwd:
        flags |= EPI_GELU_BWD
    if pre_out is not None:
        flags |= EPI_STORE_PRE
    if dact:
        flags |= EPI_DACT
        # the derivative tensor in 8 bits (include/a2v.h: A2V_EPI_DACT8): the dtype of the tensor handed in says so
        d8 = pre_out if pre_out is not None else (res if gelu_bwd else None)
        if d8 is not None and d8.dtype == torch.uint8:
            flags |= EPI_DACT8
    if colsum is not None:
        flags |= EPI_COLSUM
    if drop is not None and drop[0] > 0:
        flags |= EPI_DROPOUT
    if x3img:                          # out is the bf16 [hi | lo | hi] image of the result (pitch ldc), include/a2v.h: A2V_EPI_X3IMG
        flags |= EPI_X3IMG
    if qknorm is not None:             # include/a2v.h: A2V_EPI_QKNORM -- the table rides in the colsum argument
        assert colsum is None
        flags |= EPI_QKNORM
        colsum = qknorm
    dp, ds = (drop if drop is not None and drop[0] > 0 else (0.0, 0))
    dtype_code = dt(x) if x3 is None else (BF16X3S if x3 == "stacked" else BF16X3)
    args = (dtype_code, ptr(x), a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, Cv,
            ptr(rowmask), bdiv, ptr(w), ldb, b_gstride, ptr(out), ldc, c_goff, ptr(pre_out),
            ptr(res), ldr if res is not None else 0, ptr(bias), ptr(colsum), M, N, K, G, flags,
            dp, ds & 0xFFFFFFFF, stream())
    rc = _L().a2v_conv_gemm(*args) if rowidx is None else _L().a2v_conv_gemm_rows(ptr(rowidx), *args)
    if qknorm is not None and rc == ERR_UNSUPPORTED:
        return None                    # (no kernel forms the table for this call: linear_qknorm repeats it without)
    check(rc, "a2v_conv_gemm")
    return out


# ---- bf16x3 parity mode (include/a2v.h: a2v_split_bf16, A2V_BF16X3): fp32 storage everywhere, the Linears' products as three
# exact bf16 products on the bf16 matrix cores.  A weight takes part when its split image has been registered (params.Packs with
# x3=True does that at every refresh); activations are split on the fly.  The image hangs on the weight TENSOR OBJECT (attribute
# _a2v_x3), not in a table keyed by address: a freed weight's address is handed to the next allocation.
X3_CONV_FUSED = os.environ.get("A2V_X3_CONV_FUSED", "1") != "0"     # 64 outputs per group: the three split products of a 64-wide grouped conv in one launch (A/B switch)
X3_OUT_IMAGE = os.environ.get("A2V_X3_OUT_IMAGE", "1") != "0"         # the Mlp's hidden activation leaves fc1 as its split image only (A/B switch)
X3_DH_IMAGE = os.environ.get("A2V_X3_DH_IMAGE", "1") != "0"           # the fc2 input gradient leaves its epilogue as a split image only (A/B switch)
DROPMASK = os.environ.get("A2V_DROPMASK", "1") != "0"                # attention dropout: forward writes a bit mask, backward reads it (0: both hash)
X3_ATTN_BWD = os.environ.get("A2V_X3_ATTN_BWD", "1") != "0"          # whole-sequence attention backward with split products (A/B switch)
QKNORM = os.environ.get("A2V_QKNORM", "1") != "0"                   # the teacher's qkv product also forms the row-norm maxima of its attention's band limit (A/B switch)
DACT8 = os.environ.get("A2V_DACT8", "1") != "0"     # bf16 mode: the Mlp's kept gelu' in 8 bits (A/B switch; include/a2v.h: A2V_EPI_DACT8)
X3_WGRAD = False         # the engine of a bf16x3 run switches the weight-gradient products over too (they have no weight operand)
X3_KEEP = False          # set by the engine during the student's saving forward: operand images stay on the activations for the backward
X3_BACKWARD = False      # set by the engine while it runs a backward: a gradient tensor is the left operand of its input-gradient product
                         # AND of a weight-gradient product -- its [hi | lo | hi] image is made once and kept on the tensor object
                         # between the two calls (never in the forward: the saved activations would keep their images alive)


def _x3_cached(t, mode, holder=None):
    """The split image of a 2-D fp32 tensor (mode 0: [hi | lo | hi], mode 2: [hi ; lo]), kept on a tensor OBJECT (`holder`, default
    t itself; never in a table keyed by address: a freed tensor's address is handed to the next allocation) while the engine says a
    second product will want it: during a backward (a gradient is the left operand of its input-gradient product and of a
    weight-gradient product) and during the student's saving forward (the input of a Linear / conv is the operand of its
    weight-gradient product in the backward: the engine saves the very object it passed to the forward product)."""
    key = "_a2v_x3m0" if mode == 0 else "_a2v_x3m2"
    h = t if holder is None else holder
    M, K = t.shape
    img = getattr(h, key, None)
    if img is not None and tuple(img.shape) == ((M, 3 * K) if mode == 0 else (2 * M, K)):
        return img
    img = split_bf16(t, mode)
    if X3_BACKWARD or X3_KEEP:
        setattr(h, key, img)
    return img


class X3Image:
    """A (M, K) fp32 activation of a bf16x3 run that exists ONLY as its [hi | lo | hi] split image -- the Mlp's hidden activation,
    which nothing but two products reads (fc2's forward, fc2's weight gradient): linear(..., out_image=True) writes the image from
    the GEMM epilogue (A2V_EPI_X3IMG) and skips the fp32 tensor and its split pass.  Only the split-product paths accept one."""
    dtype = torch.float32

    def __init__(self, img, M, K):
        self.img, self.shape, self.device = img, (M, K), img.device

    def stride(self, i):
        return (self.shape[1], 1)[i]


def _x3_m0(t):
    return t.img if isinstance(t, X3Image) else _x3_cached(t, 0)


def split_bf16(x, mode, out=None, ldo=0):
    """fp32 (M, K) -> bf16 image: mode 0 (M, 3K) [hi|lo|hi], 1 (M, 3K) [hi|hi|lo], 2 (2M, K) [hi; lo]; ldo: output row pitch
    (the caller's `out`, zeroed where it wants padding)"""
    M, K = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1
    if out is None:
        out = _E((2 * M, K) if mode == 2 else (M, 3 * K), dtype=torch.bfloat16, device=x.device)
    check(_L().a2v_split_bf16(ptr(x), ptr(out), M, K, x.stride(0), ldo, mode, stream()), "a2v_split_bf16")
    return out


def _x3_conv_image(w_packed, k, cv):
    G, N, KK = w_packed.shape
    Cgp = KK // k
    if (3 * Cgp) % 64 or cv % 8 or w_packed.dtype != torch.float32:
        return None
    ent = getattr(w_packed, "_a2v_x3", None)
    if ent is None or ent[0] != ("conv", tuple(w_packed.shape), k, cv):
        return None
    return ent[1]


def x3_register_conv_pack(w_packed, k, cv):
    """(re)build the conv split image; called by params.Packs.refresh in the bf16x3 mode"""
    G, N, KK = w_packed.shape
    Cgp = KK // k
    if (3 * Cgp) % 64 or cv % 8:
        return None
    tag = ("conv", tuple(w_packed.shape), k, cv)
    ent = getattr(w_packed, "_a2v_x3", None)
    img = ent[1] if ent is not None and ent[0] == tag else torch.zeros(G, N, k * 3 * Cgp, dtype=torch.bfloat16, device=w_packed.device)
    rows = w_packed.view(G * N * k, Cgp)
    split_bf16(rows[:, :cv], 1, out=img.view(G * N * k, 3 * Cgp), ldo=3 * Cgp)
    w_packed._a2v_x3 = (tag, img)
    if Cgp == 64 and N <= 64:       # the slab kernel's shapes: hi and lo packs apart, the three products as three launches
        hl = getattr(w_packed, "_a2v_x3hl", None)
        hl = split_bf16(w_packed.view(G * N, KK), 2, out=hl[2] if hl is not None and hl[2].shape == (2 * G * N, KK) else None)
        w_packed._a2v_x3hl = (hl[:G * N].view(G, N, KK), hl[G * N:].view(G, N, KK), hl)
    return img


def x3_register(w, image=None):
    """(re)build the split image of the fp32 (N, K) operand w; called by params.Packs.refresh in the bf16x3 mode"""
    ent = getattr(w, "_a2v_x3", None)
    img = split_bf16(w, 1, out=ent[1] if ent is not None and ent[0] == w.shape else None)
    w._a2v_x3 = (w.shape, img)
    return img


def _x3_image(w, M):
    """the registered split image of w if the product (M rows) is one the 8-phase kernel takes, else None"""
    ent = getattr(w, "_a2v_x3", None)
    if ent is None or ent[0] != w.shape or w.dtype != torch.float32:
        return None
    N, K = w.shape
    return ent[1] if (M >= 1024 and N % 256 == 0 and (3 * K) % 128 == 0 and 3 * K >= 256 and w.stride(0) == K) else None


def linear(x, w, bias=None, *, out=None, res=None, gelu=False, gelu_bwd_aux=None, pre_out=None, colsum=None, drop=None, dact=False,
           out_image=False):
    """out[M,N] = epi(x[M,K] @ w[N,K]^T).  x/w/out/res row-major 2-D (leading dims from strides).
    out_image (bf16x3 runs): where the split-product path takes this product and the result's own products will too, the result is
    returned as an X3Image (written by the epilogue) instead of an fp32 tensor."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
    r = gelu_bwd_aux if gelu_bwd_aux is not None else res
    img = _x3_image(w, M) if x.dtype == torch.float32 else None
    if isinstance(x, X3Image) and img is None:
        raise RuntimeError("an image-only activation needs the split-product path (registered weight image, M >= 1024)")
    if (out_image and img is not None and X3_WGRAD and X3_OUT_IMAGE and out is None and res is None and (drop is None or drop[0] <= 0)
            and M >= 2048 and N % 256 == 0):
        # (round 4: also the fc2 input gradient -- GELU_BWD with its fp32 derivative tensor in R -- whose result nothing but fc1's
        #  input-gradient and weight-gradient products reads)
        oimg = _E(M, 3 * N, dtype=torch.bfloat16, device=w.device)
        conv_gemm(_x3_m0(x), img, oimg, M=M, N=N, K=3 * K, T_in=M, T_out=M, Cg=3 * K, a_pitch=3 * K, a_bstride=0, ldb=3 * K,
                  ldc=3 * N, bias=bias, res=r, ldr=r.stride(0) if r is not None else 0, gelu=gelu, gelu_bwd=gelu_bwd_aux is not None,
                  pre_out=pre_out, colsum=colsum, dact=dact, x3=True, x3img=True)
        return X3Image(oimg, M, N)
    if out is None:
        out = _E(M, N, dtype=x.dtype, device=w.device)
    if img is not None:
        xs = _x3_m0(x)
        return conv_gemm(xs, img, out, M=M, N=N, K=3 * K, T_in=M, T_out=M, Cg=3 * K, a_pitch=3 * K, a_bstride=0, ldb=3 * K,
                         ldc=out.stride(0), bias=bias, res=r, ldr=r.stride(0) if r is not None else 0, gelu=gelu,
                         gelu_bwd=gelu_bwd_aux is not None, pre_out=pre_out, colsum=colsum, drop=drop, dact=dact, x3=True)
    return conv_gemm(x, w, out, M=M, N=N, K=K, T_in=M, T_out=M, Cg=K, a_pitch=x.stride(0), a_bstride=0,
                     ldb=w.stride(0), ldc=out.stride(0), bias=bias, res=r, ldr=r.stride(0) if r is not None else 0,
                     gelu=gelu, gelu_bwd=gelu_bwd_aux is not None, pre_out=pre_out, colsum=colsum, drop=drop, dact=dact)


def linear_qknorm(x, w, bias, S):
    """The qkv Linear of a long-sequence attention layer (x: N clips x S rows, row-major): out = x @ w^T + bias, and -- where the
    library's persistent GEMM takes the product -- the (N, w.shape[0] / 32) fp32 table of per-clip maxima of the squared norms of the
    output's 32-column segments (include/a2v.h: A2V_EPI_QKNORM), which attention_fwd(qk_norms=...) turns into its ALiBi band limit
    instead of reading q and k again.  -> (out, table or None)."""
    M, K = x.shape
    N = w.shape[0]
    if (QKNORM and _L()._a2v_abi >= 14 and x.dtype == w.dtype and dt(x) == BF16 and x.dtype != torch.float32 and bias is not None
            and M % S == 0 and S >= 512 and N % 256 == 0 and x.stride(1) == 1 and w.stride(1) == 1):
        out = _E(M, N, dtype=x.dtype, device=w.device)
        table = torch.zeros(M // S, N // 32, dtype=torch.float32, device=w.device)
        r = conv_gemm(x, w, out, M=M, N=N, K=K, T_in=S, T_out=S, Cg=K, a_pitch=x.stride(0), a_bstride=S * x.stride(0),
                      ldb=w.stride(0), ldc=out.stride(0), bias=bias, qknorm=table)
        if r is not None:
            return out, table
    return linear(x, w, bias), None


def conv1d_cl(x, w_packed, bias, *, k, stride, pad, groups, T_out, n_out, out=None, rowmask=None, bdiv=1,
              n_batch=None, res=None, colsum=None, cv=None, rowidx=None):
    """Channels-last Conv1d as implicit GEMM.  x (Bsrc,T_in,Cp); w_packed (G, n_out/G, k*Cg) where Cg is
    the (possibly zero-padded) tap width; cv = real input channels per group (default Cg);
    out (B,T_out,n_out) with B = n_batch or Bsrc."""
    Bsrc, T_in, Cp = x.shape
    B = n_batch if n_batch is not None else Bsrc
    G = groups
    Cg = w_packed.shape[-1] // k
    cv = Cg if cv is None else cv
    N = n_out // G
    M = B * T_out if rowidx is None else rowidx.numel()       # rowidx: only these rows (flat b*T_out + t), compact output
    if out is None:
        out = _E((B, T_out, n_out) if rowidx is None else (M, n_out), dtype=x.dtype, device=x.device)
    img = _x3_conv_image(w_packed, k, cv) if x.dtype == torch.float32 else None
    hl = getattr(w_packed, "_a2v_x3hl", None) if img is not None else None
    if (hl is not None and x.is_contiguous() and stride == 1 and T_in == T_out and T_out >= 256 and rowidx is None and Cg == 64
            and N <= 64 and N % 4 == 0 and cv % 8 == 0 and 2 <= k <= 24):
        # bf16x3 on the slab kernel (64-wide taps): X_hi W_hi (+ bias, residual), then X_lo W_hi and X_hi W_lo accumulated through R = out
        R_ = Bsrc * T_in
        xs = _x3_cached(x.view(R_, Cp), 2, holder=x)
        xh, xl = xs[:R_].view(x.shape), xs[R_:].view(x.shape)
        kw = dict(M=M, N=N, K=k * Cg, G=G, T_in=T_in, T_out=T_out, stride=stride, pad=pad, taps=k, Cg=Cg, Cv=cv, a_pitch=x.stride(1),
                  a_bstride=x.stride(0), a_goff=cv if G > 1 else 0, rowmask=rowmask, bdiv=bdiv, ldb=w_packed.stride(-2),
                  b_gstride=w_packed.stride(0) if G > 1 else 0, ldc=n_out, c_goff=N, x3=True)
        if N == 64 and colsum is None and X3_CONV_FUSED:
            # 64 outputs per group (the pos-conv): one pass over both slabs (conv_nt_slab3_kernel); xs / hl[2] ARE the stacked images
            kw["x3"] = "stacked"
            return conv_gemm(xs, hl[2], out, bias=bias, res=res, ldr=n_out if res is not None else 0, **kw)
        conv_gemm(xh, hl[0], out, bias=bias, res=res, ldr=n_out if res is not None else 0, **kw)
        conv_gemm(xl, hl[0], out, res=out, ldr=n_out, **kw)
        conv_gemm(xh, hl[1], out, res=out, ldr=n_out, colsum=colsum, **kw)
        return out
    if img is not None and x.is_contiguous() and Cp == G * cv:
        # bf16x3: implicit im2col of the split activation image -- 3 cv channels per group [hi | lo | hi] against taps [hi | hi | lo]
        xs = split_bf16(x.view(-1, cv), 0).view(Bsrc, T_in, G * 3 * cv)
        return conv_gemm(xs, img, out, M=M, N=N, K=k * 3 * Cg, G=G, T_in=T_in, T_out=T_out, stride=stride, pad=pad, taps=k,
                         Cg=3 * Cg, Cv=3 * cv, a_pitch=G * 3 * cv, a_bstride=T_in * G * 3 * cv, a_goff=3 * cv if G > 1 else 0,
                         rowmask=rowmask, bdiv=bdiv, ldb=img.stride(-2), b_gstride=img.stride(0) if G > 1 else 0, ldc=n_out,
                         c_goff=N, bias=bias, res=res, ldr=n_out if res is not None else 0, colsum=colsum, rowidx=rowidx, x3=True)
    return conv_gemm(x, w_packed, out, M=M, N=N, K=k * Cg, G=G, T_in=T_in, T_out=T_out, stride=stride,
                     pad=pad, taps=k, Cg=Cg, Cv=cv, a_pitch=x.stride(1), a_bstride=x.stride(0), a_goff=cv if G > 1 else 0,
                     rowmask=rowmask, bdiv=bdiv, ldb=w_packed.stride(-2), b_gstride=w_packed.stride(0) if G > 1 else 0,
                     ldc=n_out, c_goff=N, bias=bias, res=res, ldr=n_out if res is not None else 0, colsum=colsum,
                     rowidx=rowidx)


def _has_entry(name):
    """False only for an older build loaded through A2V_LIB_PATH (same-box A/B runs): _lib.load()"""
    return hasattr(_L(), name)


CONV_SPARSE = os.environ.get("A2V_CONV_SPARSE", "1") != "0"     # =0: the row-sparse pos-conv products (a2v_conv_sparse_in) off (A/B runs)


def conv_sparse_ok(x, w_packed, *, k, groups, n_out, rows_per_seq=0):
    """a2v_conv_sparse_in takes it: bf16, the positional encoder's shape (19 taps, 64 channels per group in and out); a row list
    of at most 1024 rows per sequence"""
    return (CONV_SPARSE and rows_per_seq <= 1024 and _has_entry("a2v_conv_sparse_in") and x.dtype == torch.bfloat16 and w_packed.dtype == torch.bfloat16 and k == 19 and n_out == groups * 64
            and w_packed.shape[-1] == k * 64 and w_packed.shape[-2] == 64 and x.stride(-1) == 1)


def conv1d_cl_sparse_in(x, w_packed, bias, *, k, pad, groups, T, n_out, n_seq, rowmask=None, bdiv=1, rowidx=None, out=None):
    """Channels-last grouped conv whose INPUT is row-sparse (include/a2v.h: a2v_conv_sparse_in) -> (n_seq, T, n_out).
    rowmask (n_seq, T) u8: x (Bsrc, T, C) holds the sequences (clone c reads sequence c // bdiv), non-zero where the mask is 0;
    rowidx (n_seq * K) int32: x (n_seq * K, C) is compact, row i belongs at flat frame rowidx[i]."""
    G = groups
    if out is None:
        out = _E(n_seq, T, n_out, dtype=x.dtype, device=x.device)
    if rowidx is not None:
        K = rowidx.numel() // n_seq
        assert x.dim() == 2 and x.shape[0] == rowidx.numel() and K * n_seq == rowidx.numel()
        xp, xb = x.stride(0), 0
    else:
        K = 0
        xp, xb = x.stride(1), x.stride(0)
    rc = _L().a2v_conv_sparse_in(dt(x), ptr(x), xp, xb, 64 if G > 1 else 0, ptr(rowmask), bdiv, ptr(rowidx), K, n_seq, T, pad, k,
                                 ptr(w_packed), w_packed.stride(-2), w_packed.stride(0) if G > 1 else 0, ptr(bias), ptr(out), n_out,
                                 64, G, stream())
    check(rc, "a2v_conv_sparse_in")
    return out


def conv_wgrad(x, dy, gw, *, M, N, K, G=1, T_in, T_out, stride=1, pad=0, taps=1, Cg, Cv=0, a_pitch, a_bstride, a_goff=0,
               rowmask=None, bdiv=1, ldp, p_goff=0, ldw, w_gstride=0, rowidx=None, n_seq=0):
    """n_seq (with rowidx): the rows are the kept frames of n_seq sequences, M / n_seq each, ascending (masking.make_maskinfo) --
    the kept-frame entry (include/a2v.h: a2v_conv_wgrad_kept), same result as the gathered one."""
    args = (dt(x), ptr(x), a_pitch, a_bstride, a_goff, T_in, T_out, stride, pad, taps, Cg, Cv,
            ptr(rowmask), bdiv, ptr(dy), ldp, p_goff, ptr(gw), ldw, w_gstride, M, N, K, G, stream())
    if rowidx is None:
        rc = _L().a2v_conv_wgrad(*args)
    elif n_seq > 0 and M % n_seq == 0 and _has_entry("a2v_conv_wgrad_kept"):
        rc = _L().a2v_conv_wgrad_kept(ptr(rowidx), n_seq, *args)
    else:
        rc = _L().a2v_conv_wgrad_rows(ptr(rowidx), *args)
    check(rc, "a2v_conv_wgrad")
    return gw


def _x3_wgrad_ok(x, dy):
    M, K = x.shape
    N = dy.shape[1]
    ok = (X3_WGRAD and x.dtype == torch.float32 and dy.dtype == torch.float32 and M >= 2048 and N % 256 == 0 and K % 256 == 0
          and x.stride(1) == 1 and dy.stride(1) == 1)
    if (isinstance(x, X3Image) or isinstance(dy, X3Image)) and not ok:
        # (dh leaves the fc2 input-gradient epilogue as an image only -- the plain path below would take ptr() of it)
        raise RuntimeError(f"an image-only operand needs the split-product weight-gradient path (M={M} >= 2048, N={N} % 256, K={K} % 256, "
                           "fp32 storage, ops.X3_WGRAD)")
    return ok


def _x3_wgrad_terms(x, dy):
    """(X_hi, dY_hi), (X_lo, dY_hi), (X_hi, dY_lo) as bf16 column views (pitch 3 K / 3 N) of the [hi | lo | hi] images"""
    K, N = x.shape[1], dy.shape[1]
    xs, ds = _x3_m0(x), _x3_m0(dy)
    return ((xs[:, :K], ds[:, :N]), (xs[:, K:2 * K], ds[:, :N]), (xs[:, :K], ds[:, N:2 * N]))


def linear_wgrad(x, dy, gw):
    """gw[N,K] (fp32) += dy[M,N]^T @ x[M,K]"""
    M, K = x.shape
    N = dy.shape[1]
    assert gw.dtype == torch.float32 and gw.shape == (N, K)
    if _x3_wgrad_ok(x, dy):
        for xa, da in _x3_wgrad_terms(x, dy):
            conv_wgrad(xa, da, gw, M=M, N=N, K=K, T_in=M, T_out=M, Cg=K, a_pitch=3 * K, a_bstride=0, ldp=3 * N, ldw=gw.stride(0))
        return gw
    return conv_wgrad(x, dy, gw, M=M, N=N, K=K, T_in=M, T_out=M, Cg=K, a_pitch=x.stride(0), a_bstride=0,
                      ldp=dy.stride(0), ldw=gw.stride(0))


def linear_wgrad_pair(xa, dya, gwa, xb, dyb, gwb):
    """gwa[Na,Ka] += dya^T @ xa and gwb[Nb,Kb] += dyb^T @ xb (same M rows) in one launch where the kernel allows."""
    M, Ka = xa.shape
    Na, Nb, Kb = dya.shape[1], dyb.shape[1], xb.shape[1]
    assert xb.shape[0] == M and dya.shape[0] == M and dyb.shape[0] == M and xa.dtype == xb.dtype
    assert gwa.dtype == torch.float32 and gwb.dtype == torch.float32 and gwa.shape == (Na, Ka) and gwb.shape == (Nb, Kb)
    ok_a, ok_b = _x3_wgrad_ok(xa, dya), _x3_wgrad_ok(xb, dyb)     # (both evaluated: each raises for an image-only operand it cannot take)
    if ok_a and ok_b:
        for (xa_, da_), (xb_, db_) in zip(_x3_wgrad_terms(xa, dya), _x3_wgrad_terms(xb, dyb)):
            rc = _L().a2v_linear_wgrad_pair(BF16, M, ptr(xa_), 3 * Ka, ptr(da_), 3 * Na, ptr(gwa), gwa.stride(0), Na, Ka,
                                            ptr(xb_), 3 * Kb, ptr(db_), 3 * Nb, ptr(gwb), gwb.stride(0), Nb, Kb, stream())
            check(rc, "a2v_linear_wgrad_pair")
        return
    rc = _L().a2v_linear_wgrad_pair(dt(xa), M, ptr(xa), xa.stride(0), ptr(dya), dya.stride(0), ptr(gwa), gwa.stride(0), Na, Ka,
                                    ptr(xb), xb.stride(0), ptr(dyb), dyb.stride(0), ptr(gwb), gwb.stride(0), Nb, Kb, stream())
    check(rc, "a2v_linear_wgrad_pair")


def conv1d_cl_wgrad(x, dy, gw_packed, *, k, stride, pad, groups, T_out, rowmask=None, bdiv=1, n_batch=None, cv=None,
                    rowidx=None, rows_kept=False):
    """gw_packed (G, N, k*Cg) fp32 += implicit-im2col(x)^T dy.  rowidx: dy holds only those rows (compact); rows_kept: they are
    the kept frames of the batch's sequences, equally many per sequence and ascending (conv_wgrad's n_seq)."""
    Bsrc, T_in, Cp = x.shape
    B = n_batch if n_batch is not None else Bsrc
    G = groups
    Cg = gw_packed.shape[-1] // k
    N = gw_packed.shape[-2]
    n_out = dy.shape[-1]
    cv = Cg if cv is None else cv
    if X3_WGRAD and x.dtype == torch.float32 and dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous() and Cg % 8 == 0:
        # bf16x3: dY_hi^T X_hi + dY_hi^T X_lo + dY_lo^T X_hi through the bf16 weight-gradient kernels (fp32 accumulation)
        R = x.shape[0] * x.shape[1]
        xs = _x3_cached(x.view(R, Cp), 2, holder=x)
        d2 = dy.reshape(-1, n_out)
        ds = _x3_cached(d2, 2, holder=dy)
        Rd = d2.shape[0]
        xh, xl = xs[:R].view(x.shape), xs[R:].view(x.shape)
        dh, dl = ds[:Rd].view(dy.shape), ds[Rd:].view(dy.shape)
        for xa, da in ((xh, dh), (xl, dh), (xh, dl)):
            conv_wgrad(xa, da, gw_packed, M=B * T_out if rowidx is None else rowidx.numel(), N=N, K=k * Cg, G=G, T_in=T_in,
                       T_out=T_out, stride=stride, pad=pad, rowidx=rowidx, taps=k, Cg=Cg, Cv=cv, a_pitch=x.stride(1),
                       a_bstride=x.stride(0), a_goff=cv if G > 1 else 0, rowmask=rowmask, bdiv=bdiv, ldp=n_out, p_goff=N,
                       ldw=gw_packed.stride(-2), w_gstride=gw_packed.stride(0) if G > 1 else 0)
        return gw_packed
    return conv_wgrad(x, dy, gw_packed, M=B * T_out if rowidx is None else rowidx.numel(), N=N, K=k * Cg, G=G, T_in=T_in,
                      T_out=T_out, stride=stride, pad=pad, rowidx=rowidx, n_seq=B if (rowidx is not None and rows_kept) else 0,
                      taps=k, Cg=Cg, Cv=cv, a_pitch=x.stride(1), a_bstride=x.stride(0), a_goff=cv if G > 1 else 0,
                      rowmask=rowmask, bdiv=bdiv, ldp=n_out, p_goff=N, ldw=gw_packed.stride(-2),
                      w_gstride=gw_packed.stride(0) if G > 1 else 0)


# ----------------------------------------------------------------------------------------------
#                                         attention
# ----------------------------------------------------------------------------------------------
def attention_fwd(qkv, pos, slopes, alibi_scale, H, *, need_lse=True, drop=None, qk_norms=None):
    """qkv (N,S,3*H*HD) -> out (N,S,H*HD), lse (2,N,H,S): [0] the log-sum-exp, [1] the mean attended distance
    sum_j P_ij |pos_i - pos_j| of every row (what attention_bwd centres the ALiBi distance on; include/a2v.h)"""
    N, S, W = qkv.shape
    HD = W // (3 * H)
    out = _E(N, S, H * HD, dtype=qkv.dtype, device=qkv.device)
    lse = _E(2, N, H, S, dtype=torch.float32, device=qkv.device) if need_lse else None
    # bf16x3 run (fp32 storage): the forward's two products split on the bf16 matrix cores (the backward: attention_bwd below)
    code = BF16X3 if (X3_WGRAD and qkv.dtype == torch.float32 and HD in (32, 64)) else dt(qkv)
    # the whole-sequence kernels (bf16, HD 64, S <= 160) decide the dropout once, in the forward, and hand the backward a bit mask
    # (csrc/a2v_attn_seq2.h); it rides on the lse tensor, the object callers keep for the backward
    dm = None
    abi13 = _L()._a2v_abi >= 13                      # (an ABI-12 library: A2V_LIB_PATH, same-box A/B against the previous round)
    if abi13 and need_lse and drop is not None and drop[0] > 0 and DROPMASK:
        words = _L().a2v_attention_dropmask_words(code, S, HD)
        if words:
            dm = _E(N * H * words, dtype=torch.int32, device=qkv.device)
    abi14 = _L()._a2v_abi >= 14
    rc = _L().a2v_attention_fwd(code, ptr(qkv), ptr(out), ptr(lse), ptr(lse[1]) if need_lse else None, *((ptr(dm),) if abi13 else ()),
                                *((ptr(qk_norms),) if abi14 else ()), ptr(pos), ptr(slopes), ptr(alibi_scale), N, S, H, HD, HD ** -0.5, *(_drop(drop)), stream())
    check(rc, "a2v_attention_fwd")
    if dm is not None:
        lse._a2v_dropmask = dm
    return out, lse


def attention_bwd(qkv, out, d_out, lse, pos, slopes, alibi_scale, H, dalibi, drop=None, bias_grad=None):
    """-> dqkv (N,S,3*H*HD).  dalibi (H) fp32 is accumulated; bias_grad (3*H*HD fp32, the qkv Linear's bias gradient =
    column sums of dqkv) is accumulated too -- inside the kernel where it can (whole-sequence kernel), else by a pass
    over dqkv."""
    N, S, W = qkv.shape
    HD = W // (3 * H)
    dqkv = _EL(qkv)
    # bf16x3 run (fp32 storage): whole sequences of up to 160 tokens at HD = 64 take the split-product backward (attn_bwd_x3_kernel)
    code = BF16X3 if (X3_WGRAD and X3_ATTN_BWD and qkv.dtype == torch.float32 and HD == 64) else dt(qkv)
    direct = bool(_L().a2v_attention_bwd_direct_dq(code, S, HD))
    fused = bias_grad is not None and bool(_L().a2v_attention_bwd_fused_bias(code, S, HD))
    dq32 = None if direct else torch.zeros(N * S, H * HD, dtype=torch.float32, device=qkv.device)
    delta = _E(N, H, S, dtype=torch.float32, device=qkv.device)
    md = lse[1] if lse.dim() == 4 else None           # (a bare (N,H,S) lse: the uncentred arithmetic)
    dm = getattr(lse, "_a2v_dropmask", None) if (drop is not None and drop[0] > 0) else None
    rc = _L().a2v_attention_bwd(code, ptr(qkv), ptr(out), ptr(d_out), ptr(lse), ptr(md), *((ptr(dm),) if _L()._a2v_abi >= 13 else ()),
                                ptr(delta), ptr(dqkv), ptr(dq32),
                                ptr(dalibi), ptr(bias_grad) if fused else None, ptr(pos), ptr(slopes), ptr(alibi_scale), N, S, H,
                                HD, HD ** -0.5, *(_drop(drop)), stream())
    check(rc, "a2v_attention_bwd")
    if not direct:
        cast2d(dq32, dqkv.view(N * S, W)[:, : H * HD])
    if bias_grad is not None and not fused:
        colsum(dqkv.view(N * S, W), bias_grad)
    return dqkv


# ----------------------------------------------------------------------------------------------
#                                         normalisation
# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, *, res=None, act=0, save_stats=True, out=None, sum_out=None, post=None, drop=None):
    """x (..., C) contiguous rows.  Returns y, mean, rstd."""
    C_ = x.shape[-1]
    M = x.numel() // C_
    y = out if out is not None else _EL(x)
    mean = _E(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = _E(M, dtype=torch.float32, device=x.device) if save_stats else None
    rc = _L().a2v_layernorm_fwd(dt(x), ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(y), ptr(sum_out), ptr(mean),
                                ptr(rstd), ptr(post), M, C_, C_, C_, eps, act, *(_drop(drop)), stream())
    check(rc, "a2v_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, beta, mean, rstd, *, act=0, dgamma=None, dbeta=None, dxsum=None, dres=None, out=None,
                  drop=None, drop_mode=0):
    """drop_mode 1: mask incoming dy (dropout followed this LN); drop_mode 2: also return dropout(dx) (second result)."""
    C_ = x.shape[-1]
    M = x.numel() // C_
    dx = out if out is not None else _EL(x)
    mode = drop_mode if (drop is not None and drop[0] > 0) else 0
    dx_drop = _EL(x) if mode == 2 else None
    rc = _L().a2v_layernorm_bwd(dt(x), ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dres),
                                ptr(dgamma), ptr(dbeta), ptr(dxsum), M, C_, C_, C_, act, *(_drop(drop)), mode, ptr(dx_drop),
                                stream())
    check(rc, "a2v_layernorm_bwd")
    if drop_mode == 2:
        return dx, (dx_drop if dx_drop is not None else dx)
    return dx


def instnorm_targets(layers: Sequence[torch.Tensor], eps=1e-5):
    """make_targets: mean over layers of instance_norm over T.  layers: list of (B,T,D) -> fp32 (B,T,D)"""
    L = len(layers)
    B, T, D = layers[0].shape
    dev = layers[0].device
    arr = (C.c_void_p * L)(*[ptr(t) for t in layers])
    mean = _E(L * B * D, dtype=torch.float32, device=dev)
    rstd = _E(L * B * D, dtype=torch.float32, device=dev)
    y = _E(B, T, D, dtype=torch.float32, device=dev)
    rc = _L().a2v_instnorm_targets(dt(layers[0]), arr, L, ptr(mean), ptr(rstd), ptr(y), B, T, D, eps, stream())
    check(rc, "a2v_instnorm_targets")
    return y


def mean_layers(layers: Sequence[torch.Tensor], scale=None):
    """scale * sum of same-shaped tensors (default 1/L: the mean)."""
    L = len(layers)
    arr = (C.c_void_p * L)(*[ptr(t) for t in layers])
    out = _EL(layers[0])
    rc = _L().a2v_mean_layers(dt(layers[0]), arr, L, ptr(out), out.numel(), 1.0 / L if scale is None else scale, stream())
    check(rc, "a2v_mean_layers")
    return out


def focal_loss(logits2d, targets2d, loss_sum, *, alpha=0.25, gamma=2.0, gscale=1.0, want_grad=True):
    """logits2d (M, ld) activation dtype with targets2d.shape[1] valid columns; loss_sum (1,) float64 accumulated.
    Returns dlogits (M, ld) or None."""
    M, ld = logits2d.shape
    C_ = targets2d.shape[1]
    assert targets2d.dtype == torch.float32 and targets2d.shape[0] == M and loss_sum.dtype == torch.float64
    dl = _EL(logits2d) if want_grad else None
    rc = _L().a2v_focal_loss(dt(logits2d), ptr(logits2d), ptr(targets2d), ptr(dl), ptr(loss_sum), M, C_, ld, alpha, gamma,
                             gscale, stream())
    check(rc, "a2v_focal_loss")
    return dl


def colsum(x2d, out):
    M, C_ = x2d.shape
    rc = _L().a2v_colsum(dt(x2d), ptr(x2d), ptr(out), M, C_, x2d.stride(0), stream())
    check(rc, "a2v_colsum")
    return out


# ----------------------------------------------------------------------------------------------
#                                         sinc front end
# ----------------------------------------------------------------------------------------------
def sinc_filters_fwd(low_hz, band_hz, n_, window, *, C_, Cp, k, min_low_hz, min_band_hz, sample_rate):
    filt = _E(Cp, k, dtype=torch.float32, device=low_hz.device)
    rc = _L().a2v_sinc_filters_fwd(ptr(low_hz), ptr(band_hz), ptr(n_), ptr(window), ptr(filt), C_, Cp, k,
                                   float(min_low_hz), float(min_band_hz), float(sample_rate), stream())
    check(rc, "a2v_sinc_filters_fwd")
    return filt


def sinc_filters_bwd(low_hz, band_hz, n_, window, dfilt, *, C_, k, min_low_hz, min_band_hz, sample_rate, dlow=None,
                     dband=None):
    dlow = dlow if dlow is not None else torch.zeros(C_, dtype=torch.float32, device=low_hz.device)
    dband = dband if dband is not None else torch.zeros(C_, dtype=torch.float32, device=low_hz.device)
    rc = _L().a2v_sinc_filters_bwd(ptr(low_hz), ptr(band_hz), ptr(n_), ptr(window), ptr(dfilt), ptr(dlow), ptr(dband),
                                   C_, k, float(min_low_hz), float(min_band_hz), float(sample_rate), stream())
    check(rc, "a2v_sinc_filters_bwd")
    return dlow, dband


def sinc_out_channels(C_, k):
    """channel pitch of the fused SincConv output for a (filters, taps) spec (128 or 320); raises if unsupported"""
    cp = _L().a2v_sinc_out_channels(C_, k)
    if cp < 0:
        raise ValueError(f"SincConv front end: {C_} filters x {k} taps is outside the kernels' limits (<= 320 x 255, odd k)")
    return cp


def sinc_fused_fwd(x, filt, ln_w, ln_b, ps_a, ps_b, *, C_, k, dtype, eps=1e-5, save_stats=True):
    B, L = x.shape
    y = _E(B, L, sinc_out_channels(C_, k), dtype=dtype, device=x.device)
    mean = _E(B, L, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = _E(B, L, dtype=torch.float32, device=x.device) if save_stats else None
    rc = _L().a2v_sinc_fused_fwd(dt(y), ptr(x), ptr(filt), ptr(ln_w), ptr(ln_b), ptr(ps_a), ptr(ps_b), ptr(y), ptr(mean),
                                 ptr(rstd), B, L, C_, k, eps, stream())
    check(rc, "a2v_sinc_fused_fwd")
    return y, mean, rstd


def sinc_fused_bwd(x, filt, ln_w, ln_b, ps_a, ps_b, dy, mean, rstd, dfilt, dpar, *, C_, k):
    B, L = x.shape
    cp = sinc_out_channels(C_, k)
    scratch = _E(B, L, cp, dtype=torch.float32, device=x.device) if cp != 128 else None     # blocked kernels: dconv
    rc = _L().a2v_sinc_fused_bwd(dt(dy), ptr(x), ptr(filt), ptr(ln_w), ptr(ln_b), ptr(ps_a), ptr(ps_b), ptr(dy), ptr(mean),
                                 ptr(rstd), ptr(dfilt), ptr(dpar), ptr(scratch), B, L, C_, k, stream())
    check(rc, "a2v_sinc_fused_bwd")


# ----------------------------------------------------------------------------------------------
#                                         tokens / loss / optimizer
# ----------------------------------------------------------------------------------------------
def span_mask(N, T, *, mask_prob, mask_length, seed, min_masks=1, mask_dropout=0.0, device="cuda", bufs=None):
    """Device span-mask sampler + maskinfo (include/a2v.h: a2v_span_mask; reference base.py:401-455).  Returns
    (mask_u8 (N,T), ids_keep_flat (N*T int32 -- view the first N*K as (N,K) once K is known), ids_restore (N,T) int32,
    k_dev (1,) int32).  Asynchronous on the current stream; K = k_dev[0] is data dependent.  bufs: the same tuple (plus
    the counts scratch) from an earlier call, to be overwritten."""
    if bufs is None:
        dev = torch.device(device)
        bufs = (_E(N, T, dtype=torch.uint8, device=dev), _E(N * T, dtype=torch.int32, device=dev),
                _E(N, T, dtype=torch.int32, device=dev), _E(1, dtype=torch.int32, device=dev),
                _E(N, dtype=torch.int32, device=dev))
    mask, keep, restore, kdev, counts = bufs
    rc = _L().a2v_span_mask(ptr(mask), ptr(counts), ptr(keep), ptr(restore), ptr(kdev), N, T, float(mask_prob), int(mask_length),
                            int(min_masks), float(mask_dropout), int(seed) & 0xFFFFFFFFFFFFFFFF, stream())
    check(rc, "a2v_span_mask")
    return bufs


def maskinfo(mask_u8, K, rows_bad=None):
    """make_maskinfo (base.py:427-455) of a given (N,T) uint8 mask with K kept frames per row -> ids_keep (N,K) ascending,
    ids_restore (N,T), both int32."""
    N, T = mask_u8.shape
    keep = _E(N, K, dtype=torch.int32, device=mask_u8.device)
    restore = _E(N, T, dtype=torch.int32, device=mask_u8.device)
    rc = _L().a2v_maskinfo(ptr(mask_u8), ptr(keep), ptr(restore), ptr(rows_bad), N, T, int(K), stream())
    check(rc, "a2v_maskinfo")
    return keep, restore


def gather_tokens(a, ids, *, cb=1, b=None, drop=None):
    """out[n,k,:] = a[n//cb, ids[n,k], :] (+ b[n, ids[n,k], :]);  a (Ba,Ta,D), ids (N,K) int32"""
    N, K = ids.shape
    Ta, D = a.shape[1], a.shape[2]
    out = _E(N, K, D, dtype=a.dtype, device=a.device)
    rc = _L().a2v_gather_tokens(dt(a), ptr(a), ptr(b), ptr(ids), ptr(out), N, K, D, Ta, cb, *(_drop(drop)), stream())
    check(rc, "a2v_gather_tokens")
    return out


def scatter_tokens(dtok, restore, T):
    N, K, D = dtok.shape
    out = _E(N, T, D, dtype=dtok.dtype, device=dtok.device)
    rc = _L().a2v_scatter_tokens(dt(dtok), ptr(dtok), ptr(restore), ptr(out), N, K, D, T, stream())
    check(rc, "a2v_scatter_tokens")
    return out


def clone_reduce(dxm, dtok, restore, cb):
    """dfeat[b,t] = sum over the clones n of b that keep frame t of (dxm[n,t] + dtok[n, restore[n,t]]); dxm None:
    only the dtok term (it then already carries the pos-conv input gradient of the kept rows)."""
    N, K, D = dtok.shape
    T = restore.shape[1]
    out = _E(N // cb, T, D, dtype=dtok.dtype, device=dtok.device)
    rc = _L().a2v_clone_reduce(dt(dtok), ptr(dxm), ptr(dtok), ptr(restore), ptr(out), N // cb, cb, K, D, T, stream())
    check(rc, "a2v_clone_reduce")
    return out


def decoder_input(x, noise, restore, drop=None):
    """noise: (N, T-K, D) fp32 rows, or a (std, seed) pair -- then the masked positions draw N(0, std) inside the kernel."""
    N, K, D = x.shape
    T = restore.shape[1]
    out = _E(N, T, D, dtype=x.dtype, device=x.device)
    if isinstance(noise, tuple):
        rc = _L().a2v_decoder_input_rng(dt(x), ptr(x), ptr(restore), ptr(out), N, K, D, T, float(noise[0]),
                                        int(noise[1]) & 0xFFFFFFFF, *(_drop(drop)), stream())
        check(rc, "a2v_decoder_input_rng")
        return out
    rc = _L().a2v_decoder_input(dt(x), ptr(x), ptr(noise), ptr(restore), ptr(out), N, K, D, T, *(_drop(drop)), stream())
    check(rc, "a2v_decoder_input")
    return out


def d2v_loss(x, y, mask_u8, cb, *, stats, want_dx=True, gscale=1.0, loss_rows=None):
    """x (N,T,D), y (B,T,D) fp32, mask (N,T) uint8.  stats: double[4D+2] accumulated."""
    N, T, D = x.shape
    dx = _EL(x) if want_dx else None
    rc = _L().a2v_d2v_loss(dt(x), ptr(x), ptr(y), ptr(mask_u8), ptr(dx), ptr(loss_rows), ptr(stats), N, T, D, cb,
                           D ** -0.5, gscale, stream())
    check(rc, "a2v_d2v_loss")
    return dx


def a_weight_table(fs: int, n_fft: int, min_db: float = -80.0):
    """Linear-power A-weighting per rfft bin, reference data2vec2.py:461-479 (host, numpy, once)."""
    import numpy as np
    freq = np.linspace(0, fs // 2, n_fft // 2 + 1)
    fsq = freq ** 2
    fsq[0] = 1.0
    w = 2.0 + 20.0 * (2 * np.log10(12194) + 2 * np.log10(fsq) - np.log10(fsq + 12194 ** 2) - np.log10(fsq + 20.6 ** 2)
                      - 0.5 * np.log10(fsq + 107.7 ** 2) - 0.5 * np.log10(fsq + 737.9 ** 2))
    return torch.from_numpy(np.power(10, np.maximum(w, min_db) / 10)).float()


class Mixup:
    """BC-learning source mixup on device (data2vec2.py:536-598): tables once, two launches per batch."""

    def __init__(self, sample_rate: int, window_s: float, device, min_db: float = -80.0):
        self.n_fft = round(sample_rate * window_s)
        self.hop = self.n_fft // 2
        self.window = torch.hann_window(self.n_fft).to(device)
        self.aw = a_weight_table(sample_rate, self.n_fft, min_db).to(device)
        self.floor = 10 ** (min_db / 10)

    def gain_db(self, x):
        B, L = x.shape
        nframes = (L - self.n_fft) // self.hop + 1
        g = _E(B, nframes, dtype=torch.float32, device=x.device)
        rc = _L().a2v_mixup_gain(ptr(x), ptr(self.window), ptr(self.aw), ptr(g), B, L, self.n_fft, self.hop, self.floor,
                                 stream())
        check(rc, "a2v_mixup_gain")
        return g

    def __call__(self, x, perm, r: float, mix_mask=None):
        """x (B,L) fp32, perm (B) int32 device tensor, r in (0,1] -> mixed (B,L); mix_mask (B) uint8: only those clips
        are mixed (mixup_prob < 1), None = all"""
        g = self.gain_db(x)
        out = _EL(x)
        assert mix_mask is None or (mix_mask.dtype == torch.uint8 and mix_mask.numel() == x.shape[0])
        rc = _L().a2v_mixup_apply(ptr(x), ptr(perm), ptr(g), ptr(out), x.shape[0], x.shape[1], g.shape[1], float(r),
                                  ptr(mix_mask), stream())
        check(rc, "a2v_mixup_apply")
        return out


def dropout_mask(n, p, seed, device="cuda"):
    """keep-mask (uint8) of linear indices [0,n) for a dropout site -- what every kernel regenerates."""
    keep = _E(n, dtype=torch.uint8, device=device)
    rc = _L().a2v_dropout_mask(ptr(keep), n, float(p), int(seed) & 0xFFFFFFFF, stream())
    check(rc, "a2v_dropout_mask")
    return keep


def dropout_mask2d(rows, cols, p, seed, device="cuda"):
    """keep-mask (rows, cols) uint8 of the attention-probability dropout site: rows = (n*H + h)*S + q, cols = key."""
    keep = _E(rows, cols, dtype=torch.uint8, device=device)
    rc = _L().a2v_dropout_mask2d(ptr(keep), rows, cols, float(p), int(seed) & 0xFFFFFFFF, stream())
    check(rc, "a2v_dropout_mask2d")
    return keep


def ema_update(ema, p, decay):
    rc = _L().a2v_ema_update(ptr(ema), ptr(p), ema.numel(), decay, stream())
    check(rc, "a2v_ema_update")


def adam_step(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, decay_mask=None,
              grad_scale_dev=None, lr_scale=None):
    """decay_mask (uint8) / lr_scale (fp32): one entry per 64 parameters of this slice (the slice must start on a
    64-parameter boundary of the buffer the tables were built for)."""
    rc = _L().a2v_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                            grad_scale, ptr(decay_mask), ptr(grad_scale_dev), ptr(lr_scale), stream())
    check(rc, "a2v_adam_step")


def sumsq(x, out):
    rc = _L().a2v_sumsq(ptr(x), x.numel(), ptr(out), stream())
    check(rc, "a2v_sumsq")


def pack_weight(w, out, *, G, N, Cg, k, Cgp=None, Np=None, mode=0):
    """w: fp32 param (G*N, Cg, k) (or (N,K) Linear with k=1) -> packed operand in out's dtype."""
    Cgp = Cg if Cgp is None else Cgp
    Np = N if Np is None else Np
    rc = _L().a2v_pack_weight(dt(out), ptr(w), ptr(out), G, N, Cg, k, Cgp, Np, mode, stream())
    check(rc, "a2v_pack_weight")
    return out


def pack_weight_batch(desc: torch.Tensor, dtype: torch.dtype):
    """desc: (n, 10) int64 device tensor of (src, dst, G, N, Cg, k, Cgp, Np, mode, total) -- see a2v.h."""
    code = BF16 if dtype in (torch.bfloat16, torch.float16) else F32
    rc = _L().a2v_pack_weight_batch(code, ptr(desc), desc.shape[0], stream())
    check(rc, "a2v_pack_weight_batch")


def unpack_wgrad(packed, gw, *, GN, Cg, k, Cgp):
    rc = _L().a2v_unpack_wgrad(ptr(packed), ptr(gw), GN, Cg, k, Cgp, stream())
    check(rc, "a2v_unpack_wgrad")


def col2im(dA, *, B, T_in, T_out, C_, k, s, p, out=None):
    """out (B,T_in,>=C_) may be wider than C_ (extra channels untouched)."""
    dX = out if out is not None else _E(B, T_in, C_, dtype=dA.dtype, device=dA.device)
    rc = _L().a2v_col2im(dt(dA), ptr(dA), ptr(dX), B, T_in, T_out, C_, k, s, p, dA.stride(-2), dX.stride(-2), stream())
    check(rc, "a2v_col2im")
    return dX


def cast2d(src, dst):
    """dst[m, :C] = src[m, :C] with dtype conversion; 2-D views with unit inner stride."""
    M, C_ = src.shape
    rc = _L().a2v_cast2d(dt(src), dt(dst), ptr(src), ptr(dst), M, C_, src.stride(0), dst.stride(0), stream())
    check(rc, "a2v_cast2d")
    return dst


def add_drop(a, b, c=None, *, drop=None, out=None):
    """out = a + dropout(b + c); a, c optional; drop = (p, seed) or None."""
    assert b.is_contiguous() and (a is None or a.is_contiguous()) and (c is None or c.is_contiguous())
    out = _EL(b) if out is None else out
    dp, ds = (drop if drop is not None else (0.0, 0))
    rc = _L().a2v_add_drop(dt(b), ptr(a), ptr(b), ptr(c), ptr(out), b.numel(), float(dp), int(ds) & 0xFFFFFFFF, stream())
    check(rc, "a2v_add_drop")
    return out


def mask_frames(x, mask_t=None, mask_c=None, *, noise_std=0.0, seed=0, backward=False, out=None):
    """apply_mask (base.py:457-485): x (B,T,D); mask_t (B,T) uint8, mask_c (B,D) uint8; see a2v.h."""
    B, T, D = x.shape
    assert x.is_contiguous() and (mask_t is None or (mask_t.dtype == torch.uint8 and mask_t.is_contiguous()))
    assert mask_c is None or (mask_c.dtype == torch.uint8 and mask_c.is_contiguous() and mask_c.shape == (B, D))
    out = _EL(x) if out is None else out
    rc = _L().a2v_mask_frames(dt(x), ptr(x), ptr(mask_t), ptr(mask_c), ptr(out), B, T, D, float(noise_std),
                              int(seed) & 0xFFFFFFFF, 1 if backward else 0, stream())
    check(rc, "a2v_mask_frames")
    return out


# ----------------------------------------------------------------------------------------------
#                                  caller-owned scratch (include/a2v.h, ABI 10)
# ----------------------------------------------------------------------------------------------
_WORKSPACES = {}          # (device index, stream handle) -> [zeroed uint8 tensor, users]


def bind_workspace(stream_=None):
    """Gives the library a caller-owned workspace for `stream_` (default: the current stream): a2v_query_workspace() bytes of
    zeroed device memory, bound with a2v_bind_workspace -- the library then keeps no scratch of its own for that stream and
    none of its four static sets is taken.  Counted per stream: every bind_workspace needs one release_workspace."""
    st = stream_ if stream_ is not None else torch.cuda.current_stream()
    key = (st.device.index, st.cuda_stream)
    ent = _WORKSPACES.get(key)
    if ent is None:
        lib = _L()
        n = int(lib.a2v_query_workspace())
        with torch.cuda.stream(st):                      # the zero fill is ordered before every later call on that stream
            t = torch.zeros(n + 256, dtype=torch.uint8, device=st.device)
        off = (-t.data_ptr()) % 256
        with torch.cuda.device(st.device):
            check(lib.a2v_bind_workspace(st.cuda_stream, t.data_ptr() + off, n), "a2v_bind_workspace")
        ent = _WORKSPACES[key] = [t, 0]
    ent[1] += 1
    return key


def release_workspace(key) -> None:
    ent = _WORKSPACES.get(key)
    if ent is None:
        return
    ent[1] -= 1
    if ent[1] <= 0:
        del _WORKSPACES[key]
        dev, handle = key
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)                  # kernels queued on the stream may still use the workspace
            _L().a2v_bind_workspace(handle, None, 0)


def debug_set(key: int, value: int):
    check(_L().a2v_debug_set(int(key), int(value)), "a2v_debug_set")
