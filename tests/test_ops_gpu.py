"""GPU parity of every C-ABI op against plain fp32 PyTorch-on-CPU references (and oracle functions).

Tolerances (relative L2, ||a-b|| / ||b||):
  fp32 mode  : 2e-5  (exact-fp32 MFMA; only summation order differs)
  bf16 mode  : 1e-2  (inputs are pre-rounded to bf16, so the error is output rounding 2^-9 + fp32 accumulate)
"""
import math

import pytest
import torch
import torch.nn.functional as F

import a2v_oracle as O
from conftest import rel_err

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype, scale=1.0):
    return (2e-5 if dtype == torch.float32 else 1e-2) * scale


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from animal2vec_amd import ops as o
    return o


def rnd(*shape, dtype, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    return x.to(dtype).float()      # value representable in dtype, held as fp32 on CPU


def dev(x, dtype):
    return x.to(device="cuda", dtype=dtype).contiguous()


# ------------------------------------------------------------------------------------------ linear
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 136), (77, 48, 72), (1000, 384, 1024), (64, 3072, 1024)])
def test_linear_plain(ops, dtype, M, N, K):
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, seed=3)
    y = ops.linear(dev(x, dtype), dev(w, dtype), dev(b, torch.float32))
    assert rel_err(y.float(), x @ w.t() + b) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_epilogues(ops, dtype):
    M, N, K = 200, 160, 96
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, seed=3)
    r = rnd(M, N, dtype=dtype, seed=4)
    xd, wd, bd, rd = dev(x, dtype), dev(w, dtype), dev(b, torch.float32), dev(r, dtype)
    pre = x @ w.t() + b
    # bias + gelu + stored pre-activation
    pre_out = torch.empty(M, N, dtype=dtype, device="cuda")
    y = ops.linear(xd, wd, bd, gelu=True, pre_out=pre_out)
    assert rel_err(pre_out.float(), pre) < tol(dtype)
    assert rel_err(y.float(), F.gelu(pre)) < tol(dtype)
    # bias + residual
    y = ops.linear(xd, wd, bd, res=rd)
    assert rel_err(y.float(), pre + r) < tol(dtype)
    # gelu backward epilogue + column sums
    cs = torch.zeros(N, dtype=torch.float32, device="cuda")
    y = ops.linear(xd, wd, None, gelu_bwd_aux=rd, colsum=cs)
    rr = r.clone().requires_grad_(True)
    F.gelu(rr).backward(x @ w.t())
    assert rel_err(y.float(), rr.grad) < tol(dtype)
    assert rel_err(cs, y.float().sum(0)) < 1e-4


@pytest.mark.parametrize("M,N,K", [(13440, 1024, 256), (16500, 2048, 256), (3000, 256, 512)])
def test_linear_p8_tile_handout_modes(ops, M, N, K):
    """Persistent NT kernel: static tile list (0), dynamic after the first tile (1), every tile handed out
    (2, default: robust when collectives hold CUs): identical results, several launches back to back (counter reset)."""
    from animal2vec_amd import _lib
    dtype = torch.bfloat16
    xd, wd = dev(rnd(M, K, dtype=dtype, seed=1), dtype), dev(rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5), dtype)
    bd = dev(rnd(N, dtype=torch.float32, seed=3), torch.float32)
    outs = {}
    try:
        for mode in (0, 1, 2):
            _lib.load().a2v_debug_set(7, mode)
            ys = [ops.linear(xd, wd, bd, gelu=True) for _ in range(4)]
            torch.cuda.synchronize()
            assert all(torch.equal(ys[0], y) for y in ys[1:]), mode
            outs[mode] = ys[0]
    finally:
        _lib.load().a2v_debug_set(7, 2)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_err(outs[2].float(), F.gelu(xd.float() @ wd.float().t() + bd)) < tol(dtype)


@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("ih", [0, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(1100, 512, 256), (1024, 256, 384), (2049, 768, 1024), (13440 + 100, 1024, 512)])
def test_linear_p8_kernel(ops, M, N, K, ih, direct):
    """Shapes eligible for the 256x256 8-phase bf16 kernel (N % 256 == 0, K % 128 == 0, M >= 1024): M tails,
    every epilogue, against fp32 torch and against the 128x128 kernel (a2v_debug_set(1, 0)); direct = 1: the
    direct-store epilogue (swapped MFMA operands, permuted B rows, shifted last m-tile), 0: the LDS-staged one."""
    from animal2vec_amd import _lib
    dtype = torch.bfloat16
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, seed=3)
    r = rnd(M, N, dtype=dtype, seed=4)
    xd, wd, bd, rd = dev(x, dtype), dev(w, dtype), dev(b, torch.float32), dev(r, dtype)
    pre = x @ w.t() + b

    def run_all():
        out = {}
        out["plain"] = ops.linear(xd, wd, bd)
        pre_out = torch.empty(M, N, dtype=dtype, device="cuda")
        out["gelu"] = ops.linear(xd, wd, bd, gelu=True, pre_out=pre_out)
        out["pre"] = pre_out
        out["res"] = ops.linear(xd, wd, bd, res=rd)
        cs = torch.zeros(N, dtype=torch.float32, device="cuda")
        out["gbwd"] = ops.linear(xd, wd, None, gelu_bwd_aux=rd, colsum=cs)
        out["cs"] = cs
        out["drop"] = ops.linear(xd, wd, bd, res=rd, drop=(0.25, 99))
        # the remaining flag sets that have fixed-flavour instantiations of the 256- / 224-row tiles
        out["none"] = ops.linear(xd, wd, None)
        out["res_nobias"] = ops.linear(xd, wd, None, res=rd)
        out["gelu_only"] = ops.linear(xd, wd, bd, gelu=True)
        keep = torch.empty(M, N, dtype=dtype, device="cuda")
        out["gelu_dact"] = ops.linear(xd, wd, bd, gelu=True, pre_out=keep, dact=True)
        out["dact"] = keep
        cs2 = torch.zeros(N, dtype=torch.float32, device="cuda")
        out["gbwd_dact"] = ops.linear(xd, wd, None, gelu_bwd_aux=rd, colsum=cs2, dact=True)
        out["cs2"] = cs2
        return out

    _lib.load().a2v_debug_set(5, ih)          # tile height 192 / 224 / 256 rows, 0 = the launcher's choice
    _lib.load().a2v_debug_set(10, direct)
    try:
        new = run_all()
        again = run_all()                     # several multi-tile launches back to back (tile counters, scratch reuse)
    finally:
        _lib.load().a2v_debug_set(5, 0)
        _lib.load().a2v_debug_set(10, 1)
    for k in new:
        assert torch.equal(new[k], again[k]) or (k in ("cs", "cs2") and rel_err(new[k], again[k]) < 1e-5), k
    _lib.load().a2v_debug_set(1, 0)
    try:
        old = run_all()
    finally:
        _lib.load().a2v_debug_set(1, 1)
    assert rel_err(new["plain"].float(), pre) < tol(dtype)
    assert rel_err(new["pre"].float(), pre) < tol(dtype)
    assert rel_err(new["gelu"].float(), F.gelu(pre)) < tol(dtype)
    assert rel_err(new["res"].float(), pre + r) < tol(dtype)
    assert rel_err(new["none"].float(), pre - b) < tol(dtype)
    assert rel_err(new["res_nobias"].float(), pre - b + r) < tol(dtype)
    assert rel_err(new["gelu_only"].float(), F.gelu(pre)) < tol(dtype)
    assert torch.equal(new["gelu_dact"], new["gelu_only"])                 # one evaluation or two: the same gelu
    pg = pre.clone().requires_grad_(True)
    F.gelu(pg).sum().backward()
    assert rel_err(new["dact"].float(), pg.grad) < tol(dtype)
    assert rel_err(new["gbwd_dact"].float(), (pre - b) * r.float()) < tol(dtype)
    assert rel_err(new["cs2"], new["gbwd_dact"].float().sum(0)) < 5e-3
    for k in new:          # same arithmetic order per output element (K accumulated in one chain): bit-identical
        assert torch.equal(new[k], old[k]) or rel_err(new[k].float(), old[k].float()) < 1e-5, k


@pytest.mark.parametrize("K", [256, 1024])
def test_linear_p8_fixed_flavours_multi_round(ops, K):
    """the fixed-flavour instantiations' tile seam lets the first K-tile of a tile wait with vmcnt(6 + E), E = the vector-memory
    operations every wave certainly issues in the epilogue before it -- a hand-derived count.  If it were too high, K-tile 1's LDS-DMA
    could still be in flight when its buffer is read: wrong products from the SECOND tile of a workgroup on, silently.  So every
    fixed flavour runs a launch of more than two rounds of tiles (561 tiles > 2 x 256 CUs; K = 256: the shortest K loop, two K-tiles
    between seams; K = 1024: the step's) against the generic kernel (a2v_debug_set(20, 0)), bit for bit.  tests/test_abi.py counts
    the operations in the ISA."""
    dtype = torch.bfloat16
    M, N = 256 * 140 + 77, 1024
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b, r = rnd(N, dtype=torch.float32, seed=3), rnd(M, N, dtype=dtype, seed=4)
    xd, wd, bd, rd = dev(x, dtype), dev(w, dtype), dev(b, torch.float32), dev(r, dtype)

    def run_all():
        out = {}
        out["bias"] = ops.linear(xd, wd, bd)
        out["none"] = ops.linear(xd, wd, None)
        out["bias_res"] = ops.linear(xd, wd, bd, res=rd)
        out["res"] = ops.linear(xd, wd, None, res=rd)
        out["bias_res_drop"] = ops.linear(xd, wd, bd, res=rd, drop=(0.1, 99))
        out["bias_gelu"] = ops.linear(xd, wd, bd, gelu=True)
        keep = torch.empty(M, N, dtype=dtype, device="cuda")
        out["fc1"] = ops.linear(xd, wd, bd, gelu=True, pre_out=keep, dact=True)
        out["fc1_keep"] = keep
        cs = torch.zeros(N, dtype=torch.float32, device="cuda")
        out["fc2d"] = ops.linear(xd, wd, None, gelu_bwd_aux=rd, colsum=cs, dact=True)
        out["fc2d_cs"] = cs
        return out

    fixed = run_all()
    again = run_all()
    ops.debug_set(20, 0)
    try:
        generic = run_all()
    finally:
        ops.debug_set(20, 1)
    ref = x @ w.t()
    assert rel_err(fixed["none"].float(), ref) < tol(dtype)
    for k in fixed:
        if k.endswith("_cs"):
            assert rel_err(fixed[k], generic[k]) < 1e-5 and rel_err(fixed[k], again[k]) < 1e-5, k
        else:
            assert torch.equal(fixed[k], generic[k]), (k, rel_err(fixed[k].float(), generic[k].float()))
            assert torch.equal(fixed[k], again[k]), k


@pytest.mark.parametrize("M,N,K", [(256 * 140 + 77, 1024, 256), (13440 + 100, 1024, 512), (2049, 768, 1024), (300, 200, 136)])
def test_linear_dact8(ops, M, N, K):
    """A2V_EPI_DACT8 (include/a2v.h, ABI 11): the Mlp's kept gelu' in 8 bits.  fc1 form: the activation is bit-identical to the bf16
    form's, the byte tensor decodes to gelu'(pre) within half a grid step (1/400) + the rounding of pre; fc2-input-gradient form:
    (dy W^T) * decode(bytes) and its column sums.  Shapes: more than two rounds of tiles of the persistent kernel's fixed
    flavours (561 tiles), one round with an M tail (the tail rows go to the 128 x 128 kernel), N = 768 and a ragged shape
    (128 x 128 kernels only); every one also against the 128 x 128 kernel alone (a2v_debug_set(1, 0)), bit for bit."""
    dtype = torch.bfloat16
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, seed=3)
    dy, w2 = rnd(M, K, dtype=dtype, seed=4), rnd(N, K, dtype=dtype, seed=5, scale=K ** -0.5)
    xd, wd, bd, dyd, w2d = dev(x, dtype), dev(w, dtype), dev(b, torch.float32), dev(dy, dtype), dev(w2, dtype)

    def run():
        keep16 = torch.empty(M, N, dtype=dtype, device="cuda")
        h16 = ops.linear(xd, wd, bd, gelu=True, pre_out=keep16, dact=True)
        keep8 = torch.empty(M, N, dtype=torch.uint8, device="cuda")
        h8 = ops.linear(xd, wd, bd, gelu=True, pre_out=keep8, dact=True)
        cs = torch.zeros(N, dtype=torch.float32, device="cuda")
        dh = ops.linear(dyd, w2d, None, gelu_bwd_aux=keep8, colsum=cs, dact=True)
        return h16, keep16, h8, keep8, dh, cs

    h16, keep16, h8, keep8, dh, cs = run()
    assert torch.equal(h8, h16)
    dec = (keep8.float() - 26.0) / 200.0
    pre = (x @ w.t() + b).requires_grad_(True)
    F.gelu(pre).sum().backward()
    assert float((dec.cpu() - pre.grad).abs().max()) < 0.0025 + 0.008        # half a grid step + gelu' of a pre-activation rounded in bf16 products
    assert float((dec - keep16.float()).abs().max()) < 0.0025 + 0.005          # against the bf16 form of the same derivative
    assert int(keep8.max()) <= 252 and int(keep8.min()) >= 0
    ref = (dy @ w2.t()) * dec.cpu()
    assert rel_err(dh.float(), ref) < tol(dtype)
    assert rel_err(cs, dh.float().sum(0)) < 5e-3
    ops.debug_set(1, 0)
    try:
        o = run()
    finally:
        ops.debug_set(1, 1)
    assert torch.equal(o[2], h8) and torch.equal(o[3], keep8) and torch.equal(o[4], dh) and rel_err(o[5], cs) < 1e-5
    with pytest.raises(Exception):          # the byte tensor exists for the bf16 Linear pair only
        ops.linear(xd.float(), wd.float(), bd, gelu=True, pre_out=keep8, dact=True)


@pytest.mark.parametrize("M,N,K", [(2049, 768, 1024), (13440 + 100, 1024, 512), (1100, 512, 256)])
def test_linear_w2_experiment_kernel(ops, M, N, K):
    """a2v_gemm_w2.hip (round-4 experiment, off by default: a2v_debug_set(25, 2)): 256 x 128 tiles, two workgroups per CU, the
    256 x 256 kernel's direct-store epilogue with a half width of 64.  Every epilogue against the 128 x 128 kernel: same arithmetic
    order per output element, so bit-identical (column sums to 1e-5: their partials add in another order)."""
    dtype = torch.bfloat16
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, seed=3)
    r = rnd(M, N, dtype=dtype, seed=4)
    xd, wd, bd, rd = dev(x, dtype), dev(w, dtype), dev(b, torch.float32), dev(r, dtype)

    def run_all():
        out = {}
        out["plain"] = ops.linear(xd, wd, bd)
        out["none"] = ops.linear(xd, wd, None)
        out["res"] = ops.linear(xd, wd, bd, res=rd)
        out["drop"] = ops.linear(xd, wd, bd, res=rd, drop=(0.25, 99))
        keep = torch.empty(M, N, dtype=dtype, device="cuda")
        out["gelu_dact"] = ops.linear(xd, wd, bd, gelu=True, pre_out=keep, dact=True)
        out["dact"] = keep
        cs = torch.zeros(N, dtype=torch.float32, device="cuda")
        out["gbwd_dact"] = ops.linear(xd, wd, None, gelu_bwd_aux=rd, colsum=cs, dact=True)
        out["cs"] = cs
        return out

    try:
        ops.debug_set(25, 2)
    except Exception:
        pytest.skip("the product library does not carry the experiment kernel (python -m animal2vec_amd.build --variant w2 -DA2V_W2, "
                    "A2V_LIB_PATH=.../liba2v_hip_w2.so)")
    try:
        new = run_all()
    finally:
        ops.debug_set(25, 0)
    ops.debug_set(1, 0)
    try:
        old = run_all()
    finally:
        ops.debug_set(1, 1)
    assert rel_err(new["plain"].float(), x @ w.t() + b) < tol(dtype)
    for k in new:
        assert torch.equal(new[k], old[k]) or rel_err(new[k].float(), old[k].float()) < 1e-5, k


@pytest.mark.parametrize("M,N,K", [(2100, 2048, 1024), (4096, 1024, 2048), (2048 + 64 * 3, 2048, 1024)])
def test_linear_wgrad_p8_kernel(ops, M, N, K):
    """256x256 TN weight-gradient kernel (N, K % 256 == 0, >= 32 tiles): ragged M (rows past M come from the zero
    page), accumulation into a non-zero buffer, against fp32 torch and the 128x128 kernel (a2v_debug_set(4, 0))."""
    from animal2vec_amd import _lib
    dtype = torch.bfloat16
    x, dy = rnd(M, K, dtype=dtype, seed=1), rnd(M, N, dtype=dtype, seed=2)
    g0 = rnd(N, K, dtype=torch.float32, seed=3)
    ref = g0 + dy.t() @ x
    xd, dyd = dev(x, dtype), dev(dy, dtype)
    gw = g0.clone().cuda()
    ops.linear_wgrad(xd, dyd, gw)
    _lib.load().a2v_debug_set(4, 0)
    try:
        gw_old = g0.clone().cuda()
        ops.linear_wgrad(xd, dyd, gw_old)
    finally:
        _lib.load().a2v_debug_set(4, 1)
    assert rel_err(gw, ref) < tol(dtype)
    assert rel_err(gw, gw_old) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Na,Ka,Nb,Kb", [(2100, 1024, 256, 256, 1024), (4160, 768, 256, 256, 256), (300, 96, 64, 40, 72)])
def test_linear_wgrad_pair(ops, dtype, M, Na, Ka, Nb, Kb):
    """two weight gradients over the same rows in one call (one 256x256 TN launch when eligible, else two launches)"""
    xa, dya = rnd(M, Ka, dtype=dtype, seed=1), rnd(M, Na, dtype=dtype, seed=2)
    xb, dyb = rnd(M, Kb, dtype=dtype, seed=3), rnd(M, Nb, dtype=dtype, seed=4)
    ga0, gb0 = rnd(Na, Ka, dtype=torch.float32, seed=5), rnd(Nb, Kb, dtype=torch.float32, seed=6)
    ga, gb = ga0.clone().cuda(), gb0.clone().cuda()
    ops.linear_wgrad_pair(dev(xa, dtype), dev(dya, dtype), ga, dev(xb, dtype), dev(dyb, dtype), gb)
    assert rel_err(ga, ga0 + dya.t() @ xa) < tol(dtype)
    assert rel_err(gb, gb0 + dyb.t() @ xb) < tol(dtype)


# ------------------------------------------------------------------------------------------ convs
def to_cl(x_bct, cp, dtype):
    B, C_, T = x_bct.shape
    out = torch.zeros(B, T, cp)
    out[:, :, :C_] = x_bct.transpose(1, 2)
    return dev(out, dtype)


def pack_ref(w, G, cgp):
    """(Cout, Cg, k) -> (G, N, k*cgp) forward operand"""
    Cout, Cg, k = w.shape
    p = torch.zeros(Cout, k, cgp)
    p[:, :, :Cg] = w.permute(0, 2, 1)
    return p.reshape(G, Cout // G, k * cgp)


CONVS = [  # Cin, Cout, k, stride, pad_left, groups, T
    (127, 64, 10, 5, 3, 1, 400),
    (64, 96, 3, 2, 1, 1, 201),
    (64, 64, 3, 1, 1, 1, 150),
    (64, 64, 2, 1, 0, 1, 150),
    (128, 128, 19, 1, 9, 16, 200),
    (128, 192, 7, 1, 3, 8, 130),
    (192, 192, 7, 1, 3, 8, 130),
]


def conv_ref(x, w, k, s, pl, G, T_out):
    T = x.shape[-1]
    pr = max(0, (T_out - 1) * s + k - pl - T)
    return F.conv1d(F.pad(x, (pl, pr)), w, stride=s, groups=G)[..., :T_out]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd_and_wgrad(ops, dtype, cfg):
    Cin, Cout, k, s, pl, G, T = cfg
    B = 3
    Cg = Cin // G
    cgp = (Cg + 7) // 8 * 8
    cp = cgp * G if G > 1 else cgp
    x = rnd(B, Cin, T, dtype=dtype, seed=5)
    w = rnd(Cout, Cg, k, dtype=dtype, seed=6, scale=(Cg * k) ** -0.5)
    bias = rnd(Cout, dtype=torch.float32, seed=7)
    if s == 1:
        T_out = T
    else:
        T_out = (T + 2 * pl - k) // s + 1
    ref = conv_ref(x, w, k, s, pl, G, T_out) + bias.view(1, -1, 1)
    xcl = to_cl(x, cp, dtype)
    wp = dev(pack_ref(w, G, cgp), dtype)
    y = ops.conv1d_cl(xcl, wp, dev(bias, torch.float32), k=k, stride=s, pad=pl, groups=G, T_out=T_out, n_out=Cout)
    assert rel_err(y.float().transpose(1, 2), ref) < tol(dtype)
    # the device-side packer agrees with the reference packing
    wp2 = torch.empty_like(wp)
    ops.pack_weight(dev(w, torch.float32), wp2, G=G, N=Cout // G, Cg=Cg, k=k, Cgp=cgp, mode=0)
    assert torch.equal(wp2.float().cpu(), wp.float().cpu())
    # weight gradient
    dy = rnd(B, Cout, T_out, dtype=dtype, seed=8)
    wr = w.clone().requires_grad_(True)
    conv_ref(x, wr, k, s, pl, G, T_out).backward(dy)
    gw = torch.zeros(G, Cout // G, k * cgp, dtype=torch.float32, device="cuda")
    ops.conv1d_cl_wgrad(xcl, to_cl(dy, Cout, dtype), gw, k=k, stride=s, pad=pl, groups=G, T_out=T_out)
    gw_un = torch.zeros(Cout, Cg, k, dtype=torch.float32, device="cuda")
    ops.unpack_wgrad(gw, gw_un, GN=Cout, Cg=Cg, k=k, Cgp=cgp)
    assert rel_err(gw_un, wr.grad) < tol(dtype, 2)


@pytest.mark.parametrize("Cin,k,s,pl,T,B", [(128, 10, 5, 3, 8000, 3), (512, 3, 2, 1, 4000, 3), (128, 10, 5, 3, 80000, 2)])
def test_strided_feature_convs_on_the_8phase_gemms(ops, Cin, k, s, pl, T, B):
    """Layers 1-4 of the feature extractor (nn/utils.py:1085-1091: k 10 / s 5 / pad 3 on 128 channels, then k 3 / s 2 / pad 1 on
    512) as flat overlapping-row GEMMs on the persistent 8-phase kernels + the edge-frame kernels (a2v_conv_strided.hip): forward
    and weight gradient against torch's conv1d on the bf16-rounded operands, and against the implicit-im2col kernels they replace
    (a2v_debug_set(21, 0)) -- the rows at clip boundaries, where the flat view aliases the neighbouring clip instead of zero
    padding, are looked at separately."""
    from animal2vec_amd import _lib
    dtype, Cout = torch.bfloat16, 512
    x = rnd(B, Cin, T, dtype=dtype, seed=5)
    w = rnd(Cout, Cin, k, dtype=dtype, seed=6, scale=(Cin * k) ** -0.5)
    T_out = (T + 2 * pl - k) // s + 1
    assert T == T_out * s
    ref = conv_ref(x, w, k, s, pl, 1, T_out)
    xcl = to_cl(x, Cin, dtype)
    wp = dev(pack_ref(w, 1, Cin), dtype)
    dy = rnd(B, Cout, T_out, dtype=dtype, seed=8)
    dycl = to_cl(dy, Cout, dtype)
    wr = w.clone().requires_grad_(True)
    conv_ref(x, wr, k, s, pl, 1, T_out).backward(dy)

    def both():
        y = ops.conv1d_cl(xcl, wp, None, k=k, stride=s, pad=pl, groups=1, T_out=T_out, n_out=Cout)
        gw = torch.zeros(1, Cout, k * Cin, dtype=torch.float32, device="cuda")
        ops.conv1d_cl_wgrad(xcl, dycl, gw, k=k, stride=s, pad=pl, groups=1, T_out=T_out)
        gw_un = torch.zeros(Cout, Cin, k, dtype=torch.float32, device="cuda")
        ops.unpack_wgrad(gw, gw_un, GN=Cout, Cg=Cin, k=k, Cgp=Cin)
        return y.float().transpose(1, 2).cpu(), gw_un.cpu()

    y1, g1 = both()
    _lib.load().a2v_debug_set(21, 0)
    try:
        y0, g0 = both()
    finally:
        _lib.load().a2v_debug_set(21, 1)
    for y in (y1, y0):
        assert rel_err(y, ref) < tol(dtype)
    edge = [0, 1, T_out - 2, T_out - 1]                 # first / last output frames of every clip
    assert rel_err(y1[:, :, edge], ref[:, :, edge]) < tol(dtype)
    assert rel_err(y1, y0) < 4e-3                       # both bf16 outputs of fp32 sums in different orders
    assert rel_err(g1, wr.grad) < tol(dtype, 2) and rel_err(g0, wr.grad) < tol(dtype, 2)
    # the taps that hang over the clip edges see the aliasing most: their weight gradient alone
    assert rel_err(g1[:, :, :pl], wr.grad[:, :, :pl]) < tol(dtype, 2)
    assert rel_err(g1[:, :, k - 1:], wr.grad[:, :, k - 1:]) < tol(dtype, 2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_padded_taps(ops, dtype):
    """Cg = 48 real channels per group, taps zero-padded to 64 in the packed operand (fast bf16 path)."""
    B, T, G, k, Cg, N = 2, 300, 4, 7, 48, 48
    x = rnd(B, G * Cg, T, dtype=dtype, seed=1)
    w = rnd(G * N, Cg, k, dtype=dtype, seed=2, scale=0.05)
    ref = F.conv1d(x, w, padding=k // 2, groups=G)
    xcl = to_cl(x, G * Cg, dtype)
    wp = dev(pack_ref(w, G, 64), dtype)
    y = ops.conv1d_cl(xcl, wp, None, k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=G * N, cv=Cg)
    assert rel_err(y.float().transpose(1, 2), ref) < tol(dtype)
    dy = rnd(B, G * N, T, dtype=dtype, seed=3)
    wr = w.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    F.conv1d(xr, wr, padding=k // 2, groups=G).backward(dy)
    gw = torch.zeros(G, N, k * 64, dtype=torch.float32, device="cuda")
    dycl = to_cl(dy, G * N, dtype)
    ops.conv1d_cl_wgrad(xcl, dycl, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, cv=Cg)
    gw_un = torch.zeros(G * N, Cg, k, dtype=torch.float32, device="cuda")
    ops.unpack_wgrad(gw, gw_un, GN=G * N, Cg=Cg, k=k, Cgp=64)
    assert rel_err(gw_un, wr.grad) < tol(dtype, 2)
    wt = torch.empty(G, Cg, k * 64, dtype=dtype, device="cuda")
    ops.pack_weight(dev(w, torch.float32), wt, G=G, N=N, Cg=Cg, k=k, Np=64, mode=1)
    dx = ops.conv1d_cl(dycl, wt, None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=G * Cg, cv=N)
    assert rel_err(dx.float().transpose(1, 2), xr.grad) < tol(dtype, 2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_clone_mask_view(ops, dtype):
    """x.repeat_interleave(cb) * (1 - mask) -> grouped conv, without materialising the clones"""
    B, cb, T, D, G, k = 2, 3, 96, 64, 4, 19
    x = rnd(B, T, D, dtype=dtype, seed=1)
    w = rnd(D, D // G, k, dtype=dtype, seed=2, scale=0.05)
    mask = (torch.rand(B * cb, T, generator=torch.Generator().manual_seed(3)) < 0.8).to(torch.uint8)
    xm = x.repeat_interleave(cb, 0) * (1 - mask.float().unsqueeze(-1))
    ref = F.conv1d(xm.transpose(1, 2), w, padding=k // 2, groups=G).transpose(1, 2)
    wp = dev(pack_ref(w, G, D // G), dtype)
    y = ops.conv1d_cl(dev(x, dtype), wp, None, k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D,
                      rowmask=mask.cuda(), bdiv=cb, n_batch=B * cb)
    assert rel_err(y.float(), ref) < tol(dtype)


def test_conv_slab_kernels_masked_long(ops):
    """bf16 slab kernels (a2v_conv_slab.hip / a2v_wgrad_slab.hip): k19 grouped conv with 64 channels per group over
    clips longer than one 1024-row block, on the never-materialised clone * (1 - mask) view; forward, input gradient
    and weight gradient against fp32 torch and against the implicit-GEMM kernels (a2v_debug_set 2 / 3)."""
    from animal2vec_amd import _lib
    dtype = torch.bfloat16
    B, cb, T, D, G, k = 2, 2, 1100, 128, 2, 19
    Cg = D // G
    x = rnd(B, T, D, dtype=dtype, seed=1)
    w = rnd(D, Cg, k, dtype=dtype, seed=2, scale=(Cg * k) ** -0.5)
    bias = rnd(D, dtype=torch.float32, seed=4)
    mask = (torch.rand(B * cb, T, generator=torch.Generator().manual_seed(3)) < 0.7).to(torch.uint8)
    xm = (x.repeat_interleave(cb, 0) * (1 - mask.float().unsqueeze(-1))).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv1d(xm.transpose(1, 2), wr, bias, padding=k // 2, groups=G).transpose(1, 2)
    dy = rnd(B * cb, T, D, dtype=dtype, seed=5)
    ref.backward(dy)
    xd, dyd, md = dev(x, dtype), dev(dy, dtype), mask.cuda()
    wp = dev(pack_ref(w, G, Cg), dtype)
    wt = torch.empty(G, Cg, k * Cg, dtype=dtype, device="cuda")
    ops.pack_weight(dev(w, torch.float32), wt, G=G, N=Cg, Cg=Cg, k=k, Np=Cg, mode=1)

    def run():
        y = ops.conv1d_cl(xd, wp, dev(bias, torch.float32), k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D,
                          rowmask=md, bdiv=cb, n_batch=B * cb)
        dx = ops.conv1d_cl(dyd, wt, None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D)
        gw = torch.zeros(G, Cg, k * Cg, dtype=torch.float32, device="cuda")
        ops.conv1d_cl_wgrad(xd, dyd, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, rowmask=md, bdiv=cb, n_batch=B * cb)
        return y, dx, gw

    y, dx, gw = run()
    lib = _lib.load()
    lib.a2v_debug_set(2, 0); lib.a2v_debug_set(3, 0); lib.a2v_debug_set(37, 0)
    try:
        y0, dx0, gw0 = run()
    finally:
        lib.a2v_debug_set(2, 1); lib.a2v_debug_set(3, 1); lib.a2v_debug_set(37, 1)
    assert rel_err(y.float(), ref.detach()) < tol(dtype)
    assert rel_err(dx.float(), xm.grad) < tol(dtype, 2)
    assert rel_err(gw, gw0) < 1e-4                              # fp32 atomics in a different order
    gw_un = torch.zeros(D, Cg, k, dtype=torch.float32, device="cuda")
    ops.unpack_wgrad(gw, gw_un, GN=D, Cg=Cg, k=k, Cgp=Cg)
    assert rel_err(gw_un, wr.grad) < tol(dtype, 2)
    assert float(gw.abs().max()) == 0.0                         # unpack leaves the packed accumulator clear
    assert torch.equal(y, y0) and torch.equal(dx, dx0)        # same per-element accumulation order


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D,G", [(128, 2), (64, 4)])
def test_conv_row_gather(ops, dtype, D, G):
    """a2v_conv_gemm_rows / a2v_conv_wgrad_rows: only a subset of output rows (the kept frames) is evaluated, on the
    clone * (1 - mask) view; equals the dense result gathered at those rows / fed with dY scattered to them."""
    B, cb, T, k = 2, 3, 300, 19
    Cg = D // G
    x = rnd(B, T, D, dtype=dtype, seed=1)
    w = rnd(D, Cg, k, dtype=dtype, seed=2, scale=(Cg * k) ** -0.5)
    bias = rnd(D, dtype=torch.float32, seed=4)
    gen = torch.Generator().manual_seed(3)
    mask = (torch.rand(B * cb, T, generator=gen) < 0.7).to(torch.uint8)
    keep = torch.stack([torch.randperm(T, generator=gen)[:40].sort().values for _ in range(B * cb)])      # (N, 40)
    rowidx = (torch.arange(B * cb).unsqueeze(1) * T + keep).reshape(-1).to(torch.int32)
    xm = (x.repeat_interleave(cb, 0) * (1 - mask.float().unsqueeze(-1)))
    wr = w.clone().requires_grad_(True)
    ref = F.conv1d(xm.transpose(1, 2), wr, bias, padding=k // 2, groups=G).transpose(1, 2).reshape(B * cb * T, D)
    ref_rows = ref[rowidx.long()]
    dyk = rnd(rowidx.numel(), D, dtype=dtype, seed=5)
    ref_rows.backward(dyk)
    xd, md, ri = dev(x, dtype), mask.cuda(), rowidx.cuda()
    cgp = Cg if Cg % 64 == 0 or dtype == torch.float32 else (Cg + 7) // 8 * 8
    wp = dev(pack_ref(w, G, cgp), dtype)
    y = ops.conv1d_cl(xd, wp, dev(bias, torch.float32), k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D,
                      rowmask=md, bdiv=cb, n_batch=B * cb, rowidx=ri, cv=Cg)
    assert y.shape == (rowidx.numel(), D)
    assert rel_err(y.float(), ref_rows.detach()) < tol(dtype)
    gw = torch.zeros(G, D // G, k * cgp, dtype=torch.float32, device="cuda")
    ops.conv1d_cl_wgrad(xd, dev(dyk, dtype), gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, rowmask=md, bdiv=cb,
                        n_batch=B * cb, rowidx=ri, cv=Cg)
    gw_un = torch.zeros(D, Cg, k, dtype=torch.float32, device="cuda")
    ops.unpack_wgrad(gw, gw_un, GN=D, Cg=Cg, k=k, Cgp=cgp)
    assert rel_err(gw_un, wr.grad) < tol(dtype, 2)


@pytest.mark.parametrize("T,keep_frac,k", [(700, 0.08, 19), (2000, 0.076, 19), (300, 0.6, 19), (128, 0.3, 7), (131, 1.0, 19)])
def test_conv_wgrad_sparse_rows(ops, T, keep_frac, k):
    """a2v_wgrad_sparse.hip (round 5): the positional encoder's weight gradients where one operand is row-sparse.
    mode A -- a2v_conv_wgrad with a row mask (first layer: X = clone * (1 - mask) is zero at the masked frames),
    mode B -- a2v_conv_wgrad_kept (last layer: dY only exists at the kept frames, ascending row list),
    both against fp32 autograd of the dense convolution and against the kernels they replace (a2v_debug_set(37, 0): the masked slab
    kernel / the gathered TN kernel); a row list that is NOT ascending must take the gathered kernel (checked on the device) and
    still be right.  Ragged last window (T = 700, 300, 131), every frame kept (131), a window without kept frames (T = 2000 clone 0)."""
    dtype = torch.bfloat16
    B, cb, D, G = 2, 3, 128, 2
    Cg = D // G
    N = B * cb
    gen = torch.Generator().manual_seed(7)
    x = rnd(B, T, D, dtype=dtype, seed=1)
    w = rnd(D, Cg, k, dtype=dtype, seed=2, scale=(Cg * k) ** -0.5)
    K = max(1, int(round(T * keep_frac)))
    keep = torch.stack([torch.randperm(T, generator=gen)[:K].sort().values for _ in range(N)])                     # (N, K) ascending
    if T == 2000:
        keep[0] = torch.cat([torch.arange(0, K // 2), torch.arange(1000, 1000 + K - K // 2)])                       # clone 0: empty windows
    mask = torch.ones(N, T, dtype=torch.uint8)
    mask.scatter_(1, keep, 0)
    rowidx = (torch.arange(N).unsqueeze(1) * T + keep).reshape(-1).to(torch.int32)
    xm = x.repeat_interleave(cb, 0) * (1 - mask.float().unsqueeze(-1))
    xd, md, ri = dev(x, dtype), mask.cuda(), rowidx.cuda()

    def unpacked(gw):
        un = torch.zeros(D, Cg, k, dtype=torch.float32, device="cuda")
        ops.unpack_wgrad(gw.clone(), un, GN=D, Cg=Cg, k=k, Cgp=Cg)
        return un

    # ---- mode A: dense dY, masked X
    wr = w.clone().requires_grad_(True)
    dy = rnd(N, T, D, dtype=dtype, seed=5)
    F.conv1d(xm.transpose(1, 2), wr, None, padding=k // 2, groups=G).transpose(1, 2).backward(dy)
    res = {}
    for sw in (1, 0):
        ops.debug_set(37, sw)
        gw = torch.zeros(G, Cg, k * Cg, dtype=torch.float32, device="cuda")
        ops.conv1d_cl_wgrad(xd, dev(dy, dtype), gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, rowmask=md, bdiv=cb, n_batch=N)
        res[sw] = gw
    ops.debug_set(37, 1)
    assert rel_err(unpacked(res[1]), wr.grad) < tol(dtype, 2)
    assert rel_err(res[1], res[0]) < 1e-4                                  # the same bf16 products, fp32 sums in another order

    # ---- mode B: compact dY at the kept frames, dense X (no mask: a later layer's input)
    wr = w.clone().requires_grad_(True)
    xfull = rnd(N, T, D, dtype=dtype, seed=8)
    ref = F.conv1d(xfull.transpose(1, 2), wr, None, padding=k // 2, groups=G).transpose(1, 2).reshape(N * T, D)
    dyk = rnd(rowidx.numel(), D, dtype=dtype, seed=9)
    ref[rowidx.long()].backward(dyk)
    xf = dev(xfull, dtype)
    perm = torch.randperm(rowidx.numel(), generator=gen)
    cases = {"kept": (ri, dev(dyk, dtype), True), "rows": (ri, dev(dyk, dtype), False),
             "not ascending": (ri[perm.cuda()].contiguous(), dev(dyk[perm], dtype), True)}
    out = {}
    for name, (rr, dd, kept) in cases.items():
        gw = torch.zeros(G, Cg, k * Cg, dtype=torch.float32, device="cuda")
        ops.conv1d_cl_wgrad(xf, dd, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, rowidx=rr, rows_kept=kept)
        out[name] = gw
        assert rel_err(unpacked(gw), wr.grad) < tol(dtype, 2), name
    assert rel_err(out["kept"], out["rows"]) < 1e-4
    assert rel_err(out["not ascending"], out["rows"]) < 1e-4


@pytest.mark.parametrize("T,keep_frac", [(700, 0.08), (2000, 0.076), (300, 0.6), (160, 1.0), (95, 0.3), (1500, 0.45)])
def test_conv_sparse_in(ops, T, keep_frac):
    """a2v_conv_sparse.hip (round 5): the positional encoder's two products whose INPUT is row-sparse and whose output is dense --
    the first layer's forward on clone * (1 - mask) (rows from the mask bytes, bias) and the last layer's input gradient from the
    compact output gradient (rows from the row list, the transposed / flipped pack, no bias) -- against fp32 torch and against the
    dense kernels they replace (the zero-filled input through conv1d_cl).  Windows with more than 64 kept rows (several chunks),
    a ragged last window, a window without kept rows, a row list that is not ascending (any order is taken)."""
    dtype = torch.bfloat16
    B, cb, D, G, k = 2, 3, 128, 2, 19
    Cg = D // G
    N = B * cb
    gen = torch.Generator().manual_seed(11)
    x = rnd(B, T, D, dtype=dtype, seed=1)
    w = rnd(D, Cg, k, dtype=dtype, seed=2, scale=(Cg * k) ** -0.5)
    bias = rnd(D, dtype=torch.float32, seed=4)
    K = max(1, int(round(T * keep_frac)))
    keep = torch.stack([torch.randperm(T, generator=gen)[:K].sort().values for _ in range(N)])
    if T == 2000:
        keep[0] = torch.cat([torch.arange(0, K // 2), torch.arange(1000, 1000 + K - K // 2)])
    mask = torch.ones(N, T, dtype=torch.uint8)
    mask.scatter_(1, keep, 0)
    xd, md = dev(x, dtype), mask.cuda()
    wp = dev(pack_ref(w, G, Cg), dtype)
    # ---- forward of the first layer
    xm = x.repeat_interleave(cb, 0) * (1 - mask.float().unsqueeze(-1))
    ref = F.conv1d(xm.transpose(1, 2), w.float(), bias, padding=k // 2, groups=G).transpose(1, 2)
    assert ops.conv_sparse_ok(xd, wp, k=k, groups=G, n_out=D)
    y = ops.conv1d_cl_sparse_in(xd, wp, dev(bias, torch.float32), k=k, pad=k // 2, groups=G, T=T, n_out=D, n_seq=N, rowmask=md, bdiv=cb)
    y0 = ops.conv1d_cl(xd, wp, dev(bias, torch.float32), k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D, rowmask=md, bdiv=cb, n_batch=N)
    assert rel_err(y.float(), ref) < tol(dtype)
    assert rel_err(y.float(), y0.float()) < 3e-3                         # bf16 roundings of two summation orders
    y_again = ops.conv1d_cl_sparse_in(xd, wp, dev(bias, torch.float32), k=k, pad=k // 2, groups=G, T=T, n_out=D, n_seq=N, rowmask=md, bdiv=cb)
    assert torch.equal(y, y_again)                                       # a tile column is summed by one wave in program order
    # ---- input gradient of the last layer: dY at the kept frames only
    wt = torch.empty(G, Cg, k * Cg, dtype=dtype, device="cuda")
    ops.pack_weight(dev(w, torch.float32), wt, G=G, N=Cg, Cg=Cg, k=k, mode=1)
    dyk = rnd(N * K, D, dtype=dtype, seed=5)
    rowidx = (torch.arange(N).unsqueeze(1) * T + keep).reshape(-1).to(torch.int32)
    dyd = torch.zeros(N * T, D)
    dyd[rowidx.long()] = dyk.float()
    xin = torch.zeros(N, D, T, requires_grad=True)
    F.conv1d(xin, w.float(), None, padding=k // 2, groups=G).backward(dyd.view(N, T, D).transpose(1, 2))
    ref_dx = xin.grad.transpose(1, 2)
    dx = ops.conv1d_cl_sparse_in(dev(dyk, dtype), wt, None, k=k, pad=k - 1 - k // 2, groups=G, T=T, n_out=D, n_seq=N, rowidx=rowidx.cuda())
    dx0 = ops.conv1d_cl(dev(dyd.view(N, T, D), dtype), wt, None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D)
    assert rel_err(dx.float(), ref_dx) < tol(dtype, 2)
    assert rel_err(dx.float(), dx0.float()) < 3e-3
    perm = torch.cat([torch.randperm(K, generator=gen) + c * K for c in range(N)])       # any order inside a sequence's rows
    dxp = ops.conv1d_cl_sparse_in(dev(dyk[perm], dtype), wt, None, k=k, pad=k - 1 - k // 2, groups=G, T=T, n_out=D, n_seq=N,
                                  rowidx=rowidx[perm].cuda().contiguous())
    assert rel_err(dxp.float(), ref_dx) < tol(dtype, 2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_dgrad_paths(ops, dtype):
    # stride-1 grouped conv: input gradient as a convolution with flipped/transposed weights (pack mode 1)
    B, T, D, G, k = 2, 120, 64, 4, 7
    Cg = D // G
    x = rnd(B, D, T, dtype=dtype, seed=1).requires_grad_(True)
    w = rnd(D, Cg, k, dtype=dtype, seed=2, scale=0.1)
    dy = rnd(B, D, T, dtype=dtype, seed=3)
    F.conv1d(x, w, padding=k // 2, groups=G).backward(dy)
    wt = torch.empty(G, Cg, k * Cg, dtype=dtype, device="cuda")
    ops.pack_weight(dev(w, torch.float32), wt, G=G, N=Cg, Cg=Cg, k=k, mode=1)
    dx = ops.conv1d_cl(to_cl(dy, D, dtype), wt, None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D)
    assert rel_err(dx.float().transpose(1, 2), x.grad) < tol(dtype, 2)
    # strided dense conv: dA = dY @ Wmat (pack mode 2) then col2im
    Cin, Cout, k, s, p = 64, 96, 10, 5, 3
    T = 200
    T_out = (T + 2 * p - k) // s + 1
    x = rnd(B, Cin, T, dtype=dtype, seed=4).requires_grad_(True)
    w = rnd(Cout, Cin, k, dtype=dtype, seed=5, scale=0.05)
    dy = rnd(B, Cout, T_out, dtype=dtype, seed=6)
    F.conv1d(x, w, stride=s, padding=p).backward(dy)
    wm = torch.empty(k * Cin, Cout, dtype=dtype, device="cuda")
    ops.pack_weight(dev(w, torch.float32), wm, G=1, N=Cout, Cg=Cin, k=k, mode=2)
    dA = ops.linear(to_cl(dy, Cout, dtype).view(B * T_out, Cout), wm)
    dx = ops.col2im(dA, B=B, T_in=T, T_out=T_out, C_=Cin, k=k, s=s, p=p)
    assert rel_err(dx.float().transpose(1, 2), x.grad) < tol(dtype, 2)
    # Linear: dX = dY @ W via W^T from pack mode 1, dW via wgrad
    M, N, K = 150, 96, 64
    xx, ww, dd = rnd(M, K, dtype=dtype, seed=7), rnd(N, K, dtype=dtype, seed=8), rnd(M, N, dtype=dtype, seed=9)
    wT = torch.empty(K, N, dtype=dtype, device="cuda")
    ops.pack_weight(dev(ww, torch.float32), wT, G=1, N=N, Cg=K, k=1, mode=1)
    assert rel_err(ops.linear(dev(dd, dtype), wT).float(), dd @ ww) < tol(dtype)
    gw = torch.zeros(N, K, dtype=torch.float32, device="cuda")
    ops.linear_wgrad(dev(xx, dtype), dev(dd, dtype), gw)
    assert rel_err(gw, dd.t() @ xx) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_large_m(ops, dtype):
    M, N, K = 5000, 256, 192
    x, dy = rnd(M, K, dtype=dtype, seed=1), rnd(M, N, dtype=dtype, seed=2)
    gw = torch.zeros(N, K, dtype=torch.float32, device="cuda")
    ops.linear_wgrad(dev(x, dtype), dev(dy, dtype), gw)
    assert rel_err(gw, dy.t() @ x) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(13440, 1024, 512), (1100, 512, 256), (300, 128, 96)])
def test_linear_dact_pair(ops, dtype, M, N, K):
    """A2V_EPI_DACT: the forward keeps gelu'(x W^T + b) instead of the pre-activation (one evaluation gives gelu and gelu'), the
    backward epilogue multiplies by it -- the Mlp's fc1 / fc2-input-gradient pair of the engine.  Every GEMM kernel that serves
    these shapes (persistent 8-phase, 128 x 128 LDS-DMA, fp32) must implement both sides."""
    x, w, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2) * K ** -0.5, rnd(N, dtype=torch.float32, seed=3)
    dy = rnd(M, N, dtype=dtype, seed=4)
    pre = x.float() @ w.float().t() + b
    xd, wd, bd = dev(x, dtype), dev(w, dtype), b.cuda()
    keep = torch.empty(M, N, dtype=dtype, device="cuda")
    out = ops.linear(xd, wd, bd, gelu=True, pre_out=keep, dact=True)
    assert rel_err(out.float(), F.gelu(pre)) < tol(dtype)
    pre_g = pre.clone().requires_grad_(True)
    F.gelu(pre_g).sum().backward()
    assert rel_err(keep.float(), pre_g.grad) < tol(dtype)
    # backward side: dy W2-like product times the kept derivative (here: an identity-free check with a second weight)
    w2 = rnd(N, N, dtype=dtype, seed=5) * N ** -0.5
    cs = torch.zeros(N, dtype=torch.float32, device="cuda")
    dh = ops.linear(dev(dy, dtype), dev(w2, dtype), None, gelu_bwd_aux=keep, colsum=cs, dact=True)
    ref = (dy.float() @ w2.float().t()) * keep.float().cpu()
    assert rel_err(dh.float(), ref) < tol(dtype)
    assert rel_err(cs, dh.float().sum(0)) < 5e-3


# ------------------------------------------------------------------------------------------ attention
def attn_ref(qkv, pos, slopes, scale_h, H):
    N, S, W = qkv.shape
    hd = W // (3 * H)
    q, k, v = qkv.reshape(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, scale_h.view(1, 1, H, 1, 1))
    att = (q * hd ** -0.5) @ k.transpose(-2, -1) + bias
    att = att.softmax(-1)
    return (att @ v).transpose(1, 2).reshape(N, S, H * hd)


@pytest.mark.parametrize("dtype,hd", [(torch.float32, 16), (torch.float32, 64), (torch.bfloat16, 64), (torch.bfloat16, 32)])
@pytest.mark.parametrize("S", [37, 140, 145, 151, 160, 161, 192, 200])
def test_attention_fwd_bwd(ops, dtype, hd, S):
    N, H = 3, 4
    qkv = rnd(N, S, 3 * H * hd, dtype=dtype, seed=1).requires_grad_(True)
    g = torch.Generator().manual_seed(2)
    pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = (torch.rand(H, generator=g) + 0.5).requires_grad_(True)
    ref = attn_ref(qkv, pos, slopes, sc, H)
    do = rnd(N, S, H * hd, dtype=dtype, seed=3)
    ref.backward(do)
    qd = dev(qkv.detach(), dtype)
    out, lse = ops.attention_fwd(qd, pos.cuda(), slopes.cuda(), sc.detach().cuda(), H)
    assert rel_err(out.float(), ref) < tol(dtype, 2)
    dal = torch.zeros(H, dtype=torch.float32, device="cuda")
    dqkv = ops.attention_bwd(qd, out, dev(do, dtype), lse, pos.cuda(), slopes.cuda(), sc.detach().cuda(), H, dal)
    t = tol(dtype, 3)
    assert rel_err(dqkv.float(), qkv.grad) < t
    assert rel_err(dal, sc.grad) < max(t, 2e-4)


@pytest.mark.parametrize("S,kind", [(140, "kept"), (152, "kept"), (180, "kept"), (386, "kept"), (772, "kept"), (2000, "arange")])
def test_alibi_scale_gradient_is_well_conditioned_in_bf16(ops, S, kind):
    """d alibi_scale_h = -slope_h sum_ij dS_ij |p_i - p_j| cancels heavily, and dS_ij = P_ij (dP_ij - delta_i) with delta_i formed
    from the bf16-ROUNDED forward output leaves a residue e_i = sum_j dS_ij != 0 in every row, which the uncentred sum multiplies
    by the row's mean attended distance (8e-2 at 8 kHz, 2.4e-1 at T = 11 025).  The kernels centre the
    distance on c_i = sum_j P_ij |p_i - p_j| (a2v.h: mdist), which removes that term exactly.  Checked on every bf16 backward kernel
    (S <= 160 three-wave, <= 192 resident, two-pass long with kept positions, two-pass long with arange) against an fp64 autograd
    reference on the bf16-rounded inputs, with a dO that correlates with O (what makes delta large): the centred gradient is held
    to 1e-3 (identical bf16 inputs on both sides: what is left is fp32 summation and the bf16 rounding of P / dS inside the
    kernel); the uncentred one (mdist = NULL, the ABI-7 arithmetic) is worse on the long sequences (measured 1.6e-4 against
    6.4e-5 at S = 386).  What centring cannot remove is the sensitivity of this sum to the rounding of its INPUTS
    (d dalibi / d dO_i = sum_j P_ij (|.| - c_i) V_j): the engine-level tests state that part."""
    N, H, hd = 2, 4, 64
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    qkv = (rnd(N, S, 3 * H * hd, dtype=dtype, seed=11) * 1.5).double().requires_grad_(True)
    if kind == "kept":
        T = 2000 if S <= 200 else 11025
        pos = torch.stack([torch.randperm(T, generator=g)[:S].sort()[0] for _ in range(N)]).int()
    else:
        pos = torch.arange(S).repeat(N, 1).int()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = (torch.rand(H, generator=g) * 0.2 + 0.05).double().requires_grad_(True)     # gentle slopes: long attended distances
    ref = attn_ref(qkv, pos, slopes.double(), sc, H)
    do = (ref.detach().float() * 0.7 + rnd(N, S, H * hd, dtype=dtype, seed=13) * 0.3).to(dtype).double()
    ref.backward(do)
    qd = dev(qkv.detach().float(), dtype)
    pd = None if kind == "arange" else pos.cuda()
    scd = sc.detach().float().cuda()
    out, lse = ops.attention_fwd(qd, pd, slopes.cuda(), scd, H)
    dal = torch.zeros(H, dtype=torch.float32, device="cuda")
    dqkv = ops.attention_bwd(qd, out, dev(do.float(), dtype), lse, pd, slopes.cuda(), scd, H, dal)
    assert rel_err(dqkv.float(), qkv.grad) < 3e-2
    e_c = rel_err(dal, sc.grad)
    dal0 = torch.zeros(H, dtype=torch.float32, device="cuda")
    ops.attention_bwd(qd, out, dev(do.float(), dtype), lse[0].contiguous(), pd, slopes.cuda(), scd, H, dal0)     # bare lse: uncentred
    e_u = rel_err(dal0, sc.grad)
    assert e_c < 1e-3, (e_c, e_u)
    if S >= 386:
        assert e_u > e_c, (e_c, e_u)


@pytest.mark.parametrize("kern", [1, 0])
@pytest.mark.parametrize("S", [700, 64, 130, 257, 1000, 2000])
@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_teacher_arange_long(ops, dtype, S, kern, request):
    """positions = arange (the teacher): the bf16 kernel takes the per-key ALiBi form off the diagonal (bias as the MFMA's
    C operand, row constant dropped, running maximum re-based at the two phase changes) -- output AND log-sum-exp must come
    out in true units whatever phase the last key tile was in (ragged last tile in either form: S = 700, 130, 257, 1000).
    kern: a2v_debug_set(36): the 32 x 32 x 16 kernel of round 5 (default) and the 16 x 16 x 32 one it replaced for S >= 512."""
    N, H, hd = 1, 2, 64
    if dtype == torch.float32 and S > 1000:
        pytest.skip("fp32 reference kernel: covered at S <= 1000")
    if kern == 0 and (dtype != torch.bfloat16 or S < 512):
        pytest.skip("the switch only exists for the bf16 long-sequence forward")
    ops.debug_set(36, kern)
    request.addfinalizer(lambda: ops.debug_set(36, 1))
    qkv = rnd(N, S, 3 * H * hd, dtype=dtype, seed=4)
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = torch.tensor([1.3, -0.2])          # negative learned scale is clamped to 0
    pos = torch.arange(S).unsqueeze(0).int()
    ref = attn_ref(qkv, pos, slopes, sc, H)
    out, lse = ops.attention_fwd(dev(qkv, dtype), None, slopes.cuda(), sc.cuda(), H, need_lse=True)
    assert rel_err(out.float(), ref) < tol(dtype, 2)
    q, k, _ = qkv.float().view(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, sc.view(1, 1, H, 1, 1))
    sc_ = (q * hd ** -0.5) @ k.transpose(-2, -1) + bias
    lse_ref = torch.logsumexp(sc_, dim=-1)
    assert lse.shape == (2, N, H, S)                     # [0] log-sum-exp, [1] mean attended distance (a2v.h: mdist)
    assert (lse[0].cpu() - lse_ref.view_as(lse[0].cpu())).abs().max() < (2e-2 if dtype == torch.bfloat16 else 1e-4)
    dist = (pos[0].float().view(S, 1) - pos[0].float().view(1, S)).abs()
    md_ref = (sc_.softmax(-1) * dist).sum(-1)
    assert rel_err(lse[1], md_ref.view_as(lse[1].cpu())) < (2e-2 if dtype == torch.bfloat16 else 1e-4)
    # steep bias x long distances (a learned scale of 40-80: the per-key terms reach 1e4 accumulator units far from the diagonal)
    sc2 = sc.abs() * 64
    out2, _ = ops.attention_fwd(dev(qkv, dtype), None, slopes.cuda(), sc2.cuda(), H, need_lse=False)
    assert rel_err(out2.float(), attn_ref(qkv, pos, slopes, sc2, H)) < tol(dtype, 2)


@pytest.mark.parametrize("S", [700, 2000])
def test_attention_alibi_band_limit_changes_nothing(ops, S):
    """positions = arange, S >= 512: key tiles beyond d_cut = (2B + 20) / (slope * alibi_scale) of every query row of a workgroup are
    skipped (B = bound of |q.k| scale from the row norms, attn_qk_bound_kernel).  The skipped softmax weights are below e^-20 of
    the row sum EACH (the bound; 4e-6 over 2000 keys, 500 times below the 2^-9 rounding of the kernel's P), so the output must agree
    with the all-tiles run bit for bit in bf16 and the lse to 1e-5 -- whatever the slopes (steep: most tiles skipped; flat / zero
    scale: none)."""
    N, H, hd = 2, 4, 64
    qkv = dev(rnd(N, S, 3 * H * hd, dtype=torch.bfloat16, seed=11) * 1.5, torch.bfloat16)
    slopes = torch.tensor(O.alibi_slopes(H)).cuda()
    for sc in ([1.0, 1.0, 1.0, 1.0], [40.0, 3.0, 0.0, -1.0], [0.02, 0.2, 8.0, 1.0]):
        scd = torch.tensor(sc).cuda()
        ops.debug_set(19, 1)
        o1, l1 = ops.attention_fwd(qkv, None, slopes, scd, H, need_lse=True)
        ops.debug_set(19, 0)
        o0, l0 = ops.attention_fwd(qkv, None, slopes, scd, H, need_lse=True)
        ops.debug_set(19, 1)
        assert torch.equal(o1, o0), sc                      # bf16 outputs: bit-identical
        assert (l1 - l0).abs().max() < 1e-5, sc


@pytest.mark.parametrize("S,nclip", [(512, 4), (600, 4), (2000, 2)])
def test_linear_qknorm_table_and_band_limit(ops, S, nclip):
    """ABI 14: the qkv product of a long-sequence layer run with A2V_EPI_QKNORM leaves, per clip and 32-column segment, the maximum over
    the clip's rows of the segment's squared norm (from the fp32 accumulators, clips that start inside a 256- / 224-row tile included);
    the output itself is the plain Linear's bit for bit; and the attention forward fed with the table gives the all-tiles result bit
    for bit, as it does with the bound from its own pass over q and k (test_attention_alibi_band_limit_changes_nothing)."""
    H, hd, D = 4, 64, 256
    M, N3 = nclip * S, 3 * H * hd
    x = dev(rnd(M, D, dtype=torch.bfloat16, seed=5), torch.bfloat16)
    w = dev(rnd(N3, D, dtype=torch.bfloat16, seed=6) * (1.5 * D ** -0.5), torch.bfloat16)
    b = dev(rnd(N3, dtype=torch.float32, seed=7) * 0.1, torch.float32)
    out, table = ops.linear_qknorm(x, w, b, S)
    assert table is not None, "the persistent kernel must take this product (M >= 1024, N % 256 == 0, K % 128 == 0)"
    ref = ops.linear(x, w, b)
    assert torch.equal(out, ref)
    full = x.float() @ w.float().t() + b
    want = (full.view(nclip, S, N3 // 32, 32) ** 2).sum(-1).amax(1)
    assert table.shape == want.shape
    assert rel_err(table, want) < 1e-4
    assert ((table - want) / want).abs().max() < 2e-3           # every entry (fp32 sums in another order)
    # consumer: the bound made of the table (Cauchy-Schwarz per 32-column segment, summed over a head's two segments) is never
    # below the largest |q_i . k_j| of its (clip, head)
    qk = full.view(nclip, S, 3, H, hd)
    t4 = table.view(nclip, 3, H, 2)
    bnd = (t4[:, 0] * t4[:, 1]).sqrt().sum(-1)                     # (clip, H)
    scores = torch.einsum("nihd,njhd->nhij", qk[:, :, 0], qk[:, :, 1]).abs().amax((2, 3))
    assert (bnd >= scores * (1 - 1e-3)).all()
    slopes = torch.tensor(O.alibi_slopes(H)).cuda()
    for sc in ([1.0, 1.0, 1.0, 1.0], [40.0, 3.0, 0.0, -1.0]):
        scd = torch.tensor(sc).cuda()
        o1, l1 = ops.attention_fwd(out.view(nclip, S, N3), None, slopes, scd, H, need_lse=True, qk_norms=table)
        ops.debug_set(19, 0)
        o0, l0 = ops.attention_fwd(out.view(nclip, S, N3), None, slopes, scd, H, need_lse=True)
        ops.debug_set(19, 1)
        assert torch.equal(o1, o0), sc
        assert (l1 - l0).abs().max() < 1e-5, sc


def test_linear_qknorm_refused_where_no_kernel_forms_it(ops):
    """a product the persistent kernel does not take (M < 1024 here) is refused with the flag -- never run without the table -- and
    ops.linear_qknorm falls back to the plain Linear + the attention's own pass"""
    x = dev(rnd(2 * 300, 256, dtype=torch.bfloat16, seed=1), torch.bfloat16)
    w = dev(rnd(768, 256, dtype=torch.bfloat16, seed=2) * 0.1, torch.bfloat16)
    b = dev(rnd(768, dtype=torch.float32, seed=3), torch.float32)
    out = torch.empty(600, 768, dtype=torch.bfloat16, device="cuda")
    table = torch.zeros(2, 24, device="cuda")
    r = ops.conv_gemm(x, w, out, M=600, N=768, K=256, T_in=300, T_out=300, Cg=256, a_pitch=256, a_bstride=300 * 256, ldb=256, ldc=768,
                      bias=b, qknorm=table)
    assert r is None and float(table.abs().max()) == 0.0
    o2, t2 = ops.linear_qknorm(x, w, b, 300)
    assert t2 is None and torch.equal(o2, ops.linear(x, w, b))


@pytest.mark.parametrize("S,kind,drop", [(2000, "arange", None), (1100, "arange", None), (700, "kept", None), (700, "kept", (0.1, 77)),
                                         (640, "arange", (0.1, 5))])
def test_attention_fwd32_matches_the_16x16_kernel(ops, S, kind, drop):
    """Round 5: for S >= 512 the bf16 forward runs on v_mfma_f32_32x32x16_bf16 (attn_fwd32_bf16_kernel: a wave owns 32 query rows as one
    column block, K / V tiles arrive by LDS-DMA, fragments by counted raw LDS reads).  Same arithmetic as the 16 x 16 x 32 kernel up to the
    order of the fp32 sums: output (bf16), log-sum-exp and mean attended distance of all its variants (forward only; with lse + mdist;
    explicit positions = every tile in the |.| form; dropout: same mask, keyed by (row, key)) are held against that kernel."""
    N, H, hd = 2, 4, 64
    g = torch.Generator().manual_seed(3)
    qkv = dev(rnd(N, S, 3 * H * hd, dtype=torch.bfloat16, seed=21) * 1.3, torch.bfloat16)
    slopes = torch.tensor(O.alibi_slopes(H)).cuda()
    sc = torch.tensor([1.0, 0.3, 6.0, 0.0]).cuda()
    pos = None if kind == "arange" else torch.stack([torch.randperm(4000, generator=g)[:S].sort()[0] for _ in range(N)]).int().cuda()
    res = {}
    for kern in (1, 0):
        ops.debug_set(36, kern)
        o, lse = ops.attention_fwd(qkv, pos, slopes, sc, H, need_lse=True, drop=drop)
        o2 = ops.attention_fwd(qkv, pos, slopes, sc, H, need_lse=False, drop=drop)[0] if drop is None else o
        res[kern] = (o.float(), lse.clone(), o2.float())
    ops.debug_set(36, 1)
    assert rel_err(res[1][0], res[0][0]) < 4e-3                      # bf16 outputs of two summation orders
    assert rel_err(res[1][2], res[0][2]) < 4e-3
    assert (res[1][1][0] - res[0][1][0]).abs().max() < 1e-4          # lse
    assert rel_err(res[1][1][1], res[0][1][1]) < 1e-4                # mean attended distance


@pytest.mark.parametrize("S", [700, 2000])
def test_attention_x3_alibi_band_limit_changes_nothing(ops, S):
    """The bf16x3 forward's ALiBi band limit (round 4; the bf16 kernel's rule with exact fp32 row norms): positions = arange, S >= 512;
    against the all-tiles run (a2v_debug_set(19, 0)) output, LSE and mean attended distance agree to fp32 round-off, whatever the
    slopes (steep: most key tiles skipped; flat / zero / negative scale: none)."""
    N, H, hd = 2, 4, 64
    qkv = dev(rnd(N, S, 3 * H * hd, dtype=torch.float32, seed=12) * 1.5, torch.float32)
    slopes = torch.tensor(O.alibi_slopes(H)).cuda()
    ops.X3_WGRAD = True
    try:
        for sc in ([1.0, 1.0, 1.0, 1.0], [40.0, 3.0, 0.0, -1.0], [0.02, 0.2, 8.0, 1.0]):
            scd = torch.tensor(sc).cuda()
            ops.debug_set(19, 1)
            o1, l1 = ops.attention_fwd(qkv, None, slopes, scd, H, need_lse=True)
            ops.debug_set(19, 0)
            o0, l0 = ops.attention_fwd(qkv, None, slopes, scd, H, need_lse=True)
            ops.debug_set(19, 1)
            assert rel_err(o1, o0) < 1e-6, sc
            assert (l1[0] - l0[0]).abs().max() < 1e-5, sc
            assert rel_err(l1[1], l0[1]) < 1e-5, sc
    finally:
        ops.debug_set(19, 1)
        ops.X3_WGRAD = False


# ------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C_,act,affine", [(64, 0, True), (512, 1, True), (1024, 0, True), (768, 1, False), (32, 1, True)])
def test_layernorm_fwd_bwd(ops, dtype, C_, act, affine):
    M = 203
    x = rnd(M, C_, dtype=dtype, seed=1).requires_grad_(True)
    w = (rnd(C_, dtype=torch.float32, seed=2) * 0.2 + 1).requires_grad_(affine)
    b = (rnd(C_, dtype=torch.float32, seed=3) * 0.2).requires_grad_(affine)
    y = F.layer_norm(x, (C_,), w if affine else None, b if affine else None, 1e-5)
    if act:
        y = F.gelu(y)
    dy = rnd(M, C_, dtype=dtype, seed=4)
    y.backward(dy)
    wd = dev(w.detach(), torch.float32) if affine else None
    bd = dev(b.detach(), torch.float32) if affine else None
    xd = dev(x.detach(), dtype)
    yd, mean, rstd = ops.layernorm_fwd(xd, wd, bd, 1e-5, act=act)
    assert rel_err(yd.float(), y) < tol(dtype)
    dg = torch.zeros(C_, device="cuda") if affine else None
    db = torch.zeros(C_, device="cuda") if affine else None
    dxs = torch.zeros(C_, device="cuda")
    dx = ops.layernorm_bwd(dev(dy, dtype), xd, wd, bd, mean, rstd, act=act, dgamma=dg, dbeta=db, dxsum=dxs)
    assert rel_err(dx.float(), x.grad) < tol(dtype, 2)
    assert rel_err(dxs, dx.float().sum(0)) < 1e-3
    if affine:
        assert rel_err(dg, w.grad) < tol(dtype, 2)
        assert rel_err(db, b.grad) < tol(dtype, 2)


@pytest.mark.parametrize("C_,act,affine,post", [(1024, 0, True, False), (512, 1, True, False), (1024, 1, False, False),
                                                (768, 1, False, True), (768, 1, False, False), (512, 0, True, False),
                                                (256, 1, False, False)])
@pytest.mark.parametrize("rows", [0, 1, 2, 4, 8])
def test_layernorm_stream_fwd(ops, C_, act, affine, post, rows):
    """The persistent streaming forward (bf16, a2v_norm.hip round 5) against torch and, bit for bit, against the row-per-wave
    kernel it replaces; `rows` forces the rows per batch (ragged last batch: M is not a multiple of any of them)."""
    dtype = torch.bfloat16
    M = 3 * 2048 + 5
    x = (rnd(M, C_, dtype=dtype, seed=1) * 1.5 + 0.3).to(dtype).float()
    w = rnd(C_, dtype=torch.float32, seed=2) * 0.2 + 1 if affine else None
    b = rnd(C_, dtype=torch.float32, seed=3) * 0.2 if affine else None
    pz = rnd(M, C_, dtype=dtype, seed=5) if post else None
    y = F.layer_norm(x.float(), (C_,), w, b, 1e-5)
    if act:
        y = F.gelu(y)
    if post:
        y = y + pz.float()
    xd, wd, bd, pd = dev(x, dtype), (w.cuda() if affine else None), (b.cuda() if affine else None), (dev(pz, dtype) if post else None)
    try:
        ops.debug_set(31, rows)
        y1, m1, r1 = ops.layernorm_fwd(xd, wd, bd, 1e-5, act=act, post=pd)
        ops.debug_set(30, 0)
        y0, m0, r0 = ops.layernorm_fwd(xd, wd, bd, 1e-5, act=act, post=pd)
    finally:
        ops.debug_set(30, 3)
        ops.debug_set(31, 0)
    assert rel_err(y1.float(), y) < tol(dtype)
    assert torch.equal(m1, m0) and rel_err(r1, r0) < 1e-6       # (a ragged last chunk sums its zeros in another place: last-bit differences)
    assert rel_err(y1.float(), y0.float()) < (1e-6 if C_ % 512 == 0 else 1e-4)


@pytest.mark.parametrize("C_,act,affine,mode,xs", [(1024, 0, True, 2, True), (1024, 0, True, 0, True), (512, 0, True, 0, False),
                                                   (1024, 1, False, 0, True), (768, 1, False, 0, True), (512, 1, True, 0, False),
                                                   (768, 0, True, 2, True), (256, 1, False, 0, True)])
@pytest.mark.parametrize("rows", [0, 1, 2, 4])
def test_layernorm_stream_bwd(ops, C_, act, affine, mode, xs, rows):
    """The streaming backward against autograd and against the row-per-wave kernel (same dropout mask, same column sums)."""
    dtype = torch.bfloat16
    M = 3 * 2048 + 3
    x = (rnd(M, C_, dtype=dtype, seed=1) * 1.5 + 0.3).to(dtype).float().requires_grad_(True)
    w = (rnd(C_, dtype=torch.float32, seed=2) * 0.2 + 1).requires_grad_(affine)
    b = (rnd(C_, dtype=torch.float32, seed=3) * 0.2).requires_grad_(affine)
    y = F.layer_norm(x, (C_,), w if affine else None, b if affine else None, 1e-5)
    if act:
        y = F.gelu(y)
    dy = rnd(M, C_, dtype=dtype, seed=4)
    y.backward(dy)
    xd, dyd = dev(x.detach(), dtype), dev(dy, dtype)
    wd, bd = (w.detach().cuda(), b.detach().cuda()) if affine else (None, None)
    _, mean, rstd = ops.layernorm_fwd(xd, wd, bd, 1e-5, act=act)
    kw = dict(drop=(0.1, 4242), drop_mode=2) if mode == 2 else {}
    outs = []
    try:
        for stream in (3, 0):
            ops.debug_set(30, stream)
            ops.debug_set(31, rows)
            dg = torch.zeros(C_, device="cuda") if affine else None
            db = torch.zeros(C_, device="cuda") if affine else None
            dxs = torch.zeros(C_, device="cuda") if xs else None
            r = ops.layernorm_bwd(dyd, xd, wd, bd, mean, rstd, act=act, dgamma=dg, dbeta=db, dxsum=dxs, **kw)
            dx, dxd = r if mode == 2 else (r, None)
            outs.append((dx, dxd, dg, db, dxs))
    finally:
        ops.debug_set(30, 3)
        ops.debug_set(31, 0)
    (dx1, dxd1, dg1, db1, dxs1), (dx0, dxd0, dg0, db0, dxs0) = outs
    assert rel_err(dx1.float(), x.grad.float()) < tol(dtype, 2)
    assert rel_err(dx1.float(), dx0.float()) < 3e-3              # (another grouping of the same fp32 terms, then bf16 rounding)
    if mode == 2:
        keep = ops.dropout_mask(M * C_, 0.1, 4242).view(M, C_).float()
        assert rel_err(dxd1.float(), dx1.float() * keep / 0.9) < tol(dtype)
        assert rel_err(dxd1.float(), dxd0.float()) < 3e-3
    if xs:
        assert rel_err(dxs1, (dxd1 if mode == 2 else dx1).float().sum(0)) < 1e-4
        assert rel_err(dxs1, dxs0) < 2e-2          # (sums of ~6 k bf16-rounded terms that differ in the last bit here and there)
    if affine:
        assert rel_err(dg1, w.grad.cuda()) < tol(dtype, 2) and rel_err(db1, b.grad.cuda()) < tol(dtype, 2)
        assert rel_err(dg1, dg0) < 1e-4 and rel_err(db1, db0) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_residual_in(ops, dtype):
    M, C_ = 100, 128
    x, r = rnd(M, C_, dtype=dtype, seed=1), rnd(M, C_, dtype=dtype, seed=2)
    w, b = torch.ones(C_), torch.zeros(C_)
    y, _, _ = ops.layernorm_fwd(dev(x, dtype), w.cuda(), b.cuda(), 1e-5, res=dev(r, dtype), save_stats=False)
    assert rel_err(y.float(), F.layer_norm(x + r, (C_,))) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,D,Ln", [(2, 200, 64, 3), (3, 301, 1024, 16), (2, 77, 512, 5), (1, 2000, 1024, 12)])
def test_instnorm_targets(ops, dtype, B, T, D, Ln):
    """make_targets: (the D % 512 == 0 shapes take the statistics-in-registers kernel, the others the frame-per-wave one)"""
    ts = [(rnd(B, T, D, dtype=dtype, seed=10 + i) * (i + 1) + i).to(dtype).float() for i in range(Ln)]
    ref = O.make_targets([t.clone() for t in ts], Ln)
    y = ops.instnorm_targets([dev(t, dtype) for t in ts])
    assert rel_err(y, ref) < 2e-5     # statistics and output are fp32 in both modes


def test_instnorm_targets_large_channel_offsets_fp32(ops):
    """channels whose mean dwarfs their deviation (|mean| / std ~ 1e3: bias-dominated or outlier channels of trained
    checkpoints).  The fp32 / bf16x3 parity modes must subtract the mean per element -- the statistics-in-registers form
    (sum x rstd - sum mean rstd) cancels two terms of size 1e3 there -- and the shifted-moment statistics must hold too."""
    B, T, D, Ln = 2, 300, 1024, 8
    g = torch.Generator().manual_seed(4)
    off = torch.randn(1, 1, D, generator=g) * 1e3
    ts = [(torch.randn(B, T, D, generator=g) * (0.5 + i) + off * (1 + 0.1 * i)) for i in range(Ln)]
    ref = sum((t.double() - t.double().mean(1, keepdim=True)) / (t.double().var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
              for t in ts).div(Ln).float()                   # instance_norm over time per (clip, channel), fp64 (data2vec2.py:1023-1066)
    y = ops.instnorm_targets([dev(t, torch.float32) for t in ts])
    assert rel_err(y, ref) < 1e-4, rel_err(y, ref)          # (fp32 data: |mean| / std x 2^-24 = 6e-5 is the floor of ANY fp32 evaluation)


# ------------------------------------------------------------------------------------------ sinc
@pytest.mark.parametrize("C_,k,sr", [(127, 63, 8000), (31, 15, 8000), (127, 63, 44100), (296, 149, 44100)])
def test_sinc_filters_fwd_bwd(ops, C_, k, sr):
    lo, band = O.sinc_init(C_, k, sr)
    g = torch.Generator().manual_seed(1)
    lo = (lo + torch.randn(lo.shape, generator=g) * 10).requires_grad_(True)
    band = (band + torch.randn(band.shape, generator=g) * 10).requires_grad_(True)
    ref = O.sinc_filters(lo, band, k, sr)[:, 0]
    win, n_ = O.sinc_window_and_n(k, sr)
    kw = dict(C_=C_, k=k, min_low_hz=50.0, min_band_hz=float(math.ceil(sr / k)), sample_rate=sr)
    args = (lo.detach().flatten().cuda(), band.detach().flatten().cuda(), n_.flatten().cuda(), win.cuda())
    Cp = ops.sinc_out_channels(C_, k)
    f = ops.sinc_filters_fwd(*args, Cp=Cp, **kw)
    assert rel_err(f[:C_], ref) < 1e-5 and float(f[C_:].abs().max() if C_ < Cp else 0) == 0
    df = torch.randn(C_, k, generator=g)
    ref.backward(df)
    dlow, dband = ops.sinc_filters_bwd(*args, df.cuda(), **kw)
    assert rel_err(dlow, lo.grad.flatten()) < 1e-3
    assert rel_err(dband, band.grad.flatten()) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C_,k,L,sr", [(127, 63, 1000, 8000), (31, 15, 640, 8000), (296, 149, 1500, 44100), (140, 65, 700, 16000),
                                        (100, 201, 900, 44100)])
def test_sinc_fused_fwd_bwd(ops, dtype, C_, k, L, sr):
    """(127,63), (31,15): the single-pass kernels; the others (more than 128 filters or more than 63 taps) take the
    blocked kernels -- 2, 3 and 4 tap chunks, ragged channel groups, time tails."""
    B = 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, L, generator=g)
    lo, band = O.sinc_init(C_, k, sr)
    filt = O.sinc_filters(lo, band, k, sr)[:, 0].clone().requires_grad_(True)
    lw = (torch.randn(C_, generator=g) * 0.2 + 1).requires_grad_(True)
    lb = (torch.randn(C_, generator=g) * 0.2).requires_grad_(True)
    pa = (torch.randn(C_, generator=g) * 0.2 + 2).requires_grad_(True)
    pb = (torch.randn(C_, generator=g) * 0.3).requires_grad_(True)
    pad = k // 2
    conv = F.conv1d(F.pad(x.unsqueeze(1), (pad, pad), mode="reflect"), filt.unsqueeze(1))
    yl = O.channel_layer_norm(conv, lw, lb)
    ref = O.pswish(yl, pa.view(1, -1, 1), pb.view(1, -1, 1))             # (B,C,L)
    dyr = rnd(B, C_, L, dtype=dtype, seed=4)
    ref.backward(dyr)
    cu = lambda t: t.detach().cuda()
    y, mean, rstd = ops.sinc_fused_fwd(cu(x), cu(filt), cu(lw), cu(lb), cu(pa), cu(pb), C_=C_, k=k, dtype=dtype)
    assert rel_err(y[..., :C_].float().transpose(1, 2), ref) < tol(dtype)
    assert float(y[..., C_:].float().abs().max()) == 0.0
    Cp = ops.sinc_out_channels(C_, k)
    assert y.shape[-1] == Cp == (128 if (C_ <= 128 and k <= 63) else 320)
    dycl = torch.zeros(B, L, Cp)
    dycl[..., :C_] = dyr.transpose(1, 2)
    dfilt = torch.zeros(C_, k, device="cuda")
    dpar = torch.zeros(4, Cp, device="cuda")
    ops.sinc_fused_bwd(cu(x), cu(filt), cu(lw), cu(lb), cu(pa), cu(pb), dev(dycl, dtype), mean, rstd, dfilt, dpar, C_=C_, k=k)
    # the filter gradient is defined on the symmetric filter: mirrored taps summed (what sinc_filters_bwd consumes)
    fold = lambda d: torch.cat([d[:, :k // 2] + d[:, k // 2 + 1:].flip(1), d[:, k // 2:k // 2 + 1]], dim=1)
    assert rel_err(fold(dfilt.cpu()), fold(filt.grad)) < 2e-4
    for i, p in enumerate((lw, lb, pa, pb)):
        assert rel_err(dpar[i, :C_], p.grad) < 2e-4, i


# ------------------------------------------------------------------------------------------ tokens / loss
@pytest.mark.parametrize("dtype", DTYPES)
def test_token_plumbing(ops, dtype):
    B, cb, T, D, K = 2, 3, 50, 64, 7
    N = B * cb
    mask = O.seeded_mask(N, T, K, 5)
    ids_keep, ids_restore = O.make_maskinfo(mask)
    x = rnd(B, T, D, dtype=dtype, seed=1)
    xp = rnd(N, T, D, dtype=dtype, seed=2)
    gi = ids_keep.unsqueeze(-1).expand(-1, -1, D)
    ref = torch.gather(x.repeat_interleave(cb, 0), 1, gi) + torch.gather(xp, 1, gi)
    out = ops.gather_tokens(dev(x, dtype), ids_keep.int().cuda(), cb=cb, b=dev(xp, dtype))
    assert rel_err(out.float(), ref) < tol(dtype)
    # decoder input
    tok = rnd(N, K, D, dtype=dtype, seed=3)
    noise = rnd(N, T - K, D, dtype=torch.float32, seed=4) * 0.01
    ref = O.decoder_input(tok, ids_restore, noise)
    out = ops.decoder_input(dev(tok, dtype), noise.cuda(), ids_restore.int().cuda())
    assert rel_err(out.float(), ref.to(dtype).float()) < 1e-6
    # scatter + clone reduce == autograd of (repeat_interleave, mask, gather)
    xr = x.clone().requires_grad_(True)
    xc = xr.repeat_interleave(cb, 0)
    xm = xc * (1 - mask.float().unsqueeze(-1))
    tokens = torch.gather(xc, 1, gi)
    dxm, dtok = rnd(N, T, D, dtype=dtype, seed=5), rnd(N, K, D, dtype=dtype, seed=6)
    ((xm * dxm).sum() + (tokens * dtok).sum()).backward()
    out = ops.clone_reduce(dev(dxm, dtype), dev(dtok, dtype), ids_restore.int().cuda(), cb)
    assert rel_err(out.float(), xr.grad) < tol(dtype)
    sc = ops.scatter_tokens(dev(dtok, dtype), ids_restore.int().cuda(), T)
    ref = torch.zeros(N, T, D).scatter_(1, gi, dtok)
    assert rel_err(sc.float(), ref) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_d2v_loss(ops, dtype):
    B, cb, T, D, K = 2, 3, 40, 64, 6
    N = B * cb
    mask = O.seeded_mask(N, T, K, 9)
    x = rnd(N, T, D, dtype=dtype, seed=1).requires_grad_(True)
    y = rnd(B, T, D, dtype=torch.float32, seed=2)
    mb = mask.bool()
    ysel = y.repeat_interleave(cb, 0)[mb]
    loss = O.d2v_loss(x[mb], ysel)
    loss.sum().backward()
    stats = torch.zeros(4 * D + 2, dtype=torch.float64, device="cuda")
    rows = torch.empty(N, T, D, dtype=torch.float32, device="cuda")
    dx = ops.d2v_loss(dev(x.detach(), dtype), y.cuda(), mask.cuda(), cb, stats=stats, loss_rows=rows)
    st = stats.cpu()
    assert abs(float(st[4 * D]) - float(loss.sum())) / float(loss.sum()) < 1e-5
    assert int(st[4 * D + 1]) == int(mask.sum())
    assert rel_err(rows[mb.cuda()], loss) < 1e-5
    assert rel_err(dx.float(), x.grad) < tol(dtype)
    n = float(st[4 * D + 1])
    for off, ref in ((0, x.detach()[mb]), (2 * D, ysel)):
        s, ss = st[off:off + D], st[off + D:off + 2 * D]
        var = (ss - s * s / n) / (n - 1)
        v = torch.sqrt(var + 1e-6).mean()
        assert abs(float(v) - float(O.compute_var(ref))) < 1e-5


@pytest.mark.parametrize("shape", [(3, 5, 70, 768, 9), (2, 12, 33, 1024, 4), (1, 1, 17, 8, 3), (2, 4, 20, 520, 20), (2, 3, 19, 1536, 5),
                                   (1, 6, 21, 2048, 2), (1, 2, 9, 4096, 3)])
def test_d2v_loss_clone_groups(ops, shape):
    """clone counts that are not multiples of the kernel's group of four, channel counts that leave lanes of the last
    512-column chunk idle, a single clone, nothing masked (K == T): same sums as the row-by-row definition"""
    B, cb, T, D, K = shape
    N = B * cb
    mask = O.seeded_mask(N, T, K, 11)
    x = rnd(N, T, D, dtype=torch.bfloat16, seed=1)
    y = rnd(B, T, D, dtype=torch.float32, seed=2)
    mb = mask.bool()
    xs, ys = x.float()[mb], y.repeat_interleave(cb, 0)[mb]
    stats = torch.zeros(4 * D + 2, dtype=torch.float64, device="cuda")
    rows = torch.full((N, T, D), 7.0, dtype=torch.float32, device="cuda")
    dx = ops.d2v_loss(dev(x, torch.bfloat16), y.cuda(), mask.cuda(), cb, stats=stats, loss_rows=rows)
    st = stats.cpu()
    scale = D ** -0.5
    ref_rows = torch.zeros(N, T, D)
    ref_rows[mb] = (xs - ys) ** 2 * scale
    ref_dx = torch.zeros(N, T, D)
    ref_dx[mb] = 2 * (xs - ys) * scale
    assert int(st[4 * D + 1]) == int(mask.sum())
    assert torch.allclose(rows.cpu(), ref_rows, rtol=1e-5, atol=1e-7)
    assert rel_err(dx.float(), ref_dx) < tol(torch.bfloat16) or float(ref_dx.abs().max()) == 0.0
    if int(mask.sum()):
        assert abs(float(st[4 * D]) - float(ref_rows.double().sum())) / float(ref_rows.double().sum()) < 1e-5
        for off, ref in ((0, xs), (D, xs ** 2), (2 * D, ys), (3 * D, ys ** 2)):
            assert torch.allclose(st[off:off + D].float(), ref.double().sum(0).float(), rtol=1e-4, atol=1e-4)
    else:
        assert float(st.abs().max()) == 0.0
    # the same call without the unreduced rows (the other instantiation) gives the same gradient and sums
    stats2 = torch.zeros_like(stats)
    dx2 = ops.d2v_loss(dev(x, torch.bfloat16), y.cuda(), mask.cuda(), cb, stats=stats2)
    assert torch.equal(dx2, dx)
    assert torch.allclose(stats2, stats, rtol=1e-6, atol=1e-6)       # (fp32 partial sums: the two instantiations contract differently)


def test_decoder_input_drawn_noise(ops):
    """a2v_decoder_input_rng: kept frames are the encoder's tokens, masked frames N(0, std) drawn in the kernel
    (base.py:172-176): moments of the draw, repeatable per seed, different between seeds"""
    N, T, K, D = 6, 500, 40, 256
    mask = O.seeded_mask(N, T, K, 3)
    ids_keep, ids_restore = O.make_maskinfo(mask)
    tok = rnd(N, K, D, dtype=torch.bfloat16, seed=3)
    r = ids_restore.int().cuda()
    out = ops.decoder_input(dev(tok, torch.bfloat16), (0.01, 1234), r)
    ref = O.decoder_input(tok.float(), ids_restore, torch.zeros(N, T - K, D))
    mb = mask.bool()
    assert torch.equal(out.float().cpu()[~mb], ref[~mb])
    z = out.float().cpu()[mb] / 0.01
    n = z.numel()
    assert abs(float(z.mean())) < 5 / n ** 0.5
    assert abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 4).mean()) - 3.0) < 0.1                   # kurtosis of a normal
    assert abs(float((z[:, 0::2] * z[:, 1::2]).mean())) < 5 / (n / 2) ** 0.5      # the two Box-Muller outputs of a pair
    assert abs(float((z[:-1] * z[1:]).mean())) < 5 / n ** 0.5                     # neighbouring frames
    again = ops.decoder_input(dev(tok, torch.bfloat16), (0.01, 1234), r)
    assert torch.equal(again, out)
    other = ops.decoder_input(dev(tok, torch.bfloat16), (0.01, 1235), r)
    assert not torch.equal(other, out)
    fp = ops.decoder_input(dev(tok.float(), torch.float32), (0.5, 1234), r)
    assert abs(float(fp.cpu()[mb].std()) - 0.5) < 0.005
    assert float(fp.cpu()[mb].abs().max()) < 0.5 * 7


def test_ema_adam_cast(ops):
    n = 100003
    g = torch.Generator().manual_seed(0)
    e, p = torch.randn(n + 1, generator=g)[:n], torch.randn(n + 1, generator=g)[:n]
    ed = e.cuda().clone()
    # 16-B alignment of the flat buffers is part of the contract
    ops.ema_update(ed, p.cuda(), 0.9997)
    assert rel_err(ed, e * 0.9997 + p * (1 - 0.9997)) < 1e-6
    pp, gg = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = pp.cuda().clone(), m.cuda(), v.cuda()
    ref = pp.clone()
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.98, 1e-6, 0.01
    for step in (1, 2):
        ops.adam_step(pd, gg.cuda(), md, vd, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step)
        m = m * b1 + gg * (1 - b1)
        v = v * b2 + gg * gg * (1 - b2)
        ss = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        ref = ref - wd * lr * ref
        ref = ref - ss * m / (v.sqrt() + eps)
    assert rel_err(pd, ref) < 1e-5
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.sumsq(gg.cuda(), out)
    assert abs(float(out) - float((gg.double() ** 2).sum())) / float(out) < 1e-6
    a = torch.randn(37, 64, generator=g)
    b = torch.zeros(37, 96, dtype=torch.bfloat16, device="cuda")
    ops.cast2d(a.cuda(), b[:, :64])
    assert torch.equal(b[:, :64].float().cpu(), a.to(torch.bfloat16).float()) and float(b[:, 64:].float().abs().max()) == 0


# ------------------------------------------------------------------------------------------ dropout
@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_sites_regenerate_one_mask(ops, dtype):
    """Counter-based dropout: every kernel recomputes keep(seed, linear index); a2v_dropout_mask exposes it."""
    p_, seed = 0.1, 1234567
    M, N, K = 192, 128, 64
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b, r = rnd(N, dtype=torch.float32, seed=3), rnd(M, N, dtype=dtype, seed=4)
    keep = ops.dropout_mask(M * N, p_, seed).view(M, N).float().cpu()
    assert 0.87 < float(keep.mean()) < 0.93
    y = ops.linear(dev(x, dtype), dev(w, dtype), dev(b, torch.float32), res=dev(r, dtype), drop=(p_, seed))
    ref = (x @ w.t() + b) * keep / (1 - p_) + r
    assert rel_err(y.float(), ref) < tol(dtype)
    # LayerNorm backward: mode 2 returns (dx, dropout(dx)) with the SAME mask, dxsum sums the masked one
    xs, dy = rnd(M, N, dtype=dtype, seed=5), rnd(M, N, dtype=dtype, seed=6)
    g_, b_ = torch.ones(N).cuda(), torch.zeros(N).cuda()
    _, mean, rstd = ops.layernorm_fwd(dev(xs, dtype), g_, b_, 1e-5)
    dxs = torch.zeros(N, device="cuda")
    dx, dxd = ops.layernorm_bwd(dev(dy, dtype), dev(xs, dtype), g_, b_, mean, rstd, drop=(p_, seed), drop_mode=2, dxsum=dxs)
    assert rel_err(dxd.float().cpu(), dx.float().cpu() * keep / (1 - p_)) < tol(dtype)
    assert rel_err(dxs, dxd.float().sum(0)) < 1e-3
    # mode 1: y = drop(LN x)  ->  dx = LNbwd(dy * mask)
    yf, mean, rstd = ops.layernorm_fwd(dev(xs, dtype), g_, b_, 1e-5, drop=(p_, seed))
    assert rel_err(yf.float().cpu(), F.layer_norm(xs, (N,)) * keep / (1 - p_)) < tol(dtype)
    dx1 = ops.layernorm_bwd(dev(dy, dtype), dev(xs, dtype), g_, b_, mean, rstd, drop=(p_, seed), drop_mode=1)
    dym = (dy * keep / (1 - p_)).to(dtype).float()
    dx_ref = ops.layernorm_bwd(dev(dym, dtype), dev(xs, dtype), g_, b_, mean, rstd)
    assert rel_err(dx1.float(), dx_ref.float()) < tol(dtype, 2)
    # token gather / decoder input
    Nn, Kk, T, D = 4, 6, 30, 64
    tok = rnd(Nn, Kk, D, dtype=dtype, seed=7)
    mask = O.seeded_mask(Nn, T, Kk, 2)
    keep_ids, restore = O.make_maskinfo(mask)
    noise = rnd(Nn, T - Kk, D, dtype=torch.float32, seed=8)
    km = ops.dropout_mask(Nn * Kk * D, p_, seed).view(Nn, Kk, D).float().cpu()
    out = ops.decoder_input(dev(tok, dtype), noise.cuda(), restore.int().cuda(), drop=(p_, seed))
    ref = O.decoder_input((tok * km / (1 - p_)).to(dtype).float(), restore, noise)
    assert rel_err(out.float(), ref.to(dtype).float()) < 1e-5
    full = rnd(Nn, T, D, dtype=dtype, seed=9)
    gth = ops.gather_tokens(dev(full, dtype), keep_ids.int().cuda(), cb=1, drop=(p_, seed))
    gi = keep_ids.unsqueeze(-1).expand(-1, -1, D)
    assert rel_err(gth.float(), (torch.gather(full, 1, gi) * km / (1 - p_)).to(dtype).float()) < 1e-5


@pytest.mark.parametrize("dtype,hd", [(torch.float32, 16), (torch.bfloat16, 64)])
def test_attention_dropout(ops, dtype, hd):
    N, H, S, p_, seed = 2, 2, 100, 0.1, 99
    qkv = rnd(N, S, 3 * H * hd, dtype=dtype, seed=1).requires_grad_(True)
    pos = torch.arange(S).unsqueeze(0).expand(N, -1).int().contiguous()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = torch.tensor([1.0, 0.7], requires_grad=True)
    keep = ops.dropout_mask2d(N * H * S, S, p_, seed).view(N, H, S, S).float().cpu()
    q, k, v = qkv.reshape(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, sc.view(1, 1, H, 1, 1))
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1) + bias).softmax(-1) * keep / (1 - p_)
    ref = (att @ v).transpose(1, 2).reshape(N, S, H * hd)
    do = rnd(N, S, H * hd, dtype=dtype, seed=3)
    ref.backward(do)
    qd = dev(qkv.detach(), dtype)
    out, lse = ops.attention_fwd(qd, pos.cuda(), slopes.cuda(), sc.detach().cuda(), H, drop=(p_, seed))
    assert rel_err(out.float(), ref) < tol(dtype, 2)
    dal = torch.zeros(H, device="cuda")
    bg = torch.zeros(3 * H * hd, device="cuda")
    dqkv = ops.attention_bwd(qd, out, dev(do, dtype), lse, pos.cuda(), slopes.cuda(), sc.detach().cuda(), H, dal, drop=(p_, seed),
                             bias_grad=bg)
    assert rel_err(dqkv.float(), qkv.grad) < tol(dtype, 3)
    assert rel_err(bg, dqkv.float().sum((0, 1))) < 1e-4        # bias gradient = column sums of what was written
    assert rel_err(dal, sc.grad) < max(tol(dtype, 3), 5e-4)


def _unpack_dropmask(dm, N, H, S):
    """keep[n, h, q, key] from the forward's bit mask (csrc/a2v_attn_seq2.h): per (n, h) 1280 words [word][lane quarter g][row q];
    the lane (q, g) holds keys 16 j + 4 g + r at bit 31 - (4 j + r) of word 0 (j < 8) / bit 7 - (4 (j - 8) + r) of word 1 (j = 8, 9)."""
    import numpy as np
    w = dm.view(N, H, 2, 4, 160).cpu().numpy().astype("uint32")
    keep = torch.zeros(N, H, S, 160, dtype=torch.bool)
    for j in range(10):
        for g in range(4):
            for r in range(4):
                word = w[:, :, 0 if j < 8 else 1, g, :S]
                bit = (31 if j < 8 else 7) - (4 * (j & 7) + r)
                keep[:, :, :, 16 * j + 4 * g + r] = torch.from_numpy(((word >> np.uint32(bit)) & np.uint32(1)).astype("bool"))
    return keep[:, :, :, :S]


@pytest.mark.parametrize("S", [37, 96, 140, 152, 155, 160])
def test_attention_seq2_dropout_mask_path(ops, S, request):
    """Round-6 whole-sequence kernels (bf16, HD 64, S <= 160): the forward's bit mask IS the counter hash's decision; a backward that
    reads the mask, one that re-hashes, and the round-5 kernels agree; forward / backward against the fp32 reference with that mask."""
    dtype, hd = torch.bfloat16, 64
    N, H, p_, seed = 3, 4, 0.1, 4242
    qkv = rnd(N, S, 3 * H * hd, dtype=dtype, seed=11).requires_grad_(True)
    g = torch.Generator().manual_seed(5)
    pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = (torch.rand(H, generator=g) + 0.5).requires_grad_(True)
    keep = ops.dropout_mask2d(N * H * S, S, p_, seed).view(N, H, S, S).float().cpu()
    q, k, v = qkv.reshape(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, sc.view(1, 1, H, 1, 1))
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1) + bias).softmax(-1) * keep / (1 - p_)
    ref = (att @ v).transpose(1, 2).reshape(N, S, H * hd)
    do = rnd(N, S, H * hd, dtype=dtype, seed=3)
    ref.backward(do)
    qd, dod = dev(qkv.detach(), dtype), dev(do, dtype)
    args = (pos.cuda(), slopes.cuda(), sc.detach().cuda(), H)
    ops.debug_set(39, 2)                                  # every S <= 160 through the round-6 kernels (default: 145 .. 160 only)
    request.addfinalizer(lambda: ops.debug_set(39, 1))
    out, lse = ops.attention_fwd(qd, *args, drop=(p_, seed))
    dm = getattr(lse, "_a2v_dropmask", None)
    assert dm is not None, "the whole-sequence forward did not return a dropout mask"
    assert torch.equal(_unpack_dropmask(dm, N, H, S), keep.bool())
    assert rel_err(out.float(), ref) < tol(dtype, 2)

    def bwd(lse_):
        dal = torch.zeros(H, device="cuda")
        bg = torch.zeros(3 * H * hd, device="cuda")
        d = ops.attention_bwd(qd, out, dod, lse_, *args, dal, drop=(p_, seed), bias_grad=bg)
        return d, dal, bg
    d_mask, dal_mask, bg_mask = bwd(lse)
    lse_plain = lse.clone()                               # (no mask attribute: the backward re-hashes)
    d_hash, dal_hash, _ = bwd(lse_plain)
    assert torch.equal(d_mask, d_hash)
    assert rel_err(dal_mask, dal_hash) < 1e-6
    assert rel_err(d_mask.float(), qkv.grad) < tol(dtype, 3)
    assert rel_err(bg_mask, d_mask.float().sum((0, 1))) < 1e-4
    assert rel_err(dal_mask, sc.grad) < max(tol(dtype, 3), 5e-4)
    ops.debug_set(39, 0)                                  # the round-5 kernels on the same inputs
    try:
        out5, lse5 = ops.attention_fwd(qd, *args, drop=(p_, seed))
        assert getattr(lse5, "_a2v_dropmask", None) is None
        d5, dal5, _ = bwd(lse5)
    finally:
        ops.debug_set(39, 2)
    assert rel_err(out.float(), out5.float()) < 4e-3 and rel_err(lse[0], lse5[0]) < 1e-5 and rel_err(lse[1], lse5[1]) < 1e-4
    assert rel_err(d_mask.float(), d5.float()) < 8e-3
    assert rel_err(dal_mask, dal5) < 2e-2


@pytest.mark.parametrize("S,drop", [(152, True), (147, False), (160, True)])
def test_attention_seq2_many_sequences(ops, S, drop):
    """The round-6 kernels are PERSISTENT: one workgroup per CU walks several (n, h) sequences through alternating LDS buffers and
    constant sets.  Only a launch with more than 256 sequences exercises that (every other test here has at most 12): 44 x 16 = 704
    sequences -- two or three per workgroup -- against the round-5 one-shot kernels on the same inputs (independently held to the
    fp32 reference above), forward outputs, log-sum-exp, all three gradients, the fused bias gradient and d alibi_scale; and every
    sequence against the fp32 reference for a sample of rows."""
    dtype, hd, N, H = torch.bfloat16, 64, 44, 16
    g = torch.Generator().manual_seed(17)
    qkv = torch.randn(N, S, 3 * H * hd, generator=g).to(dtype)
    pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = torch.rand(H, generator=g) + 0.5
    do = torch.randn(N, S, H * hd, generator=g).to(dtype)
    qd, dod = qkv.cuda(), do.cuda()
    args = (pos.cuda(), slopes.cuda(), sc.cuda(), H)
    dr = (0.1, 777) if drop else None

    def both():
        out, lse = ops.attention_fwd(qd, *args, drop=dr)
        dal = torch.zeros(H, device="cuda")
        bg = torch.zeros(3 * H * hd, device="cuda")
        d = ops.attention_bwd(qd, out, dod, lse, *args, dal, drop=dr, bias_grad=bg)
        torch.cuda.synchronize()
        return out, lse, d, dal, bg
    o6, l6, d6, a6, b6 = both()
    assert (getattr(l6, "_a2v_dropmask", None) is not None) == drop
    ops.debug_set(39, 0)
    try:
        o5, l5, d5, a5, b5 = both()
    finally:
        ops.debug_set(39, 1)
    assert torch.isfinite(o6.float()).all() and torch.isfinite(d6.float()).all()
    assert rel_err(o6.float(), o5.float()) < 4e-3 and rel_err(l6[0], l5[0]) < 1e-5 and rel_err(l6[1], l5[1]) < 1e-4
    # per sequence: no (n, h) may be off (a stale buffer or constant set would hit whole sequences, not the norm of the tensor)
    e = ((o6.float() - o5.float()).view(N, S, H, hd).pow(2).sum((1, 3)) / o5.float().view(N, S, H, hd).pow(2).sum((1, 3))).sqrt()
    assert float(e.max()) < 8e-3, (float(e.max()), int(e.argmax()))
    W = 3 * H * hd
    eg = ((d6.float() - d5.float()).view(N, S, 3, H, hd).pow(2).sum((1, 4)) / d5.float().view(N, S, 3, H, hd).pow(2).sum((1, 4))).sqrt()
    assert float(eg.max()) < 2e-2, (float(eg.max()), int(eg.argmax()))
    assert rel_err(d6.float(), d5.float()) < 8e-3
    assert rel_err(b6, d6.float().sum((0, 1))) < 1e-4 and rel_err(b6, b5) < 5e-3
    assert rel_err(a6, a5) < 2e-2
    # a sample of sequences against the fp32 reference (with the kernel's own mask)
    keep = None
    if drop:
        keep = ops.dropout_mask2d(N * H * S, S, dr[0], dr[1]).view(N, H, S, S).float().cpu()
    for n in (0, 17, 43):
        q, k, v = qkv[n:n + 1].float().reshape(1, S, 3, H, hd).permute(2, 0, 3, 1, 4)
        bias = O.alibi_bias_from_positions(pos[n:n + 1].long(), H, sc.view(1, 1, H, 1, 1))
        att = ((q * hd ** -0.5) @ k.transpose(-2, -1) + bias).softmax(-1)
        if drop:
            att = att * keep[n:n + 1] / (1 - dr[0])
        ref = (att @ v).transpose(1, 2).reshape(1, S, H * hd)
        assert rel_err(o6[n:n + 1].float().cpu(), ref) < tol(dtype, 2), n


def test_attention_dropout_mask_statistics(ops):
    """2-D pair hash of the attention-probability site: keep rate 1 - p, no correlation inside a pair, between
    neighbouring pairs, between rows or between seeds."""
    rows, cols, p_ = 4096, 600, 0.2
    k = ops.dropout_mask2d(rows, cols, p_, 7).float()
    assert abs(float(k.mean()) - (1 - p_)) < 2e-3
    z = k - k.mean()
    var = float((z * z).mean())
    for a_, b_ in ((z[:, 0::2], z[:, 1::2]), (z[:, 1:-1:2], z[:, 2::2]), (z[:-1], z[1:]), (z[:, :-2], z[:, 2:])):
        assert abs(float((a_ * b_).mean())) / var < 5e-3
    k2 = ops.dropout_mask2d(rows, cols, p_, 8).float()
    assert abs(float(((k2 - k2.mean()) * z).mean())) / var < 5e-3
    assert abs(float(k.mean(0).std()) - (p_ * (1 - p_) / rows) ** 0.5) < 2e-3      # column means spread like binomials


@pytest.mark.parametrize("S,p_", [(200, 0.0), (333, 0.2), (700, 0.0), (641, 0.1)])
def test_attention_bwd_long_two_pass(ops, S, p_):
    """bf16, HD = 64, S > 192: the two-pass backward (dK/dV pass + dQ pass, no atomics) against torch autograd and
    against the tiled kernel with dQ atomics (a2v_debug_set(8, 0)); ragged S (tails in both the owner and the
    streamed dimension), sorted random positions, attention dropout."""
    from animal2vec_amd import _lib
    dtype, hd, N, H, seed = torch.bfloat16, 64, 2, 3, 17
    qkv = rnd(N, S, 3 * H * hd, dtype=dtype, seed=1).requires_grad_(True)
    g = torch.Generator().manual_seed(2)
    pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = (torch.rand(H, generator=g) + 0.5).requires_grad_(True)
    q, k, v = qkv.reshape(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, sc.view(1, 1, H, 1, 1))
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1) + bias).softmax(-1)
    drop = None
    if p_ > 0:
        keep = ops.dropout_mask2d(N * H * S, S, p_, seed).view(N, H, S, S).float().cpu()
        att = att * keep / (1 - p_)
        drop = (p_, seed)
    ref = (att @ v).transpose(1, 2).reshape(N, S, H * hd)
    do = rnd(N, S, H * hd, dtype=dtype, seed=3)
    ref.backward(do)
    qd, dod = dev(qkv.detach(), dtype), dev(do, dtype)
    args = (pos.cuda(), slopes.cuda(), sc.detach().cuda(), H)
    out, lse = ops.attention_fwd(qd, *args, drop=drop)
    dal = torch.zeros(H, device="cuda")
    dqkv = ops.attention_bwd(qd, out, dod, lse, *args, dal, drop=drop)
    assert rel_err(dqkv.float(), qkv.grad) < tol(dtype, 3)
    assert rel_err(dal, sc.grad) < max(tol(dtype, 3), 5e-4)
    for name, sl in (("dq", slice(0, H * hd)), ("dk", slice(H * hd, 2 * H * hd)), ("dv", slice(2 * H * hd, 3 * H * hd))):
        assert rel_err(dqkv[..., sl].float(), qkv.grad[..., sl]) < tol(dtype, 3), name
    _lib.load().a2v_debug_set(8, 0)
    try:
        dal2 = torch.zeros(H, device="cuda")
        dqkv2 = ops.attention_bwd(qd, out, dod, lse, *args, dal2, drop=drop)
    finally:
        _lib.load().a2v_debug_set(8, 1)
    assert rel_err(dqkv.float(), dqkv2.float()) < 1e-2 and rel_err(dal, dal2) < 1e-3


def _alibi_well_conditioned_case(N, H, S, hd, dtype, T=2000, seed=0, arange=False):
    """Inputs for which d loss / d alibi_scale does NOT cancel: values and output gradients are smooth Fourier features of the
    frame position (lowest period 250 frames), so dO_i . V_j falls with |pos_i - pos_j| for every row -- each row's term of the
    gradient is a positive multiple of the variance of the attended distance.  (At random init the model's own gradient is a sum
    of 10^5 .. 10^8 terms of random sign: condition numbers 20 .. 673, tests/test_pretrain_gpu.py::alibi_term_scale.)"""
    g = torch.Generator().manual_seed(seed)
    if arange:
        pos = torch.arange(S).unsqueeze(0).expand(N, -1).contiguous()
    else:
        pos = torch.stack([torch.randperm(T, generator=g)[:S].sort()[0] for _ in range(N)])
    w = 2 * math.pi * (torch.arange(hd // 2) + 1.0) / 8000.0
    ph = pos[:, :, None].float() * w
    feat = torch.cat([ph.cos(), ph.sin()], -1)[:, :, None, :].expand(N, S, H, hd)
    q = torch.randn(N, S, H, hd, generator=g) * 0.3
    k = torch.randn(N, S, H, hd, generator=g) * 0.3
    v = feat + torch.randn(N, S, H, hd, generator=g) * 0.1
    do = feat * 0.05 + torch.randn(N, S, H, hd, generator=g) * 0.005
    qkv = torch.stack([q, k, v], 2).reshape(N, S, 3 * H * hd).to(dtype).float()
    return pos, qkv, do.reshape(N, S, H * hd).to(dtype).float()


@pytest.mark.parametrize("dtype,N,S,arange,bound", [(torch.bfloat16, 96, 140, False, 5e-2), (torch.bfloat16, 96, 152, False, 5e-2),
                                                    (torch.bfloat16, 1, 2000, True, 5e-2), (torch.float32, 8, 140, False, 1e-4)])
def test_alibi_scale_gradient_well_conditioned_full_dims(ops, dtype, N, S, arange, bound):
    """d alibi_scale at the large model's attention dims (16 heads x 64, the student's 96 sequences of 140 / 152
    kept frames of 2000 -- the whole-sequence kernels -- and a 2000-frame sequence -- the two-pass long kernels of the finetune
    path) on inputs whose gradient is not cancellation-dominated.  The construction is checked here (condition number <= 30 from
    the fp64 reference's own terms), then the RELATIVE error of the kernel's gradient is held to 5e-2 in bf16 (1e-4 in fp32):
    the 6 % / 24 % relative errors of the model-level comparisons are condition number x bf16 input rounding, not the kernel."""
    H, hd = 16, 64
    pos, qkv, do = _alibi_well_conditioned_case(N, H, S, hd, dtype, arange=arange)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float64)
    x = qkv.double().requires_grad_(True)
    sc = torch.ones(H, dtype=torch.float64, requires_grad=True)
    q, k, v = x.reshape(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    dist = (pos[:, :, None] - pos[:, None, :]).abs().double().unsqueeze(1)                  # (N,1,S,S)
    sco = (q * hd ** -0.5) @ k.transpose(-2, -1) - (slopes * sc.clamp_min(0)).view(1, H, 1, 1) * dist
    sco.retain_grad()
    out_ref = (sco.softmax(-1) @ v).transpose(1, 2).reshape(N, S, H * hd)
    out_ref.backward(do.double())
    A = (sco.grad.abs() * slopes.view(1, H, 1, 1) * dist).sum((0, 2, 3))                    # sum of |terms| per head
    cond = float(A.norm() / sc.grad.norm())
    assert cond <= 30, cond
    qd = dev(qkv, dtype)
    pd = None if arange else pos.int().cuda()
    args = (pd, slopes.float().cuda(), torch.ones(H, device="cuda"), H)
    out, lse = ops.attention_fwd(qd, *args)
    dal = torch.zeros(H, device="cuda")
    dqkv = ops.attention_bwd(qd, out, dev(do, dtype), lse, *args, dal)
    rel = rel_err(dal.double().cpu(), sc.grad)
    assert rel < bound, (rel, cond)
    assert rel_err(dqkv.double().cpu(), x.grad) < tol(dtype, 3)
    # per head too: no head may be off by more than 2.5 x the bound (the steep heads attend to 2 - 3 frames)
    worst = float(((dal.double().cpu() - sc.grad).abs() / sc.grad.abs()).max())
    from conftest import parity_report
    parity_report(f"alibi_well_conditioned_N{N}_S{S}", dtype, [(rel, "d_alibi_scale"), (worst, "d_alibi_scale_worst_head")], alibi_cond=cond)
    assert worst < 2.5 * bound, worst


def test_mixup_golden(ops):
    """BC mixup against the vector dumped from the live reference (oracle/gen_golden.py: G14)."""
    from conftest import load_golden
    mx = load_golden("mixup.npz")
    m = ops.Mixup(8000, 0.05, "cuda")
    src = mx["source"].cuda()
    g = m.gain_db(src)
    assert float((g.cpu() - mx["gain_db"]).abs().max()) < 2e-3          # dB
    out = m(src, mx["perm"].int().cuda(), float(mx["r"]))
    assert rel_err(out, mx["mixed"]) < 1e-4


def test_mixup_partial(ops):
    """mixup_prob < 1 (data2vec2.py:540-547,584-589): only the flagged clips are mixed -- each with x[perm[b]], gains
    taken over the whole batch -- the others pass through unchanged."""
    g = torch.Generator().manual_seed(4)
    B, L = 6, 8000
    x = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,)) * (0.5 + torch.rand(B, 1, generator=g))
    perm = torch.randperm(B, generator=g)
    mm = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.uint8)
    r = 0.7
    ref_all = O.bc_mixup(x, perm, r)
    ref = torch.where(mm.bool().view(-1, 1), ref_all, x)
    m = ops.Mixup(8000, 0.05, "cuda")
    out = m(x.cuda(), perm.int().cuda(), r, mm.cuda())
    assert rel_err(out, ref) < 1e-4
    assert torch.equal(out[1].cpu(), x[1]) and torch.equal(out[4].cpu(), x[4])
    assert rel_err(m(x.cuda(), perm.int().cuda(), r), ref_all) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_frames(ops, dtype):
    """apply_mask (base.py:457-485): replaced frames ~ N(0, std), zeroed channels, untouched elsewhere; the backward
    form zeroes exactly the replaced positions."""
    B, T, D = 3, 200, 64
    x = dev(rnd(B, T, D, dtype=dtype, seed=1), dtype)
    g = torch.Generator().manual_seed(2)
    mt = (torch.rand(B, T, generator=g) < 0.6).to(torch.uint8).cuda()
    mc = (torch.rand(B, D, generator=g) < 0.3).to(torch.uint8).cuda()
    y = ops.mask_frames(x, mt, mc, noise_std=0.25, seed=5)
    keep_c = (mc == 0).unsqueeze(1).expand(B, T, D)
    mtb = mt.bool().unsqueeze(-1).expand(B, T, D)
    assert torch.equal(y[~mtb & keep_c], x[~mtb & keep_c])
    assert float(y[~keep_c].abs().max()) == 0.0
    noise = y[mtb & keep_c].float()
    assert abs(float(noise.mean())) < 0.01 and abs(float(noise.std()) - 0.25) < 0.01
    assert abs(float((noise.abs() > 0.5).float().mean()) - 0.0455) < 0.01          # two-sigma tail of a gaussian
    assert torch.equal(y, ops.mask_frames(x, mt, mc, noise_std=0.25, seed=5))         # deterministic in the seed
    assert not torch.equal(y, ops.mask_frames(x, mt, mc, noise_std=0.25, seed=6))
    z = ops.mask_frames(x, mt, None, noise_std=0.0)                                   # encoder_zero_mask
    assert float(z[mtb].abs().max()) == 0.0 and torch.equal(z[~mtb], x[~mtb])
    d = ops.mask_frames(x, mt, mc, backward=True)
    assert torch.equal(d, torch.where(~mtb & keep_c, x, torch.zeros_like(x)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_add_drop(ops, dtype):
    n, D = 777, 64
    a, b, c = (dev(rnd(n, D, dtype=dtype, seed=s), dtype) for s in (1, 2, 3))
    keep = ops.dropout_mask(n * D, 0.2, 31).view(n, D).float()
    ref = a.float() + (b.float() + c.float()).to(dtype).float() * keep / 0.8 if dtype == torch.float32 else None
    out = ops.add_drop(a, b, c, drop=(0.2, 31))
    exp = a.float() + (b.float() + c.float()) * keep / 0.8
    assert rel_err(out.float(), exp) < tol(dtype)
    assert rel_err(ops.add_drop(None, b, None, drop=(0.2, 31)).float(), b.float() * keep / 0.8) < tol(dtype)
    assert rel_err(ops.add_drop(a, b).float(), a.float() + b.float()) < tol(dtype)


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (2048, 512, 256)])
def test_linear_gelu_dropout_epilogues(ops, M, N, K):
    """The Mlp's activation dropout (timm Mlp drop1): fc1 epilogue GELU then dropout; its backward: the fc2 input
    gradient epilogue applies the same mask and GELU'."""
    dtype = torch.bfloat16
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, seed=3)
    xd, wd, bd = dev(x, dtype), dev(w, dtype), dev(b, torch.float32)
    keep = ops.dropout_mask(M * N, 0.1, 55).view(M, N).float().cpu()
    pre_out = torch.empty(M, N, dtype=dtype, device="cuda")
    y = ops.linear(xd, wd, bd, gelu=True, pre_out=pre_out, drop=(0.1, 55))
    pre = x @ w.t() + b
    assert rel_err(y.float(), F.gelu(pre) * keep / 0.9) < tol(dtype)
    assert rel_err(pre_out.float(), pre) < tol(dtype)
    # backward epilogue: dh = drop(dy @ W2^T) * gelu'(hpre)
    dy, w2t = rnd(M, K, dtype=dtype, seed=4), rnd(N, K, dtype=dtype, seed=5, scale=K ** -0.5)
    hpre = rnd(M, N, dtype=dtype, seed=6)
    cs = torch.zeros(N, dtype=torch.float32, device="cuda")
    dh = ops.linear(dev(dy, dtype), dev(w2t, dtype), None, gelu_bwd_aux=dev(hpre, dtype), colsum=cs, drop=(0.1, 55))
    hp = hpre.clone().requires_grad_(True)
    F.gelu(hp).sum().backward()
    exp = (dy @ w2t.t()) * keep / 0.9 * hp.grad
    assert rel_err(dh.float(), exp) < tol(dtype)
    assert rel_err(cs, dh.float().sum(0)) < 1e-3


def test_adam_lr_scale_table(ops):
    """per-64-parameter lr_scale (composite optimizer lr_scale, e.g. layer_decay): scales both the decay and the step"""
    n = 64 * 37
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g)
    ls = torch.rand(n // 64, generator=g)
    p, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lr, b1, b2, eps, wd = 1e-2, 0.9, 0.98, 1e-8, 0.1
    ops.adam_step(p, gr.cuda(), m, v, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=1, lr_scale=ls.cuda())
    lse = ls.repeat_interleave(64)
    m1, v1 = (1 - b1) * gr, (1 - b2) * gr * gr
    step_size = lr * math.sqrt(1 - b2) / (1 - b1)
    exp = p0 - wd * lr * lse * p0 - step_size * lse * m1 / (v1.sqrt() + eps)
    assert rel_err(p, exp) < 1e-5


def test_streams_have_their_own_scratch(ops):
    """Re-entrancy across streams (include/a2v.h): the persistent GEMM's tile counters, its column-sum partial rows and
    the attention backward's d alibi_scale partials exist once per stream, so the same launches issued concurrently on
    two streams give bit for bit what they give one after the other (a shared set loses tiles or mixes partial sums)."""
    from animal2vec_amd import _lib
    dtype = torch.bfloat16
    shapes = [(13440, 1024, 512), (16000, 2048, 512)]

    def gemm_work(seed, M, N, K):
        x, w = dev(rnd(M, K, dtype=dtype, seed=seed), dtype), dev(rnd(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5), dtype)
        aux = dev(rnd(M, N, dtype=dtype, seed=seed + 2), dtype)
        return x, w, aux

    def attn_work(seed, N_=6, S=140, H=4):
        qkv = dev(rnd(N_, S, 3 * H * 64, dtype=dtype, seed=seed, scale=0.5), dtype)
        do = dev(rnd(N_, S, H * 64, dtype=dtype, seed=seed + 1), dtype)
        pos = torch.arange(S, device="cuda", dtype=torch.int32).repeat(N_, 1).contiguous()
        return qkv, do, pos, torch.rand(H, device="cuda") + 0.1, torch.ones(H, device="cuda"), H

    works = [gemm_work(10 + 10 * i, *shapes[i]) for i in range(2)]
    aws = [attn_work(50 + 10 * i) for i in range(2)]

    def run(i, reps=6):
        x, w, aux = works[i]
        outs = []
        for _ in range(reps):
            cs = torch.zeros(w.shape[0], dtype=torch.float32, device="cuda")
            y = ops.linear(x, w, None, gelu_bwd_aux=aux, colsum=cs)
            outs.append((y, cs))
        qkv, do, pos, slopes, asc, H = aws[i]
        att, lse = ops.attention_fwd(qkv, pos, slopes, asc, H)
        dal = torch.zeros(H, dtype=torch.float32, device="cuda")
        dqkv = ops.attention_bwd(qkv, att, do, lse, pos, slopes, asc, H, dal)
        return outs, dqkv, dal

    serial = [run(0), run(1)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    conc = [None, None]
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                conc[i] = run(i)
        torch.cuda.synchronize()
        for i in (0, 1):
            for (y, cs), (y0, cs0) in zip(conc[i][0], serial[i][0]):
                assert torch.equal(y, y0) and rel_err(cs, cs0) < 1e-5, (rep, i)     # (the fold's 16 slices add atomically: order varies)
            assert torch.equal(conc[i][1], serial[i][1]) and torch.equal(conc[i][2], serial[i][2]), (rep, i)
    for s_ in streams:
        assert _lib.load().a2v_stream_release(s_.cuda_stream) == 0
    assert _lib.load().a2v_stream_release(streams[0].cuda_stream) != 0       # already released


@pytest.mark.parametrize("CI,CO", [(48, 48), (64, 48)])
@pytest.mark.parametrize("B,T", [(3, 2000), (2, 1100), (5, 64), (1, 333)])
def test_conv_decoder_kernel(ops, B, T, CI, CO):
    """Decoder1d's grouped convs (k 7, groups 16: 768 -> 768 with 48 channels per group, and the first layer 1024 -> 768)
    through the HBM-bound all-groups kernel (a2v_conv_dec.hip): forward with bias, input gradient with the residual add
    (48 -> 48, and 48 -> 64 for the first layer's), clip edges, ragged row blocks; against fp32 torch and against the slab
    / implicit-GEMM kernels (a2v_debug_set(13, 0))."""
    from animal2vec_amd import _lib
    dtype = torch.bfloat16
    G, k = 16, 7
    Din, Dout = G * CI, G * CO
    x = rnd(B, Din, T, dtype=dtype, seed=1)
    w = rnd(Dout, CI, k, dtype=dtype, seed=2, scale=(CI * k) ** -0.5)
    bias = rnd(Dout, dtype=torch.float32, seed=3)
    dy = rnd(B, Dout, T, dtype=dtype, seed=4)
    res = rnd(B, T, Din, dtype=dtype, seed=5)
    xr = x.clone().requires_grad_(True)
    ref = F.conv1d(xr, w, bias, padding=k // 2, groups=G)
    ref.backward(dy)
    xcl, dycl = to_cl(x, Din, dtype), to_cl(dy, Dout, dtype)
    wp = dev(pack_ref(w, G, 64), dtype)
    wt = torch.empty(G, CI, k * 64, dtype=dtype, device="cuda")
    ops.pack_weight(dev(w, torch.float32), wt, G=G, N=CO, Cg=CI, k=k, Np=64, mode=1)
    rd = dev(res, dtype)

    def run():
        y = ops.conv1d_cl(xcl, wp, dev(bias, torch.float32), k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=Dout, cv=CI)
        dx = ops.conv1d_cl(dycl, wt, None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=Din, cv=CO, res=rd)
        return y, dx

    y, dx = run()
    y2, dx2 = run()
    assert torch.equal(y, y2) and torch.equal(dx, dx2)
    lib = _lib.load()
    lib.a2v_debug_set(13, 0)
    try:
        y0, dx0 = run()
    finally:
        lib.a2v_debug_set(13, 1)
    assert rel_err(y.float().transpose(1, 2), ref.detach()) < tol(dtype)
    assert rel_err(dx.float(), xr.grad.transpose(1, 2) + res) < tol(dtype, 2)
    # K is accumulated as (tap, channel) here and as tap-major 64-wide steps there: same products, fp32 sums in another order
    assert rel_err(y.float(), y0.float()) < 4e-3 and rel_err(dx.float(), dx0.float()) < 4e-3
    # the work order does not touch the arithmetic: the in-order front with three workgroups per channel slice (a workgroup then
    # walks several blocks of different clips, stride 3) and the older order (a workgroup owns consecutive blocks) give the same bits
    for order in (3, -1):
        lib.a2v_debug_set(38, order)
        try:
            yo, dxo = run()
        finally:
            lib.a2v_debug_set(38, 0)
        assert torch.equal(yo, y) and torch.equal(dxo, dx), order


# ------------------------------------------------------------------------------------------ bf16x3 parity mode
def test_split_bf16_images(ops):
    x = torch.randn(300, 136, generator=torch.Generator().manual_seed(1)) * torch.logspace(-3, 3, 136)
    xd = x.cuda()
    h = x.bfloat16().float()
    l = (x - h).bfloat16().float()
    assert float((x - h - l).abs().max() / x.abs().max()) < 2 ** -16
    i0, i1, i2 = ops.split_bf16(xd, 0).float().cpu(), ops.split_bf16(xd, 1).float().cpu(), ops.split_bf16(xd, 2).float().cpu()
    assert torch.equal(i0, torch.cat([h, l, h], 1)) and torch.equal(i1, torch.cat([h, h, l], 1)) and torch.equal(i2, torch.cat([h, l], 0))


@pytest.mark.parametrize("M,N,K", [(13440, 1024, 1024), (2100, 768, 512), (1100, 256, 128), (107520, 1024, 1024)])
def test_linear_bf16x3_matches_fp64(ops, M, N, K):
    """A2V_BF16X3: the Linear product as three exact bf16 products over split operands on the persistent 8-phase kernel, fp32 in /
    out, every epilogue flavour -- against an fp64 product of the SAME fp32 operands.  Bound 3e-5 relative (2^-16 per product
    + fp32 accumulation; the fp32 MFMA mode measures 1e-6, the bf16 mode 4e-3)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    xd, wd, bd, rd = x.cuda(), w.cuda(), b.cuda(), r.cuda()
    big = M > 50000
    ref = (xd.double() @ wd.double().t()).cpu() if big else x.double() @ w.double().t()
    ops.x3_register(wd)
    try:
        y = ops.linear(xd, wd, bd)
        assert rel_err(y, ref + b.double()) < 3e-5
        if big:
            return
        y0 = ops.linear(xd[:512], wd, bd)                      # M < 1024: stays on the fp32 kernel
        assert rel_err(y0, (ref + b.double())[:512]) < 1e-5
        pre = torch.empty(M, N, device="cuda")
        yg = ops.linear(xd, wd, bd, gelu=True, pre_out=pre)
        assert rel_err(pre, ref + b.double()) < 3e-5 and rel_err(yg, F.gelu(ref + b.double())) < 3e-5
        keep = torch.empty(M, N, device="cuda")
        yd = ops.linear(xd, wd, bd, gelu=True, pre_out=keep, dact=True)
        pg = (ref + b.double()).clone().requires_grad_(True)
        F.gelu(pg).sum().backward()
        assert rel_err(yd, F.gelu(ref + b.double())) < 3e-5 and rel_err(keep, pg.grad) < 3e-5
        yr = ops.linear(xd, wd, bd, res=rd)
        assert rel_err(yr, ref + b.double() + r.double()) < 3e-5
        cs = torch.zeros(N, device="cuda")
        yb = ops.linear(xd, wd, None, gelu_bwd_aux=keep, colsum=cs, dact=True)
        assert rel_err(yb, ref * keep.double().cpu()) < 3e-5 and rel_err(cs, yb.double().sum(0)) < 1e-5
        ydp = ops.linear(xd, wd, bd, res=rd, drop=(0.1, 77))
        keepm = ops.dropout_mask(M * N, 0.1, 77).view(M, N).cpu().double()
        assert rel_err(ydp, (ref + b.double()) * keepm / 0.9 + r.double()) < 3e-5
    finally:
        del wd._a2v_x3


def test_linear_bf16x3_image_only_output(ops):
    """A2V_EPI_X3IMG: fc1's result (bias + GELU, with the derivative store) leaves the epilogue as its [hi | lo | hi] image only
    (ops.X3Image): the image is the split of what the fp32-output launch writes, bit for bit, the derivative store is unchanged, and
    the two products that read it (fc2 forward, fc2 weight gradient) give what they give from the fp32 tensor."""
    M, K, N = 4096, 256, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, K, generator=g).cuda()
    w1 = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    b1 = torch.randn(N, generator=g).cuda()
    w2 = (torch.randn(K, N, generator=g) * N ** -0.5).cuda()
    dy = torch.randn(M, K, generator=g).cuda()
    ops.x3_register(w1)
    ops.x3_register(w2)
    ops.X3_WGRAD = True
    try:
        pre_a, pre_b = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        h32 = ops.linear(x, w1, b1, gelu=True, pre_out=pre_a, dact=True)
        himg = ops.linear(x, w1, b1, gelu=True, pre_out=pre_b, dact=True, out_image=True)
        assert isinstance(himg, ops.X3Image) and himg.shape == (M, N)
        assert torch.equal(himg.img, ops.split_bf16(h32, 0)) and torch.equal(pre_a, pre_b)
        y32, yimg = ops.linear(h32, w2), ops.linear(himg, w2)
        assert torch.equal(y32, yimg)
        ga, gb = torch.zeros(K, N, device="cuda"), torch.zeros(K, N, device="cuda")
        ops.linear_wgrad(h32, dy, ga)
        ops.linear_wgrad(himg, dy, gb)
        assert rel_err(ga, gb) < 1e-6
        # round 4: the fc2 input gradient (GELU_BWD with the derivative tensor, column sums) as an image too
        w2t = w2.t().contiguous()                                    # (N, K): dh = (dy @ w2t^T) * gelu'
        ops.x3_register(w2t)
        ca, cb = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        dh32 = ops.linear(dy, w2t, gelu_bwd_aux=pre_a, colsum=ca, dact=True)
        dhimg = ops.linear(dy, w2t, gelu_bwd_aux=pre_a, colsum=cb, dact=True, out_image=True)
        assert isinstance(dhimg, ops.X3Image) and dhimg.shape == (M, N)
        assert torch.equal(dhimg.img, ops.split_bf16(dh32, 0)) and rel_err(ca, cb) < 1e-6
        # what reads that image -- the paired weight-gradient launch (dh as the dy operand of fc1's product) and fc1's input
        # gradient -- against the same calls on the fp32 tensor; an image where the split-product path cannot take it is an error
        xin = torch.randn(M, K, generator=torch.Generator().manual_seed(9)).cuda()
        g1a, g2a, g1b, g2b = (torch.zeros(K, N, device="cuda"), torch.zeros(N, K, device="cuda"),
                              torch.zeros(K, N, device="cuda"), torch.zeros(N, K, device="cuda"))
        ops.linear_wgrad_pair(h32, dy, g1a, xin, dh32, g2a)
        ops.linear_wgrad_pair(himg, dy, g1b, xin, dhimg, g2b)
        assert rel_err(g1a, g1b) < 1e-6 and rel_err(g2a, g2b) < 1e-6
        w1t = w1.t().contiguous()                                    # (K, N): dx = dh @ w1t^T
        ops.x3_register(w1t)
        assert torch.equal(ops.linear(dh32, w1t), ops.linear(dhimg, w1t))
        del w1t._a2v_x3
        ops.X3_WGRAD = False
        with pytest.raises(RuntimeError):
            ops.linear_wgrad_pair(h32, dy, g1b, xin, dhimg, g2b)
        ops.X3_WGRAD = True
        del w2t._a2v_x3
        del w2._a2v_x3
        with pytest.raises(RuntimeError):
            ops.linear(himg, w2)
    finally:
        ops.X3_WGRAD = False
        for w in (w1, w2):
            if hasattr(w, "_a2v_x3"):
                del w._a2v_x3


def test_linear_wgrad_bf16x3_matches_fp64(ops):
    M, N, K = 13440, 1024, 512
    g = torch.Generator().manual_seed(4)
    x, dy = torch.randn(M, K, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    ref = (dy.double().t() @ x.double()).cpu()
    gw = torch.zeros(N, K, device="cuda")
    ops.X3_WGRAD = True
    try:
        ops.linear_wgrad(x, dy, gw)
        assert rel_err(gw, ref) < 3e-5
        gwa, gwb = torch.zeros(N, K, device="cuda"), torch.zeros(K, N, device="cuda")
        ops.linear_wgrad_pair(x, dy, gwa, dy, x, gwb)
        assert rel_err(gwa, ref) < 3e-5 and rel_err(gwb, ref.t()) < 3e-5
    finally:
        ops.X3_WGRAD = False
    gw32 = torch.zeros(N, K, device="cuda")
    ops.linear_wgrad(x, dy, gw32)
    assert rel_err(gw32, ref) < 1e-5


@pytest.mark.parametrize("Cin,Cout,k,s,pl,G,T,B", [(1024, 1024, 19, 1, 9, 16, 600, 2), (768, 768, 7, 1, 3, 16, 500, 2),
                                                   (128, 512, 10, 5, 3, 1, 4000, 2), (512, 512, 3, 2, 1, 1, 2000, 2)])
def test_conv_bf16x3_matches_fp64(ops, Cin, Cout, k, s, pl, G, T, B):
    """The convolutions of the step (pos-conv g16 k19, decoder g16 k7 with 48-channel taps padded to 64, the strided feature convs)
    in the bf16x3 mode: implicit im2col of the split activation image against split packed weights on the bf16 kernels with fp32
    outputs; weight gradient as three bf16 products.  Against fp64 convolutions of the same fp32 operands: 3e-5."""
    from animal2vec_amd.params import _padk
    g = torch.Generator().manual_seed(5)
    Cg = Cin // G
    N = Cout // G
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cg, k, generator=g) * (Cg * k) ** -0.5
    bias = torch.randn(Cout, generator=g)
    T_out = T if s == 1 else (T + 2 * pl - k) // s + 1
    ref = conv_ref(x.double(), w.double(), k, s, pl, G, T_out) + bias.double().view(1, -1, 1)
    cgp = _padk(Cg, torch.bfloat16) if G > 1 else Cg
    xcl = to_cl(x, Cin, torch.float32)
    wp = dev(pack_ref(w, G, cgp), torch.float32)
    assert ops.x3_register_conv_pack(wp, k, Cg) is not None
    y = ops.conv1d_cl(xcl, wp, bias.cuda(), k=k, stride=s, pad=pl, groups=G, T_out=T_out, n_out=Cout, cv=Cg)
    assert rel_err(y.transpose(1, 2), ref) < 3e-5
    dy = torch.randn(B, Cout, T_out, generator=g)
    wr = w.double().clone().requires_grad_(True)
    conv_ref(x.double(), wr, k, s, pl, G, T_out).backward(dy.double())
    gw = torch.zeros(G, N, k * cgp, dtype=torch.float32, device="cuda")
    ops.X3_WGRAD = True
    try:
        ops.conv1d_cl_wgrad(xcl, to_cl(dy, Cout, torch.float32), gw, k=k, stride=s, pad=pl, groups=G, T_out=T_out, cv=Cg)
    finally:
        ops.X3_WGRAD = False
    gw_un = torch.zeros(Cout, Cg, k, dtype=torch.float32, device="cuda")
    ops.unpack_wgrad(gw, gw_un, GN=Cout, Cg=Cg, k=k, Cgp=cgp)
    assert rel_err(gw_un, wr.grad) < 3e-5


@pytest.mark.parametrize("masked", [False, True])
def test_conv_bf16x3_one_pass_slab_kernel(ops, masked):
    """A2V_BF16X3S (conv_nt_slab3_kernel): the pos-conv's three split products in one pass over stacked [hi ; lo] images -- bias,
    fp32 residual, ragged last row block (T = 600 = 384 + 216), and the student's first layer: clones of a clip (bdiv) under a row
    mask.  Against the fp64 convolution of the same fp32 operands (3e-5) and against the three-launch form (same products, another
    summation order: 2e-6)."""
    from animal2vec_amd.params import _padk
    g = torch.Generator().manual_seed(11)
    Cin = Cout = 1024
    k, pl, G, T, Bsrc, cb = 19, 9, 16, 600, 2, 3 if masked else 1
    Cg = Cin // G
    B = Bsrc * cb
    x = torch.randn(Bsrc, Cin, T, generator=g)
    w = torch.randn(Cout, Cg, k, generator=g) * (Cg * k) ** -0.5
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, T, Cout, generator=g)
    mask = (torch.rand(B, T, generator=g) < 0.6) if masked else torch.zeros(B, T, dtype=torch.bool)
    xin = x.repeat_interleave(cb, 0) * (~mask).unsqueeze(1).to(x.dtype)
    ref = conv_ref(xin.double(), w.double(), k, 1, pl, G, T) + bias.double().view(1, -1, 1) + res.double().transpose(1, 2)
    xcl = to_cl(x, Cin, torch.float32)
    wp = dev(pack_ref(w, G, _padk(Cg, torch.bfloat16)), torch.float32)
    assert ops.x3_register_conv_pack(wp, k, Cg) is not None
    kw = dict(k=k, stride=1, pad=pl, groups=G, T_out=T, n_out=Cout, cv=Cg, res=res.cuda())
    if masked:
        kw.update(rowmask=mask.to(torch.uint8).cuda(), bdiv=cb, n_batch=B)
    assert ops.X3_CONV_FUSED
    y1 = ops.conv1d_cl(xcl, wp, bias.cuda(), **kw)
    ops.X3_CONV_FUSED = False
    try:
        y3 = ops.conv1d_cl(xcl, wp, bias.cuda(), **kw)
    finally:
        ops.X3_CONV_FUSED = True
    assert rel_err(y1.transpose(1, 2), ref) < 3e-5
    assert rel_err(y1, y3) < 2e-6


def test_linear_tail_round_goes_to_the_small_kernel(ops):
    """M = 768 x 151 rows, N = 1024: 7.08 rounds of 256-row tiles on 256 CUs.  The launcher gives the 1280 rows of the last,
    nearly empty round to the 128 x 128 kernel (a2v_debug_set(22)): outputs must be what the unsplit launch writes -- bit for bit
    on the persistent kernel's rows, to bf16 rounding on the tail rows (another summation order) -- including the dropout mask,
    which is a function of the element's index in the WHOLE output, and the column sums."""
    from animal2vec_amd import _lib
    dtype = torch.bfloat16
    M, N, K = 768 * 151, 1024, 1024
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).to(dtype).cuda()
    rows_main = 7 * 256 * 256 // 4

    def run():
        y1 = ops.linear(x, w, b, res=r, drop=(0.1, 1234))
        cs = torch.zeros(N, device="cuda")
        y2 = ops.linear(x, w, None, gelu_bwd_aux=r, colsum=cs, dact=True)
        return y1, y2, cs

    a1, a2, acs = run()
    _lib.load().a2v_debug_set(22, 0)
    try:
        b1, b2, bcs = run()
    finally:
        _lib.load().a2v_debug_set(22, 1)
    assert torch.equal(a1[:rows_main], b1[:rows_main]) and torch.equal(a2[:rows_main], b2[:rows_main])
    assert rel_err(a1[rows_main:].float(), b1[rows_main:].float()) < 4e-3
    assert rel_err(a2[rows_main:].float(), b2[rows_main:].float()) < 4e-3
    assert rel_err(acs, bcs) < 1e-4
    # same dropout decisions in the tail: dropped elements carry exactly the residual
    keep = ops.dropout_mask(M * N, 0.1, 1234).view(M, N)[rows_main:]
    dropped = keep == 0
    assert torch.equal(a1[rows_main:][dropped], r[rows_main:][dropped])
    ref = (x[rows_main:].float() @ w.float().t() + b) * keep.float() / 0.9 + r[rows_main:].float()
    assert rel_err(a1[rows_main:].float(), ref) < 1e-2


@pytest.mark.parametrize("S,p_", [(151, 0.1), (97, 0.0), (160, 0.1), (48, 0.0), (129, 0.1)])
def test_attention_bwd_bf16x3_matches_fp64(ops, S, p_):
    """The student's attention backward of the bf16x3 mode (attn_bwd_x3_kernel: whole sequence, S <= 160, HD = 64, every product as
    three bf16 products over split operands) against fp64 autograd of the same fp32 operands: dqkv 3e-5, the bias gradient (column
    sums, fused) 3e-5, d alibi_scale by its backward-error bound; S = 151 / 160 / 129 on four waves (64-key blocks), 97 / 48 on three;
    sorted random positions, attention dropout; and against the fp32-MFMA kernel of the same call."""
    N, H, hd, seed = 2, 3, 64, 31
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(N, S, 3 * H * hd, generator=g).requires_grad_(True)
    pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = (torch.rand(H, generator=g) + 0.5).requires_grad_(True)
    q, k, v = qkv.double().reshape(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, sc.double().view(1, 1, H, 1, 1))
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1) + bias).softmax(-1)
    drop = None
    if p_ > 0:
        keep = ops.dropout_mask2d(N * H * S, S, p_, seed).view(N, H, S, S).double().cpu()
        att = att * keep / (1 - p_)
        drop = (p_, seed)
    ref = (att @ v).transpose(1, 2).reshape(N, S, H * hd)
    do = torch.randn(N, S, H * hd, generator=g)
    ref.backward(do.double())
    qd, dod = qkv.detach().cuda(), do.cuda()
    args = (pos.cuda(), slopes.cuda(), sc.detach().cuda(), H)
    ops.X3_WGRAD = True
    try:
        out, lse = ops.attention_fwd(qd, *args, drop=drop)
        dal = torch.zeros(H, device="cuda")
        bg = torch.zeros(3 * H * hd, device="cuda")
        dqkv = ops.attention_bwd(qd, out, dod, lse, *args, dal, drop=drop, bias_grad=bg)
    finally:
        ops.X3_WGRAD = False
    dal32 = torch.zeros(H, device="cuda")
    dq32 = ops.attention_bwd(qd, out, dod, lse, *args, dal32, drop=drop)
    assert rel_err(out, ref) < 3e-5
    assert rel_err(dqkv, qkv.grad) < 3e-5, rel_err(dqkv, qkv.grad)
    assert rel_err(dqkv, dq32) < 3e-5
    assert rel_err(bg, qkv.grad.sum((0, 1))) < 3e-5
    assert rel_err(dal, sc.grad) < 2e-3 and rel_err(dal32, sc.grad) < 2e-3, (rel_err(dal, sc.grad), rel_err(dal32, sc.grad))


@pytest.mark.parametrize("S,kind", [(151, "kept"), (700, "arange"), (2000, "arange")])
def test_attention_fwd_bf16x3_matches_fp64(ops, S, kind):
    """A2V_BF16X3 attention forward: fp32 in / out, both products as three exact bf16 products (K, V split when staged, Q when
    loaded, P in registers), fp32 softmax.  Output, LSE and the mean attended distance against an fp64 reference: 3e-5."""
    N, H, hd = 2, 4, 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(N, S, 3 * H * hd, generator=g)
    if kind == "kept":
        pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int()
    else:
        pos = torch.arange(S).repeat(N, 1).int()
    slopes = torch.tensor(O.alibi_slopes(H))
    sc = torch.rand(H, generator=g) + 0.3
    ref = attn_ref(qkv.double(), pos, slopes.double(), sc.double(), H)
    q, k, _ = qkv.double().view(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, sc.view(1, 1, H, 1, 1)).double()
    sc_ = (q * hd ** -0.5) @ k.transpose(-2, -1) + bias
    ops.X3_WGRAD = True
    try:
        out, lse = ops.attention_fwd(qkv.cuda(), None if kind == "arange" else pos.cuda(), slopes.cuda(), sc.cuda(), H)
    finally:
        ops.X3_WGRAD = False
    out32, lse32 = ops.attention_fwd(qkv.cuda(), None if kind == "arange" else pos.cuda(), slopes.cuda(), sc.cuda(), H)
    assert rel_err(out, ref) < 3e-5, rel_err(out, ref)
    assert rel_err(out32, ref) < 1e-5
    assert float((lse[0].cpu().double() - torch.logsumexp(sc_, -1)).abs().max()) < 1e-4
    dist = (pos.double().unsqueeze(2) - pos.double().unsqueeze(1)).abs().unsqueeze(1)
    assert rel_err(lse[1], (sc_.softmax(-1) * dist).sum(-1)) < 1e-4
