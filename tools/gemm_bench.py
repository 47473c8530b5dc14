#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernels on the shapes of the large model (one GPU).
   python tools/gemm_bench.py [--reps 20]
Each line: shape, epilogue, median microseconds, TFLOP/s.  Random (gaussian) operands, bf16."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def timeit(fn, reps, inner=10):
    """median over `reps` of the per-call time of `inner` back-to-back launches between one event pair (a single
    launch per pair measures the CPU launch path for kernels shorter than ~50 us)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--p8", type=int, default=None, help="0: disable the 256x256 8-phase kernel")
    ap.add_argument("--lin", action="append", default=[], help="extra plain linear shape M,N,K (repeatable)")
    ap.add_argument("--dyn", type=int, default=None, help="persistent NT kernel tile hand-out mode 0/1/2")
    ap.add_argument("--debug-set", action="append", default=[], help="key=value for a2v_debug_set")
    ap.add_argument("--mscale", type=int, default=1, help="multiply the rows of the default Linear / wgrad shapes (4 = micro-batch 32)")
    ap.add_argument("--attn", action="append", default=[], help="extra attention shape N,S (repeatable)")
    a = ap.parse_args()
    for kv in a.debug_set:
        from animal2vec_amd import _lib
        k_, v_ = kv.split("=")
        _lib.load().a2v_debug_set(int(k_), int(v_))
        print("debug_set", kv)
    if a.dyn is not None:
        from animal2vec_amd import _lib
        _lib.load().a2v_debug_set(7, a.dyn)
        print("dyn", a.dyn)
    if a.variant is not None:
        from animal2vec_amd import _lib
        _lib.load().a2v_debug_set(0, a.variant)
        print("variant", a.variant)
    if a.p8 is not None:
        from animal2vec_amd import _lib
        _lib.load().a2v_debug_set(1, a.p8)
        print("p8", a.p8)
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g, dtype=torch.float32).to(dt)
    rows = []

    def lin(name, M, N, K, **kw):
        if a.only and a.only not in name:
            return
        if not name.startswith("lin ") and not name.startswith("big"):
            M *= a.mscale
        x, w = rn(M, K), rn(N, K) * K ** -0.5
        bias = torch.randn(N, device=dev)
        extra = {}
        if kw.get("res"):
            extra["res"] = rn(M, N)
        if kw.get("gelu"):
            extra["gelu"] = True
        if kw.get("pre"):
            extra["pre_out"] = torch.empty(M, N, device=dev, dtype=dt)
        if kw.get("gelu_bwd"):
            extra["gelu_bwd_aux"] = rn(M, N)
        if kw.get("colsum"):
            extra["colsum"] = torch.zeros(N, device=dev)
        if kw.get("drop"):
            extra["drop"] = (0.1, 123)
        if kw.get("dact"):
            extra["dact"] = True
        out = torch.empty(M, N, device=dev, dtype=dt)
        ms = timeit(lambda: ops.linear(x, w, bias if not kw.get("gelu_bwd") else None, out=out, **extra), a.reps)
        rows.append((name, M, N, K, 1, ms, 2.0 * M * N * K / ms / 1e9))

    def wgrad(name, M, N, K):
        if a.only and a.only not in name:
            return
        M *= a.mscale
        x, dy = rn(M, K), rn(M, N)
        gw = torch.zeros(N, K, device=dev)
        ms = timeit(lambda: ops.linear_wgrad(x, dy, gw), a.reps)
        rows.append((name, M, N, K, 1, ms, 2.0 * M * N * K / ms / 1e9))

    def conv(name, B, T, D, G, k, wg=False):
        if a.only and a.only not in name:
            return
        Cg = D // G
        x = rn(B, T, D)
        if wg:
            dy = rn(B, T, D)
            gw = torch.zeros(G, Cg, k * Cg, device=dev)
            ms = timeit(lambda: ops.conv1d_cl_wgrad(x, dy, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T), a.reps)
        else:
            w = rn(G, Cg, k * Cg) * (k * Cg) ** -0.5
            bias = torch.randn(D, device=dev)
            out = torch.empty(B, T, D, device=dev, dtype=dt)
            ms = timeit(lambda: ops.conv1d_cl(x, w, bias, k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D, out=out), a.reps)
        rows.append((name, B * T, Cg, k * Cg, G, ms, 2.0 * B * T * Cg * k * Cg * G / ms / 1e9))

    def dec(name, B, T, CI, CO, res=False):
        """decoder grouped conv (k 7, 16 groups, taps padded to 64 in the packed operand)"""
        if a.only and a.only not in name:
            return
        G, k = 16, 7
        x = rn(B, T, G * CI)
        w = rn(G, CO, k * 64) * (k * CI) ** -0.5
        bias = torch.randn(G * CO, device=dev)
        r = rn(B, T, G * CO) if res else None
        out = torch.empty(B, T, G * CO, device=dev, dtype=dt)
        ms = timeit(lambda: ops.conv1d_cl(x, w, None if res else bias, k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=G * CO, out=out,
                                          cv=CI, res=r), a.reps)
        rows.append((name, B * T, CO, k * CI, G, ms, 2.0 * B * T * CO * k * CI * G / ms / 1e9))

    for spec in a.lin:
        M_, N_, K_ = (int(v) for v in spec.split(","))
        lin(f"lin {spec}", M_, N_, K_)
    lin("qkv bias", 13440, 3072, 1024)
    lin("fc1 bias", 13440, 4096, 1024)
    lin("fc1 bias+gelu", 13440, 4096, 1024, gelu=True)
    lin("fc1 bias+gelu+pre", 13440, 4096, 1024, gelu=True, pre=True)
    lin("fc1 bias+gelu+dact", 13440, 4096, 1024, gelu=True, pre=True, dact=True)
    lin("fc2dgrad gelu_bwd+colsum", 13440, 4096, 1024, gelu_bwd=True, colsum=True)
    lin("fc2dgrad gelu_bwd only", 13440, 4096, 1024, gelu_bwd=True)
    lin("fc2dgrad dact+colsum", 13440, 4096, 1024, gelu_bwd=True, colsum=True, dact=True)
    lin("fc2 bias+res", 13440, 1024, 4096, res=True)
    lin("fc2 bias+res+drop", 13440, 1024, 4096, res=True, drop=True)
    lin("proj bias+res+drop", 13440, 1024, 1024, res=True, drop=True)
    lin("teacher fc2", 16000, 1024, 4096)
    lin("big 8192^3", 8192, 8192, 8192)
    wgrad("wgrad fc1", 13440, 4096, 1024)
    wgrad("wgrad fc2", 13440, 1024, 4096)
    wgrad("wgrad proj", 13440, 1024, 1024)
    dec("conv dec 48->48 bias", 384, 2000, 48, 48)
    dec("conv dec 48->48 res", 384, 2000, 48, 48, res=True)
    dec("conv dec 64->48 bias", 384, 2000, 64, 48)
    dec("conv dec 48->64 res", 384, 2000, 48, 64, res=True)
    conv("posconv fwd", 96, 2000, 1024, 16, 19)
    conv("posconv wgrad", 96, 2000, 1024, 16, 19, wg=True)
    def attn(name, N, S, H=16, HD=64, drop=None):
        if a.only and a.only not in name:
            return
        qkv = rn(N, S, 3 * H * HD)
        pos = torch.arange(S, device=dev, dtype=torch.int32).repeat(N, 1).contiguous()
        slopes = torch.rand(H, device=dev)
        asc = torch.ones(H, device=dev)
        ms = timeit(lambda: ops.attention_fwd(qkv, pos, slopes, asc, H, drop=drop), a.reps)
        fl = 4.0 * N * H * S * S * HD
        rows.append((name + " fwd", N * S, HD, S, H, ms, fl / ms / 1e9))
        out, lse = ops.attention_fwd(qkv, pos, slopes, asc, H, drop=drop)
        do = rn(N, S, H * HD)
        dal = torch.zeros(H, device=dev)
        ms = timeit(lambda: ops.attention_bwd(qkv, out, do, lse, pos, slopes, asc, H, dal, drop=drop), a.reps)
        rows.append((name + " bwd", N * S, HD, S, H, ms, 2.5 * fl / ms / 1e9))

    for spec in a.attn:
        N_, S_ = (int(v) for v in spec.split(","))
        attn(f"attn {spec}", N_, S_)
    attn("attn student", 96, 140, drop=(0.1, 7))
    attn("attn student nodrop", 96, 140)
    attn("attn teacher", 8, 2000)

    def ln(name, M, C, res=False, act=0, affine=True):
        if a.only and a.only not in name:
            return
        x = rn(M, C)
        r_ = rn(M, C) if res else None
        g_ = torch.ones(C, device=dev) if affine else None
        b_ = torch.zeros(C, device=dev) if affine else None
        out = torch.empty_like(x)
        ms = timeit(lambda: ops.layernorm_fwd(x, g_, b_, 1e-5, res=r_, act=act, out=out), a.reps)
        byt = M * C * 2 * (3 if res else 2)
        print(f"{name:28s} M={M:7d} C={C:5d}  {1e3 * ms:9.1f} us  {byt / ms / 1e6:8.1f} GB/s")
        _, mean, rstd = ops.layernorm_fwd(x, g_, b_, 1e-5, res=r_, act=act)
        dy = rn(M, C)
        dg = torch.zeros(C, device=dev) if affine else None
        db = torch.zeros(C, device=dev) if affine else None
        dxo = torch.empty_like(x)
        ms = timeit(lambda: ops.layernorm_bwd(dy, x, g_, b_, mean, rstd, act=act, dgamma=dg, dbeta=db, out=dxo), a.reps)
        print(f"{name + ' bwd':28s} M={M:7d} C={C:5d}  {1e3 * ms:9.1f} us  {M * C * 2 * 3 / ms / 1e6:8.1f} GB/s")

    ln("ln block", 13440, 1024)
    ln("ln teacher res", 16000, 1024, res=True)
    ln("ln posconv gelu", 192000, 1024, act=1, affine=False)
    ln("ln conv512 gelu", 128000, 512, act=1)
    for r in rows:
        print(f"{r[0]:28s} M={r[1]:7d} N={r[2]:5d} K={r[3]:5d} G={r[4]:3d}  {1e3 * r[5]:9.1f} us  {r[6]:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
