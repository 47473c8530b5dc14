#!/usr/bin/env python3
"""Micro-benchmark of the decoder-input and loss kernels at the large model's micro-batch (one GPU).
   python tools/elem_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402


def main():
    for B in (8, 32):
        one(B)


def one(B):
    dev, dt = "cuda", torch.bfloat16
    cb, T, K, D = 12, 2000, 140, 1024
    print(f"--- {B} clips x {cb} clones")
    N = B * cb
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(N, T, D, device=dev, generator=g).to(dt)
    y = torch.randn(B, T, D, device=dev, generator=g)
    perm = torch.rand(N, T, device=dev, generator=g).argsort(dim=1)
    restore = perm.argsort(dim=1).to(torch.int32).contiguous()
    mask = (restore >= K).to(torch.uint8).contiguous()
    stats = torch.zeros(4 * D + 2, device=dev, dtype=torch.float64)
    ms = timeit(lambda: ops.d2v_loss(x, y, mask, cb, stats=stats), 10, inner=4)
    byt = 2 * N * T * D * 2 + B * T * D * 4
    print(f"d2v_loss            {1e3 * ms:9.1f} us  {byt / ms / 1e6:8.1f} GB/s")
    xk = torch.randn(N, K, D, device=dev, generator=g).to(dt)
    noise = torch.randn(N, T - K, D, device=dev, generator=g) * 0.01
    ms = timeit(lambda: ops.decoder_input(xk, noise, restore), 10, inner=4)
    print(f"decoder_input       {1e3 * ms:9.1f} us  {(N * T * D * 2 + N * (T - K) * D * 4) / ms / 1e6:8.1f} GB/s")
    ms = timeit(lambda: ops.decoder_input(xk, (0.01, 5), restore), 10, inner=4)
    print(f"decoder_input rng   {1e3 * ms:9.1f} us  {N * T * D * 2 / ms / 1e6:8.1f} GB/s")
    ms = timeit(lambda: torch.empty(N, T - K, D, device=dev).normal_(0, 0.01), 10, inner=4)
    print(f"torch normal_       {1e3 * ms:9.1f} us")
    z = torch.empty_like(x)
    ms = timeit(lambda: z.copy_(x), 10, inner=4)
    print(f"copy (N,T,D) bf16   {1e3 * ms:9.1f} us  {2 * N * T * D * 2 / ms / 1e6:8.1f} GB/s")
    ms = timeit(lambda: z.zero_(), 10, inner=4)
    print(f"fill (N,T,D) bf16   {1e3 * ms:9.1f} us  {N * T * D * 2 / ms / 1e6:8.1f} GB/s")
    del z
    del x, noise
    # instance-normed target average over the top 16 layers
    L = 16
    layers = [torch.randn(B, T, D, device=dev, generator=g).to(dt) for _ in range(L)]
    ms = timeit(lambda: ops.instnorm_targets(layers, 1e-5), 10, inner=2)
    print(f"instnorm targets    {1e3 * ms:9.1f} us  {(2 * L * B * T * D * 2 + B * T * D * 4) / ms / 1e6:8.1f} GB/s (two passes)")
    del layers
    # col2im of the strided feature convs: (C, k, s, T_in) at 80000 samples
    for C_, k_, s_, T_in in ((512, 3, 2, 16000), (512, 3, 2, 8000), (512, 3, 2, 4000), (128, 10, 5, 80000)):
        p_ = k_ // 2
        T_out = (T_in + 2 * p_ - k_) // s_ + 1
        dA = torch.randn(B * T_out, k_ * C_, device=dev, generator=g).to(dt)
        ms = timeit(lambda: ops.col2im(dA, B=B, T_in=T_in, T_out=T_out, C_=C_, k=k_, s=s_, p=p_), 10, inner=2)
        print(f"col2im C={C_} k={k_} s={s_} T={T_in:6d} {1e3 * ms:9.1f} us  {(dA.numel() * 2 + B * T_in * C_ * 2) / ms / 1e6:8.1f} GB/s")
        del dA
    # SincConv front end: 127 filters x 63 taps on 10 s @ 8 kHz
    C_, k_, Ls = 127, 63, 80000
    xs = torch.randn(B, Ls, device=dev, generator=g)
    filt = torch.randn(128, k_, device=dev, generator=g) * 0.1
    filt = (filt + filt.flip(1)) * 0.5                       # band-pass rows are symmetric
    filt[C_:] = 0
    lw, lb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    pa, pb = torch.ones(128, device=dev), torch.ones(128, device=dev)
    ms = timeit(lambda: ops.sinc_fused_fwd(xs, filt, lw, lb, pa, pb, C_=C_, k=k_, dtype=dt), 10, inner=2)
    print(f"sinc_fused_fwd      {1e3 * ms:9.1f} us  {B * Ls * 128 * 2 / ms / 1e6:8.1f} GB/s")
    ys, mean, rstd = ops.sinc_fused_fwd(xs, filt, lw, lb, pa, pb, C_=C_, k=k_, dtype=dt)
    dys = torch.randn(B, Ls, 128, device=dev, generator=g).to(dt)
    dfilt, dpar = torch.zeros(128, k_, device=dev), torch.zeros(4, 128, device=dev)
    ms = timeit(lambda: ops.sinc_fused_bwd(xs, filt, lw, lb, pa, pb, dys, mean, rstd, dfilt, dpar, C_=C_, k=k_), 10, inner=2)
    print(f"sinc_fused_bwd      {1e3 * ms:9.1f} us  {B * Ls * 128 * 2 / ms / 1e6:8.1f} GB/s")
    del ys, dys
    mx = ops.Mixup(8000, 0.05, dev)
    ms = timeit(lambda: mx.gain_db(xs), 10, inner=2)
    print(f"mixup gain_db       {1e3 * ms:9.1f} us")
    out = ops.decoder_input(xk, (0.01, 5), restore).float()
    nz = out[mask.bool()]
    print("noise mean %.3e std %.5f (want 0.01) kurt %.3f" % (nz.mean().item(), nz.std().item(), ((nz / nz.std()) ** 4).mean().item()))


if __name__ == "__main__":
    main()
