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
    dev, dt = "cuda", torch.bfloat16
    B, cb, T, K, D = 8, 12, 2000, 140, 1024
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
    out = ops.decoder_input(xk, (0.01, 5), restore).float()
    nz = out[mask.bool()]
    print("noise mean %.3e std %.5f (want 0.01) kurt %.3f" % (nz.mean().item(), nz.std().item(), ((nz / nz.std()) ** 4).mean().item()))


if __name__ == "__main__":
    main()
