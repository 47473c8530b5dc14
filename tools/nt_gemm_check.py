#!/usr/bin/env python3
"""Quick check of the persistent NT GEMM's plain / bias flavours (deferred epilogue) against torch, several shapes, with timing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for (M, N, K, bias) in [(4096, 256, 256, False), (4096, 1024, 1024, False), (13920, 1024, 1024, True), (116736, 3072, 1024, True),
                            (116736, 1024, 1024, False), (116736, 1024, 4096, False), (50000, 512, 1280, True), (2000 * 64, 1024, 512, True)]:
        x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev) if bias else None
        y = ops.linear(x, w, bias=b)
        torch.cuda.synchronize()
        ref = x.float() @ w.float().t()
        if bias:
            ref = ref + b
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            ops.linear(x, w, bias=b)
        e0.record()
        for _ in range(10):
            ops.linear(x, w, bias=b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"M {M:7d} N {N:5d} K {K:5d} bias {int(bias)}: rel err {err:.2e}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
        assert err < 1e-2, err


if __name__ == "__main__":
    main()
