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
    # the GELU flavours of the MLP at the headline shape (timing only; parity: tests/test_ops_gpu.py)
    M, N, K = 116736, 4096, 1024
    x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dy = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    cs = torch.zeros(N, device=dev)
    for name, fn in [("bias", lambda: ops.linear(x, w, bias=b, out=out)),
                     ("bias+gelu (teacher fc1)", lambda: ops.linear(x, w, bias=b, gelu=True, out=out)),
                     ("bias+gelu+gelu' store (fc1)", lambda: ops.linear(x, w, bias=b, gelu=True, pre_out=pre, dact=True, out=out)),
                     ("x gelu' + colsum (fc2 dgrad)", lambda: ops.linear(dy, w, gelu_bwd_aux=pre, dact=True, colsum=cs, out=out))]:
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"M {M} N {N} K {K} {name:30s}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
