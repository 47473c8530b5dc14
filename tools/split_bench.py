#!/usr/bin/env python3
"""a2v_split_bf16 at the shapes of the bf16x3 large step: microseconds and TB/s per (rows, K, mode)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402

SHAPES = [(13920, 1024, 0), (13920, 4096, 0), (13920, 3072, 0), (23904, 1024, 2), (23904, 384, 2), (23904, 1024, 0), (13920, 1024, 2),
          (13920, 4096, 2), (128000, 512, 2), (128000, 512, 0), (32000, 512, 0), (1024, 4096, 1), (4096, 1024, 1)]


def main():
    dev = torch.device("cuda", 0)
    for M, K, mode in SHAPES:
        x = torch.randn(M, K, device=dev)
        out = ops.split_bf16(x, mode)
        for _ in range(3):
            ops.split_bf16(x, mode, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.split_bf16(x, mode, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        nbytes = M * K * (4 + (4 if mode == 2 else 6))
        print(f"M {M:7d} K {K:5d} mode {mode}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
