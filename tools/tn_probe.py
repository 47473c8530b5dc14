#!/usr/bin/env python3
"""Weight-gradient (TN) GEMM of the Linear layers on the step's shapes, one launch at a time (gemm_tn_bf16_p8_kernel).
   python tools/tn_probe.py [--M 116736] [--only fc1]
Product library, or a timing build selected with A2V_LIB_PATH (python -m animal2vec_amd.build --variant t8eN -DT8_EXP=N: 1 no MFMA,
2 no LDS-DMA in the K loop, 3 no fragment reads, 4 no atomics -- wrong results, durations only)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def timeit(fn, reps=12, inner=4):
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
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=116736)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    print("# library", os.environ.get("A2V_LIB_PATH", "product"), " M", a.M)
    dev = "cuda"
    M = a.M
    g = torch.Generator(device=dev).manual_seed(1)

    def mk(cols):
        return (torch.randn(M, cols, device=dev, generator=g) * 0.5).to(torch.bfloat16)

    shapes = [("fc1", 4096, 1024), ("fc2", 1024, 4096), ("qkv", 3072, 1024), ("proj", 1024, 1024)]
    for name, N, K in shapes:
        if a.only and a.only not in name:
            continue
        x, dy = mk(K), mk(N)
        gw = torch.zeros(N, K, device=dev, dtype=torch.float32)
        med, mn = timeit(lambda: ops.linear_wgrad(x, dy, gw))
        fl = 2.0 * M * N * K
        print(f"{name:5s} N {N:5d} K {K:5d}: median {med:8.1f} us  min {mn:8.1f} us  {fl / med / 1e6:8.1f} TFLOP/s", flush=True)
        del x, dy, gw
    if not a.only or "pair" in a.only:
        xa, dya, xb, dyb = mk(1024), mk(4096), mk(4096), mk(1024)
        ga = torch.zeros(4096, 1024, device=dev, dtype=torch.float32)
        gb = torch.zeros(1024, 4096, device=dev, dtype=torch.float32)
        med, mn = timeit(lambda: ops.linear_wgrad_pair(xa, dya, ga, xb, dyb, gb))
        fl = 4.0 * M * 4096 * 1024
        print(f"pair fc1+fc2           : median {med:8.1f} us  min {mn:8.1f} us  {fl / med / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
