#!/usr/bin/env python3
"""Runs one NT GEMM shape back to back for a few seconds (with / without bias epilogue) while sampling rocm-smi clocks and power."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip().replace("\n", " | ")[:400])
        except Exception as e:  # noqa: BLE001
            out.append(repr(e))
        time.sleep(0.3)


def main():
    dev = torch.device("cuda", 0)
    for (M, N, K) in [(116736, 3072, 1024), (116736, 1024, 4096), (116736, 1024, 1024)]:
        x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.time()
        n = 0
        while time.time() - t0 < 3.0:
            for _ in range(50):
                ops.linear(x, w)
            torch.cuda.synchronize()
            n += 50
        dt = time.time() - t0
        stop.set()
        th.join()
        print(f"M {M} N {N} K {K}: {1e6 * dt / n:.1f} us per launch, {2.0 * M * N * K * n / dt / 1e12:.1f} TFLOP/s")
        for o in out[2:6]:
            print("   ", o)


if __name__ == "__main__":
    main()
