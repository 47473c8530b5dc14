#!/usr/bin/env python3
"""What the vendor library (hipBLASLt through torch.matmul) reaches on the model's GEMM shapes: a practical
ceiling to compare tools/gemm_bench.py against.  Not part of the product path."""
import torch

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

import sys
# default: the shapes of today's step (64-clip micro-batch: 768 sequences x 152 kept frames = 116 736 student rows, 64 x 2000 = 128 000
# teacher rows); --small: round 2's 8-clip shapes
SHAPES = [("qkv", 116736, 3072, 1024), ("fc1", 116736, 4096, 1024), ("fc2", 116736, 1024, 4096), ("proj", 116736, 1024, 1024),
          ("qkv dgrad", 116736, 1024, 3072), ("teacher qkv", 128000, 3072, 1024), ("teacher fc1", 128000, 4096, 1024),
          ("teacher fc2", 128000, 1024, 4096), ("big", 8192, 8192, 8192)]
if "--small" in sys.argv:
    SHAPES = [("qkv", 13440, 3072, 1024), ("fc1", 13440, 4096, 1024), ("fc2", 13440, 1024, 4096), ("proj", 13440, 1024, 1024),
              ("teacher fc2", 16000, 1024, 4096), ("big", 8192, 8192, 8192)]
for name, M, N, K in SHAPES:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    ms = timeit(lambda: torch.nn.functional.linear(x, w))
    print(f"NT {name:12s} M={M} N={N} K={K}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: dy.t() @ x)
    print(f"TN {name:12s} (wgrad)            {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
