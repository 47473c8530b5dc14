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

for name, M, N, K in [("qkv", 13440, 3072, 1024), ("fc1", 13440, 4096, 1024), ("fc2", 13440, 1024, 4096),
                      ("proj", 13440, 1024, 1024), ("teacher fc2", 16000, 1024, 4096), ("big", 8192, 8192, 8192)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: torch.nn.functional.linear(x, w))
    print(f"NT {name:12s} M={M} N={N} K={K}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: dy.t() @ x)
    print(f"TN {name:12s} (wgrad)            {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
