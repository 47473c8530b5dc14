#!/usr/bin/env python3
"""In-kernel shader clock of the persistent NT GEMM (MI355X_MICROARCH.md, DVFS give-back item 6).

Needs the DIAGNOSTIC build (python -m animal2vec_amd.build --variant stamp -DP8_STAMP; run with
A2V_LIB_PATH=animal2vec_amd/liba2v_hip_stamp.so): wave 0 of every workgroup stamps s_memtime / s_memrealtime around its tile
loop into a private buffer.  Per shape: >= 2 s of back-to-back launches on random data, then the stamps of the LAST launch:
clock = d(memtime) / d(memrealtime) x 100 MHz, median over workgroups.  Also prints the launch time by HIP events of the
same run and the MFMA-busy fraction the flop count implies at that clock (16x16x32 bf16: 16 cycles per MFMA per SIMD).
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import _lib, ops  # noqa: E402


def main():
    lib = _lib.load()
    if not hasattr(lib, "a2v_diag_p8_stamps"):
        raise SystemExit("this library has no stamps: build the diagnostic variant and point A2V_LIB_PATH at it")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    M = int(os.environ.get("CLK_M", 116736))
    secs = float(os.environ.get("CLK_SECS", 2.5))
    shapes = [("qkv   bias      ", M, 3072, 1024, "b"), ("fc1   bias+gelu+gelu'", M, 4096, 1024, "fc1"), ("proj  plain     ", M, 1024, 1024, ""),
              ("fc2   bias+res  ", M, 1024, 4096, "br"), ("fc2dg gelu'+colsum", M, 4096, 1024, "fc2d"), ("fc1dg res ", M, 1024, 4096, "r")]
    buf = (C.c_ulonglong * (256 * 4))()
    for name, M_, N, K, fl in shapes:
        x = (torch.randn(M_, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        out = torch.empty(M_, N, device=dev, dtype=torch.bfloat16)
        res = (torch.randn(M_, N, device=dev) * 0.5).bfloat16() if fl in ("br", "r", "fc2d") else None
        pre = torch.empty(M_, N, device=dev, dtype=torch.bfloat16) if fl == "fc1" else None
        cs = torch.zeros(N, device=dev)
        fn = {"b": lambda: ops.linear(x, w, bias=b, out=out), "": lambda: ops.linear(x, w, out=out),
              "fc1": lambda: ops.linear(x, w, bias=b, gelu=True, pre_out=pre, dact=True, out=out),
              "br": lambda: ops.linear(x, w, bias=b, res=res, out=out), "r": lambda: ops.linear(x, w, res=res, out=out),
              "fc2d": lambda: ops.linear(x, w, gelu_bwd_aux=res, dact=True, colsum=cs, out=out)}[fl]
        t0 = time.time()
        n = 0
        while time.time() - t0 < secs:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        assert lib.a2v_diag_p8_stamps(buf) == 0
        clk, span = [], []
        for i in range(256):
            c0, c1, r0, r1 = buf[4 * i:4 * i + 4]
            if r1 > r0 and c1 > c0:
                clk.append((c1 - c0) / (r1 - r0) * 100.0)       # MHz
                span.append((r1 - r0) / 100.0)                   # us
        clk.sort()
        span.sort()
        med = clk[len(clk) // 2]
        tf = 2.0 * M_ * N * K / us / 1e6
        # MFMA cycles per SIMD: flops / (1024 SIMDs x 16*16*32*2 flop per MFMA) x 16 cycles
        busy = (2.0 * M_ * N * K / (1024 * 16384.0) * 16.0) / (med * 1e6 * us * 1e-6)
        print(f"{name:22s} M {M_} N {N} K {K}: {us:7.1f} us  {tf:7.1f} TFLOP/s | in-kernel clock median {med:6.0f} MHz (min {clk[0]:.0f} max {clk[-1]:.0f}, "
              f"{len(clk)} workgroups, tile loop {span[len(span)//2]:.0f} us) | MFMA-busy implied {busy:.3f} | dense peak at that clock {2500 * med / 2400:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
