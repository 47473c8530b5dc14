#!/usr/bin/env python3
"""Where a K = 1024 tile of the persistent NT GEMM spends the time outside its K loop (round 4).

Three experiment builds next to the product library (python -m animal2vec_amd.build --variant <tag> <defines>), selected with
A2V_LIB_PATH; this script runs the step's shapes on whichever library is loaded and prints what that build can tell:

  product / -DP8_L2STORE / -DP8_SKEW : launch time by HIP events (median of back-to-back launches); with P8_SKEW also at
                                       a2v_debug_set(24, s) for the start skews given by --skews
  -DP8_STAMP -DP8_STAMP2            : per-tile stamps of wave 0 of every workgroup (kept in LDS, copied out at the end):
                                       epilogue duration, time from the last store to the end of the next tile's first / second
                                       K-tile, the K loop's time per K-tile, and how the epilogue starts of the 256 workgroups
                                       are spread over a tile period (s_memrealtime, 100 MHz, one clock for the whole chip)
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import _lib, ops  # noqa: E402

ST2_MAX = 96


def shapes(M):
    return [("qkv   bias", M, 3072, 1024, "b"), ("fc1   bias+gelu+gelu'", M, 4096, 1024, "fc1"), ("proj  plain", M, 1024, 1024, ""),
            ("fc2dg gelu'+colsum", M, 4096, 1024, "fc2d"), ("fc2   bias+res", M, 1024, 4096, "br")]


def make(dev, M_, N, K, fl):
    x = (torch.randn(M_, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    out = torch.empty(M_, N, device=dev, dtype=torch.bfloat16)
    res = (torch.randn(M_, N, device=dev) * 0.5).bfloat16() if fl in ("br", "r", "fc2d") else None
    pre = torch.empty(M_, N, device=dev, dtype=torch.bfloat16) if fl == "fc1" else None
    cs = torch.zeros(N, device=dev)
    return {"b": lambda: ops.linear(x, w, bias=b, out=out), "": lambda: ops.linear(x, w, out=out),
            "fc1": lambda: ops.linear(x, w, bias=b, gelu=True, pre_out=pre, dact=True, out=out),
            "br": lambda: ops.linear(x, w, bias=b, res=res, out=out), "r": lambda: ops.linear(x, w, res=res, out=out),
            "fc2d": lambda: ops.linear(x, w, gelu_bwd_aux=res, dact=True, colsum=cs, out=out)}[fl]


def timed(fn, reps=30, rounds=5):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=116736)
    ap.add_argument("--skews", type=str, default="")
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    has_st2 = hasattr(lib, "a2v_diag_p8_stamps2")
    skews = [int(s) for s in args.skews.split(",") if s != ""]
    print(f"# library {os.environ.get('A2V_LIB_PATH', 'product')}  stamps {'yes' if has_st2 else 'no'}  skews {skews}", flush=True)
    for name, M_, N, K, fl in shapes(args.M):
        if args.only and args.only not in name:
            continue
        fn = make(dev, M_, N, K, fl)
        flops = 2.0 * M_ * N * K
        if skews:
            for s in skews:
                assert lib.a2v_debug_set(24, s) == 0, "this library has no start-skew switch (-DP8_SKEW)"
                m, lo = timed(fn)
                print(f"{name:22s} N {N} K {K} skew {s:3d}: median {m:7.1f} us  min {lo:7.1f} us  {flops / m / 1e6:7.1f} TFLOP/s", flush=True)
            lib.a2v_debug_set(24, 0)
            continue
        m, lo = timed(fn)
        print(f"{name:22s} N {N} K {K}: median {m:7.1f} us  min {lo:7.1f} us  {flops / m / 1e6:7.1f} TFLOP/s", flush=True)
        if not has_st2:
            continue
        buf = (C.c_ulonglong * (256 * ST2_MAX * 5))()
        nb = (C.c_int * 256)()
        assert lib.a2v_diag_p8_stamps2(buf, nb) == 0
        nk = K // 64
        epi, gap1, gap2, ktile, period, clk = [], [], [], [], [], []
        starts = {}
        for w in range(256):
            n = nb[w]
            rec = [[buf[(w * ST2_MAX + t) * 5 + j] for j in range(5)] for t in range(n)]
            for t in range(1, n - 1):                    # (the first tile has no predecessor, the last no successor)
                a_r, a_c, b_c, c_c, d_c = rec[t]
                nxt = rec[t + 1]
                if not (b_c > a_c and c_c > b_c and d_c > c_c and nxt[1] > d_c):
                    continue
                # shader clock over this tile period from the two time bases
                per_r = (nxt[0] - a_r) / 100.0                      # us
                per_c = nxt[1] - a_c
                if per_r <= 0:
                    continue
                f = per_c / per_r                                    # cycles per us = MHz
                clk.append(f)
                epi.append((b_c - a_c) / f)
                gap1.append((c_c - b_c) / f)
                gap2.append((d_c - c_c) / f)
                ktile.append((nxt[1] - d_c) / f / (nk - 2))
                period.append(per_r)
            for t in (2, 6, 12):
                if t < n:
                    starts.setdefault(t, []).append(rec[t][0] / 100.0)
        if not epi:
            print("   (no complete stamp records)")
            continue
        P = med(period)
        print(f"   per tile (median over {len(epi)} tiles): period {P:6.2f} us = epilogue {med(epi):5.2f} + last store -> end of next K-tile 0 {med(gap1):5.2f}"
              f" + K-tile 1 {med(gap2):5.2f} + {nk - 2} K-tiles x {med(ktile):5.3f}   (shader clock {med(clk):5.0f} MHz)")
        for t, v in sorted(starts.items()):
            v = sorted(v)
            v0 = v[0]
            ph = [((x - v0) % P) / P for x in v]
            hist = [0] * 10
            for p in ph:
                hist[min(9, int(p * 10))] += 1
            print(f"   epilogue starts of tile #{t} over {len(v)} workgroups: spread {v[-1] - v0:6.2f} us (p10-p90 {v[int(.9 * len(v))] - v[int(.1 * len(v))]:6.2f}); "
                  f"phase histogram over one period {hist}")


if __name__ == "__main__":
    main()
