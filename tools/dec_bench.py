#!/usr/bin/env python3
"""Decoder1d's grouped convolutions alone (a2v_conv_dec.hip) at the 64-clip micro-batch: forward with bias, input gradient with the
residual add, 48 -> 48 channels per group (the three 768 -> 768 layers); HIP events, median of 5 groups.  Against another build:
   A2V_LIB_PATH=animal2vec_amd/liba2v_hip_prev.so python tools/dec_bench.py   (tools/build_prev.sh <rev>)
   python tools/dec_bench.py [--clips 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402
from tools.ln_bench import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    a = ap.parse_args()
    N, T, G, k, C = a.clips * 12, 2000, 16, 7, 48
    D = G * C
    dt = torch.bfloat16
    x = torch.randn(N, T, D, device="cuda").to(dt)
    res = torch.randn(N, T, D, device="cuda").to(dt)
    w = (torch.randn(G, C, k * 64, device="cuda") * (C * k) ** -0.5).to(dt)
    b = torch.randn(D, device="cuda")
    y = torch.empty_like(x)
    for rep in range(2):
        us1 = timed(lambda: ops.conv1d_cl(x, w, b, k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D, cv=C, out=y))
        us2 = timed(lambda: ops.conv1d_cl(x, w, None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D, cv=C, res=res, out=y))
        by = N * T * D * 2
        gw = torch.zeros(G, C, k * 64, dtype=torch.float32, device="cuda")
        us3 = timed(lambda: ops.conv1d_cl_wgrad(x, res, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, cv=C))
        print(f"{os.environ.get('A2V_LIB_PATH', 'product library')[-24:]:24s}  bias {us1:8.1f} us ({2 * by / us1 / 1e6:5.2f} TB/s)   residual {us2:8.1f} us "
              f"({3 * by / us2 / 1e6:5.2f} TB/s)   weight gradient {us3:8.1f} us ({2 * by / us3 / 1e6:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
