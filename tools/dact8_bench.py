#!/usr/bin/env python3
"""fc1 (bias + GELU + kept derivative) and fc2 input gradient (derivative multiplied in + column sums) of one student block at the
64-clip micro-batch, with the derivative tensor in bf16 and in 8 bits (include/a2v.h: A2V_EPI_DACT8); HIP events, median of 5 groups.
   python tools/dact8_bench.py [--rows 116736]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402
from tools.ln_bench import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=768 * 152)
    a = ap.parse_args()
    M, D, H = a.rows, 1024, 4096
    dt = torch.bfloat16
    x = torch.randn(M, D, device="cuda").to(dt)
    w1 = (torch.randn(H, D, device="cuda") * D ** -0.5).to(dt)
    b1 = torch.randn(H, device="cuda")
    dy = torch.randn(M, D, device="cuda").to(dt)
    w2t = (torch.randn(H, D, device="cuda") * D ** -0.5).to(dt)
    h = torch.empty(M, H, device="cuda", dtype=dt)
    dh = torch.empty(M, H, device="cuda", dtype=dt)
    cs = torch.zeros(H, device="cuda")
    for name, kd in (("bf16", dt), ("u8", torch.uint8), ("bf16", dt), ("u8", torch.uint8)):
        keep = torch.empty(M, H, device="cuda", dtype=kd)
        us1 = timed(lambda: ops.linear(x, w1, b1, gelu=True, pre_out=keep, dact=True, out=h))
        us2 = timed(lambda: ops.linear(dy, w2t, None, gelu_bwd_aux=keep, colsum=cs, dact=True, out=dh))
        fl = 2.0 * M * D * H
        print(f"derivative {name:5s} fc1 {us1:8.1f} us ({fl / us1 / 1e6:6.0f} TFLOP/s)   fc2 input gradient {us2:8.1f} us ({fl / us2 / 1e6:6.0f} TFLOP/s)", flush=True)
        del keep


if __name__ == "__main__":
    main()
