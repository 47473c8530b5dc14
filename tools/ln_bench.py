#!/usr/bin/env python3
"""LayerNorm forward / backward alone at the row counts of the pretrain step (HIP events, median of groups):
   python tools/ln_bench.py            # micro-batch 32 sizes;  A2V_LN_ROWS=13440,16000 for micro-batch 8"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def timed(fn, iters=10, groups=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    rows = [int(r) for r in os.environ.get("A2V_LN_ROWS", "53760,64000").split(",")]
    for M in rows:
        for C, act in ((1024, 0), (512, 1)):
            Mx = M if C == 1024 else M * 8            # the conv feature extractor's longer tensors
            x = torch.randn(Mx, C, device="cuda").bfloat16()
            dy = torch.randn(Mx, C, device="cuda").bfloat16()
            w = torch.ones(C, device="cuda")
            b = torch.zeros(C, device="cuda")
            y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5, act=act)
            us = timed(lambda: ops.layernorm_fwd(x, w, b, 1e-5, act=act, out=y))
            print(f"fwd  M={Mx:7d} C={C:5d} act={act}  {us:8.1f} us  {Mx * C * 2 * 2 / us / 1e6:6.2f} TB/s", flush=True)
            dg, db, ds = (torch.zeros(C, device="cuda") for _ in range(3))
            dx = torch.empty_like(x)
            for mode, nm, nbuf in ((0, "bwd      ", 3), (2, "bwd drop2", 4)):
                if act and mode:
                    continue
                kw = dict(drop=(0.1, 7), drop_mode=2) if mode else {}
                us = timed(lambda: ops.layernorm_bwd(dy, x, w, b, mean, rstd, act=act, dgamma=dg, dbeta=db, dxsum=ds, out=dx, **kw))
                print(f"{nm} M={Mx:7d} C={C:5d} act={act}  {us:8.1f} us  {Mx * C * 2 * nbuf / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
