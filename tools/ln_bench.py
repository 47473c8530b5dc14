#!/usr/bin/env python3
"""LayerNorm forward / backward alone, every hot flavour of the pretrain step at its row count (HIP events, median of groups),
the row-per-wave kernels (a2v_debug_set(30, 0)) beside the streaming ones at forced rows per batch / workgroups per CU:
   python tools/ln_bench.py [--clips 64] [--sweep]        # --sweep: rows 1/2/4/8 x workgroups per CU
Bytes = what the flavour must move (x in, y out, + post / dy / dropout(dx) where the flavour has them)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def timed(fn, iters=6, groups=5):
    for _ in range(2):
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


def flavours(clips, kept):
    cb, T = 12, 2000
    Ns, Nt = clips * cb * kept, clips * T
    big = clips * cb * T
    return [
        # name, M, C, act, affine, post, bwd (mode, xs) or None
        ("student block LN", Ns, 1024, 0, True, False, (2, True)),
        ("teacher block LN", Nt, 1024, 0, True, False, None),
        ("pos-conv student", big, 1024, 1, False, False, (0, True)),
        ("pos-conv teacher", Nt, 1024, 1, False, False, None),
        ("decoder", big, 768, 1, False, True, (0, True)),
        ("feature conv 1", clips * 16000, 512, 1, True, False, (0, False)),
        ("feature conv 4-7", clips * 2000, 512, 1, True, False, (0, False)),
        ("project_features", clips * 2000, 512, 0, True, False, (0, False)),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--kept", type=int, default=152)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--front", action="store_true")
    a = ap.parse_args()
    settings = [("row-per-wave", dict(k30=0)), ("stream auto", dict())]
    if a.sweep:
        settings += [(f"stream rows {r}", dict(k31=r)) for r in (1, 2, 4, 8)]
        settings += [(f"stream wpc {w}", dict(k32=w)) for w in (2, 4, 6)]
    if a.front:      # non-persistent: a workgroup owns 4 * bpw consecutive batches of r rows
        settings += [(f"front r{r} bpw{w}", dict(k31=r, k33=w)) for r in (1, 2, 4) for w in (1, 2, 4, 8, 16)]
    for name, M, C, act, affine, post, bwd in flavours(a.clips, a.kept):
        if a.only and a.only not in name:
            continue
        x = torch.randn(M, C, device="cuda").bfloat16()
        w = torch.ones(C, device="cuda") if affine else None
        b = torch.zeros(C, device="cuda") if affine else None
        pz = torch.randn(M, C, device="cuda").bfloat16() if post else None
        y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5, act=act, post=pz)
        fb = M * C * 2 * (3 if post else 2)
        for sname, st in settings:
            ops.debug_set(30, st.get("k30", 3)); ops.debug_set(31, st.get("k31", 0)); ops.debug_set(32, st.get("k32", 0)); ops.debug_set(33, st.get("k33", 0))
            us = timed(lambda: ops.layernorm_fwd(x, w, b, 1e-5, act=act, post=pz, out=y))
            print(f"fwd {name:18s} M={M:8d} C={C:5d} {sname:16s} {us:8.1f} us {fb / us / 1e6:6.2f} TB/s", flush=True)
        if bwd is None:
            continue
        mode, xs = bwd
        dy = torch.randn(M, C, device="cuda").bfloat16()
        dx = torch.empty_like(x)
        dg = torch.zeros(C, device="cuda") if affine else None
        db = torch.zeros(C, device="cuda") if affine else None
        ds = torch.zeros(C, device="cuda") if xs else None
        kw = dict(drop=(0.1, 7), drop_mode=2) if mode == 2 else {}
        bb = M * C * 2 * (4 if mode == 2 else 3)
        for sname, st in settings:
            if st.get("k31", 0) == 8:
                continue
            ops.debug_set(30, st.get("k30", 3)); ops.debug_set(31, st.get("k31", 0)); ops.debug_set(32, st.get("k32", 0)); ops.debug_set(34, st.get("k33", 0))
            us = timed(lambda: ops.layernorm_bwd(dy, x, w, b, mean, rstd, act=act, dgamma=dg, dbeta=db, dxsum=ds, out=dx, **kw))
            print(f"bwd {name:18s} M={M:8d} C={C:5d} {sname:16s} {us:8.1f} us {bb / us / 1e6:6.2f} TB/s", flush=True)
        ops.debug_set(30, 3); ops.debug_set(31, 0); ops.debug_set(32, 0); ops.debug_set(33, 0); ops.debug_set(34, 0)
        del x, dy, dx, y, pz


if __name__ == "__main__":
    main()
