#!/usr/bin/env python3
"""Time (HIP events) the attention kernels alone at the shapes of the default pretrain step, so that a rocprofv3 --pmc pass
over them is seconds, not a whole bench:

  teacher forward : N=32 clips, S=2000 frames (positions = arange), H=16, hd=64   -> attn_fwd_bf16_kernel<64,false,2>
  student forward : N=384 clone-batch rows, S=140 kept frames, dropout on          -> attn_fwd_bf16_kernel<64,true,1>
  student backward: same rows                                                     -> attn_bwd_seq_kernel<true>

    python tools/attn_bench.py [--iters 10] [--which teacher,student]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--which", default="teacher,student")
    ap.add_argument("--student-rows", type=int, default=384, help="32 clips x 12 clones at micro-batch 32")
    ap.add_argument("--student-s", type=int, default=140, help="kept frames per row (152: what the sampler draws at 768 rows)")
    ap.add_argument("--no-drop", action="store_true")
    ap.add_argument("--debug-set", action="append", default=[], help="key=value for a2v_debug_set (A/B experiments)")
    a = ap.parse_args()
    H, hd = 16, 64
    for kv in a.debug_set:
        k, v = kv.split("=")
        ops.debug_set(int(k), int(v))
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda")
    sc = torch.ones(H, device="cuda")
    out = {}
    if "teacher" in a.which:
        N, S = 32, 2000
        qkv = torch.randn(N, S, 3 * H * hd, device="cuda").bfloat16()
        us = timed(lambda: ops.attention_fwd(qkv, None, slopes, sc, H, need_lse=False), a.iters)
        fl = 4.0 * S * S * hd * H * N
        out["teacher_fwd"] = dict(us=round(us, 1), tflops=round(fl / us / 1e6, 1), frac_of_2p5pf=round(fl / us / 1e6 / 2500, 4))
    if "student" in a.which:
        N, S = a.student_rows, a.student_s
        g = torch.Generator().manual_seed(1)
        pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int().cuda()
        qkv = torch.randn(N, S, 3 * H * hd, device="cuda").bfloat16()
        drop = None if a.no_drop else (0.1, 1234)
        o, lse = ops.attention_fwd(qkv, pos, slopes, sc, H, drop=drop)
        us = timed(lambda: ops.attention_fwd(qkv, pos, slopes, sc, H, drop=drop), a.iters)
        out["student_fwd"] = dict(us=round(us, 1))
        do = torch.randn_like(o)
        dal = torch.zeros(H, device="cuda")
        bg = torch.zeros(3 * H * hd, device="cuda")
        us = timed(lambda: ops.attention_bwd(qkv, o, do, lse, pos, slopes, sc, H, dal, drop=drop, bias_grad=bg), a.iters)
        out["student_bwd"] = dict(us=round(us, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
