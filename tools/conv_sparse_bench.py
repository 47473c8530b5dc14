"""Micro-benchmark of a2v_conv_sparse_in at the pretraining shape (768 clones x 2000 frames, 16 groups x 64, 19 taps, 152 kept frames):
python tools/conv_sparse_bench.py [--debug-set k=v]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animal2vec_amd import ops, masking


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--debug-set", action="append", default=[])
    a = ap.parse_args()
    for kv in a.debug_set:
        k, v = kv.split("=")
        ops.debug_set(int(k), int(v))
    B, cb, T, D, G, k, K = a.clips, 12, 2000, 1024, 16, 19, 152
    N = B * cb
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, D, generator=g).bfloat16().cuda()
    keep = torch.stack([torch.randperm(T, generator=g)[:K].sort().values for _ in range(N)])
    mask = torch.ones(N, T, dtype=torch.uint8); mask.scatter_(1, keep, 0)
    md = mask.cuda()
    rows = (torch.arange(N).unsqueeze(1) * T + keep).reshape(-1).int().cuda()
    wp = (torch.randn(G, 64, k * 64, generator=g) * 0.03).bfloat16().cuda()
    bias = torch.zeros(D).cuda()
    dyk = torch.randn(N * K, D, generator=g).bfloat16().cuda()
    out = torch.empty(N, T, D, dtype=torch.bfloat16, device="cuda")
    res = {}
    res["sparse_mask_us"] = round(timed(lambda: ops.conv1d_cl_sparse_in(x, wp, bias, k=k, pad=k // 2, groups=G, T=T, n_out=D, n_seq=N, rowmask=md, bdiv=cb, out=out), a.iters), 1)
    res["sparse_rows_us"] = round(timed(lambda: ops.conv1d_cl_sparse_in(dyk, wp, None, k=k, pad=k // 2, groups=G, T=T, n_out=D, n_seq=N, rowidx=rows, out=out), a.iters), 1)
    res["dense_mask_us"] = round(timed(lambda: ops.conv1d_cl(x, wp, bias, k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D, rowmask=md, bdiv=cb, n_batch=N, out=out), a.iters), 1)
    # the same layers' weight gradients (a2v_wgrad_sparse.hip): first layer (mask form), last layer (kept-row form)
    gw = torch.zeros(G, 64, k * 64, dtype=torch.float32, device="cuda")
    dy = torch.randn(N, T, D, generator=g).bfloat16().cuda()
    res["wgrad_mask_us"] = round(timed(lambda: ops.conv1d_cl_wgrad(x, dy, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, rowmask=md, bdiv=cb, n_batch=N), a.iters), 1)
    res["wgrad_kept_us"] = round(timed(lambda: ops.conv1d_cl_wgrad(dy, dyk, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, rowidx=rows, rows_kept=True), a.iters), 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
