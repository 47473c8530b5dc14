#!/usr/bin/env python3
"""Same-process A/B of the persistent NT GEMM's tile order (a2v_debug_set(23): 0 = 8 m-tiles x all n-tiles, 1 = W-stationary
n-groups) and, through a2v_debug_set(5/...), other switches: interleaved rounds, median and min per arm (cdna_hip_programming.md
rule 24).  `--one ARM` runs a single arm a few times (for a rocprofv3 --pmc pass: tools/pmc_pass.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops  # noqa: E402


def flavours(dev, M):
    out = {}
    for name, N, K, fl in [("qkv_bias_3072x1024", 3072, 1024, "b"), ("fc1_gelu_dact_4096x1024", 4096, 1024, "fc1"),
                           ("fc2dgrad_4096x1024", 4096, 1024, "fc2d"), ("tfc1_bias_gelu_4096x1024", 4096, 1024, "bg")]:
        x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if fl in ("fc1", "fc2d") else None
        cs = torch.zeros(N, device=dev)
        if fl == "fc2d":
            pre.copy_((torch.randn(M, N, device=dev) * 0.5).bfloat16())
        fn = {"b": lambda x=x, w=w, b=b, o=o: ops.linear(x, w, bias=b, out=o),
              "bg": lambda x=x, w=w, b=b, o=o: ops.linear(x, w, bias=b, gelu=True, out=o),
              "fc1": lambda x=x, w=w, b=b, o=o, pre=pre: ops.linear(x, w, bias=b, gelu=True, pre_out=pre, dact=True, out=o),
              "fc2d": lambda x=x, w=w, o=o, pre=pre, cs=cs: ops.linear(x, w, gelu_bwd_aux=pre, dact=True, colsum=cs, out=o)}[fl]
        out[name] = (fn, 2.0 * M * N * K)
    return out


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    M = int(os.environ.get("AB_M", 116736))
    fl = flavours(dev, M)
    if "--one" in sys.argv:
        arm = int(sys.argv[sys.argv.index("--one") + 1])
        ops.debug_set(23, arm)
        for name, (fn, _) in fl.items():
            for _ in range(4):
                fn()
        torch.cuda.synchronize()
        return
    rounds = int(os.environ.get("AB_ROUNDS", 12))
    for name, (fn, flop) in fl.items():
        ts = {0: [], 1: []}
        for r in range(rounds):
            for arm in (0, 1) if r % 2 == 0 else (1, 0):
                ops.debug_set(23, arm)
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts[arm].append(e0.elapsed_time(e1) * 100)
        ops.debug_set(23, 0)
        for arm in (0, 1):
            t = sorted(ts[arm])
            print(f"{name:28s} tile order {arm}: median {t[len(t)//2]:7.1f} us  min {t[0]:7.1f} us  {flop / t[len(t)//2] / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
