#!/usr/bin/env python3
"""Host enqueue time vs device time of one micro-batch (8 clips, large model, bf16): is the step launch-bound?
(If it were, the per-micro-batch launch sequence would be a candidate for hipGraph capture.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd.config import large_config  # noqa: E402
from animal2vec_amd.trainer import PretrainTrainer  # noqa: E402
from bench import make_microbatch  # noqa: E402


def main():
    cfg = large_config()
    tr = PretrainTrainer(cfg, dtype=torch.bfloat16, device=torch.device("cuda", 0), seed=0)
    mb = make_microbatch(cfg, 8, 80000, 140, 1234, torch.device("cuda", 0))
    T = cfg.frame_counts(80000)[-1]
    noise = torch.zeros(8 * cfg.clone_batch, T - 140, cfg.embed_dim, device="cuda")
    eng = tr.engine
    for _ in range(2):
        eng.forward_backward(mb["source"], mb["mask_u8"], mb["ids_keep"], mb["ids_restore"], noise, dropout_seed=1)
    torch.cuda.synchronize()
    host, dev = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.forward_backward(mb["source"], mb["mask_u8"], mb["ids_keep"], mb["ids_restore"], noise, dropout_seed=1)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0)
        dev.append(t2 - t0)
    host.sort(); dev.sort()
    print(f"host enqueue {1e3 * host[2]:.1f} ms, until device idle {1e3 * dev[2]:.1f} ms per micro-batch of 8 clips "
          f"({'launch' if host[2] > 0.9 * dev[2] else 'device'}-bound)")


if __name__ == "__main__":
    main()
