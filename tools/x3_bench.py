#!/usr/bin/env python3
"""The bf16x3 parity mode alone (what bench.py's `parity_mode` leg times): optimizer steps of 8 clips, large model."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from animal2vec_amd.config import large_config  # noqa: E402
from animal2vec_amd.trainer import PretrainTrainer  # noqa: E402


def main():
    cfg = large_config()
    dev = torch.device("cuda", 0)
    x3 = "--f32" not in sys.argv
    tr = PretrainTrainer(cfg, dtype=torch.float32, device=dev, seed=0, x3=x3)
    mbs = [bench.make_microbatch(cfg, 8, 80000, 0, 4242, dev, masks="sampled")]
    tr.prime_memory(8, 80000)
    tr.train_step(mbs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        tr.train_step(mbs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"{'bf16x3' if x3 else 'f32 MFMA'} parity mode: {1e3 * dt:.1f} ms per 8-clip step = {80 / dt:.1f} audio-s/s")


if __name__ == "__main__":
    main()
