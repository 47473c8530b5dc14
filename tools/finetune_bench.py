#!/usr/bin/env python3
"""Throughput of the finetune step's compute (BASELINE config 4 shape: large model, 10 s clips @ 8 kHz, all 2000
frames, 12 labels): forward + backward of one micro-batch, audio-seconds/s on one GPU.  Synthetic data."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd.config import large_config  # noqa: E402
from animal2vec_amd.engine import Engine  # noqa: E402
from animal2vec_amd.finetune import FinetuneHead, finetune_step  # noqa: E402
from animal2vec_amd.params import ParamStore  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--micro-batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--accum", type=int, default=4, help="micro-batches per optimizer step in the train-form legs")
    a = ap.parse_args()
    cfg = large_config()
    store = ParamStore(cfg, device="cuda")
    store.init_reference(0)
    eng = Engine(cfg, store, dtype=torch.bfloat16)
    head = FinetuneHead(cfg.embed_dim, 12)
    B, L = a.micro_batch, 80000
    g = torch.Generator().manual_seed(0)
    src = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,)).cuda()
    T = cfg.frame_counts(L)[-1]
    tgt = (torch.rand(B, T, 12, generator=g) < 0.1).float().cuda()
    for _ in range(a.warmup):
        store.grad.zero_(); head.zero_grad()
        finetune_step(eng, head, src, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        store.grad.zero_(); head.zero_grad()
        finetune_step(eng, head, src, tgt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"finetune fwd+bwd (eval form): {dt * 1e3:.1f} ms per micro-batch of {B} clips -> {B * L / 8000 / dt:.1f} audio-s/s "
          f"(loss/frame {float(head.loss_sum) / (B * T):.4f})")
    del eng, head
    # train form: the whole optimizer step (BC mixup x2, masking, layerdrop, dropouts, Adam on encoder + head)
    from animal2vec_amd.finetune import FinetuneConfig, FinetuneTrainer
    for freeze, label in ((0, "encoder + head"), (10 ** 9, "frozen encoder (first 10k updates)")):
        tr = FinetuneTrainer(cfg, FinetuneConfig(freeze_finetune_updates=freeze), store=store, dtype=torch.bfloat16)
        mbs = [dict(source=src, target=tgt) for _ in range(a.accum)]
        for _ in range(a.warmup):
            tr.train_step(mbs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tr.train_step(mbs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        m = tr.metrics()
        print(f"finetune train step, {label}: {dt * 1e3:.1f} ms per {a.accum} x {B} clips -> "
              f"{a.accum * B * L / 8000 / dt:.1f} audio-s/s (loss {m['loss']:.4f}, lr {m['lr']:.2e})")
        del tr


if __name__ == "__main__":
    main()
