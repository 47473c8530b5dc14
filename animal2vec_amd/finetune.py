"""Finetune path: the multi-label classification head and focal-loss step on top of the
pretrained encoder -- the part of Wav2VecEncoderModOut / Wav2VecCcasFinetune / FinetuneCrossEntropyCriterion
(reference nn/wav2vec2.py:91-482, nn/criterions.py:137-277) that sits on the compute path.  Eval-form forward
(no masking, no layerdrop, dropouts 0) with full backward; the conv feature extractor is frozen
(feature_grad_mult 0.0, configs/MeerKAT/finetune_mixup_100.yaml:85).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .engine import Engine


class FinetuneHead:
    """Linear(D -> n_labels) (wav2vec2.py:203-205, fairseq `Linear`: xavier-uniform weight, zero bias) kept as fp32
    parameters plus MFMA operand images padded to 16 outputs (the GEMM kernels want N % 8 == 0)."""

    def __init__(self, embed_dim: int, n_labels: int, dtype=torch.bfloat16, device="cuda", seed: int = 0,
                 alpha: float = 0.25, gamma: float = 2.0):
        self.D, self.n_labels, self.dtype = embed_dim, n_labels, dtype
        self.Cp = (n_labels + 15) // 16 * 16
        self.alpha, self.gamma = alpha, gamma
        gen = torch.Generator().manual_seed(seed)
        bound = (6.0 / (embed_dim + n_labels)) ** 0.5
        self.w = ((torch.rand(n_labels, embed_dim, generator=gen) * 2 - 1) * bound).to(device)
        self.b = torch.zeros(n_labels, device=device)
        self.w_op = torch.zeros(self.Cp, embed_dim, dtype=dtype, device=device)
        self.wt_op = torch.zeros(embed_dim, self.Cp, dtype=dtype, device=device)
        self.b_op = torch.zeros(self.Cp, dtype=torch.float32, device=device)
        self.gw = torch.zeros(self.Cp, embed_dim, dtype=torch.float32, device=device)
        self.gb = torch.zeros(self.Cp, dtype=torch.float32, device=device)
        self.loss_sum = torch.zeros(1, dtype=torch.float64, device=device)
        self.refresh()

    def load(self, w: torch.Tensor, b: torch.Tensor):
        self.w.copy_(w)
        self.b.copy_(b)
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """rebuild the operand images after an optimizer step"""
        C = self.n_labels
        self.w_op[:C].copy_(self.w)
        self.wt_op[:, :C].copy_(self.w.t())
        self.b_op[:C].copy_(self.b)

    def zero_grad(self):
        self.gw.zero_()
        self.gb.zero_()
        self.loss_sum.zero_()

    @property
    def grad_w(self):
        return self.gw[: self.n_labels]

    @property
    def grad_b(self):
        return self.gb[: self.n_labels]


def finetune_step(engine: Engine, head: FinetuneHead, source: torch.Tensor, target: torch.Tensor, *, backward: bool = True,
                  gscale: float = 1.0, top_k: Optional[int] = None):
    """One micro-batch of the finetune objective: returns (logits (B,T,C), loss_sum so far as a device scalar).
    Gradients accumulate into engine.store.grad (encoder, project_features) and head.gw / head.gb."""
    logits = engine.finetune_forward_backward(source, target, head, backward=backward, gscale=gscale, top_k=top_k)
    return logits, head.loss_sum
