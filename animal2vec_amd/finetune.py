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


# ---------------------------------------------------------------------------------------------- drop-in surface
import torch.nn as tnn  # noqa: E402

from .registry import register_criterion, register_model  # noqa: E402


class _FinetuneLogits(torch.autograd.Function):
    """autograd bridge: forward runs the HIP forward (keeping its saved state in the engine), backward receives
    d loss / d logits from whatever criterion sits on top and runs the hand-written backward into the flat gradient
    buffer (encoder) and the head's gradient buffers."""

    @staticmethod
    def forward(ctx, anchor, model, source):
        ctx.model = model
        eng, head = model.encoder.engine, model.head
        logits = eng.finetune_forward(source, head, save=True)
        B = source.shape[0]
        return logits.view(B, logits.shape[0] // B, -1)[..., : head.n_labels].float()

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        eng, head = model.encoder.engine, model.head
        B, T, C = dlogits.shape
        dl = torch.zeros(B * T, head.Cp, dtype=eng.dtype, device=dlogits.device)
        dl[:, :C] = dlogits.reshape(B * T, C).to(eng.dtype)
        eng.finetune_backward(dl, head)
        model._publish_head_grads()
        return torch.zeros_like(model._anchor), None, None


@register_model("wav2vec_ccas_finetune")
class Wav2VecCcasFinetune(tnn.Module):
    """Call surface of the reference's finetune model (wav2vec2.py:57-90,430-482): forward(source, padding_mask, target)
    -> {"encoder_out": logits (B,T,C), "padding_mask", "layer_results", "target"}; get_logits / get_targets as the
    criterion uses them.  `encoder` is a Data2VecMultiModel of this package (its parameters' .grad are views of the
    flat gradient buffer); `proj.weight/.bias` mirror the head."""

    def __init__(self, encoder, n_labels: int, seed: int = 0, use_focal_loss: bool = True):
        super().__init__()
        self.encoder = encoder
        self.cfg_use_focal_loss = use_focal_loss
        cfg = encoder.cfg
        self.head = FinetuneHead(cfg.embed_dim, n_labels, dtype=encoder.engine.dtype, device=encoder.store.device, seed=seed)
        self.proj = tnn.Linear(cfg.embed_dim, n_labels).to(encoder.store.device)
        with torch.no_grad():
            self.proj.weight.copy_(self.head.w)
            self.proj.bias.copy_(self.head.b)
        self._anchor = tnn.Parameter(torch.zeros(1, device=encoder.store.device))

    def sync_head(self):
        """after an optimizer step on proj.*: rebuild the head's operand images"""
        self.head.load(self.proj.weight.detach(), self.proj.bias.detach())

    def _publish_head_grads(self):
        for prm, gsrc in ((self.proj.weight, self.head.grad_w), (self.proj.bias, self.head.grad_b)):
            if prm.grad is None:
                prm.grad = gsrc.clone()
            else:
                prm.grad.add_(gsrc)
        self.head.gw.zero_()
        self.head.gb.zero_()

    def forward(self, source, padding_mask=None, target=None, **kwargs):
        if padding_mask is not None:
            raise NotImplementedError("enable_padding is false in every shipped config")
        source = source.to(self.encoder.store.device, torch.float32).contiguous()
        if torch.is_grad_enabled() and self.training:
            logits = _FinetuneLogits.apply(self._anchor, self, source)
        else:
            eng = self.encoder.engine
            lg = eng.finetune_forward(source, self.head, save=False)
            B = source.shape[0]
            logits = lg.view(B, lg.shape[0] // B, -1)[..., : self.head.n_labels].float()
        return {"encoder_out": logits, "padding_mask": None, "layer_results": None, "target": target}

    @staticmethod
    def prepare_shapes(out, transpose=True):
        if transpose:
            out = out.transpose(0, 2)
        return out.reshape(-1, out.size(-1))

    def get_logits(self, net_output, reshape=True):
        y = net_output["encoder_out"]
        return self.prepare_shapes(y, False) if reshape else y

    def get_targets(self, sample, net_output, expand_steps=True, reshape=True):
        y = net_output["target"] if net_output.get("target") is not None else sample["target"]
        return self.prepare_shapes(y, transpose=False) if reshape else y


def sigmoid_focal_loss(inputs, targets, alpha: float = 0.25, gamma: float = 2, reduction: str = "none"):
    """nn/utils.py:971-1010 on torch tensors (the criterion's form; the fused training step uses a2v_focal_loss)."""
    inputs, targets = inputs.float(), targets.float()
    pr = torch.sigmoid(inputs)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = pr * targets + (1 - pr) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean() if reduction == "mean" else (loss.sum() if reduction == "sum" else loss)


@register_criterion("finetunecriterion")
class FinetuneCrossEntropyCriterion:
    """The focal-loss branch of the reference criterion (criterions.py:137-277, use_focal_loss: true in every shipped
    finetune config): loss = sum of sigmoid_focal_loss(logits, target); sample_size = ntokens (sentence_avg false)."""

    def __init__(self, task=None, use_focal_loss=True, sentence_avg=False, metric_threshold=0.5, report_accuracy=False):
        if not use_focal_loss:
            raise NotImplementedError("label-smoothed cross entropy branch (single-label) is not on the shipped configs' path")
        self.task, self.sentence_avg, self.metric_threshold, self.report_accuracy = task, sentence_avg, metric_threshold, report_accuracy

    def __call__(self, model, sample, reduce=True):
        return self.forward(model, sample, reduce)

    def forward(self, model, sample, reduce=True):
        net_output = model(**sample["net_input"])
        logits = model.get_logits(net_output)
        target = model.get_targets(sample, net_output).to(logits.device)
        loss = sigmoid_focal_loss(logits, target, reduction="sum" if reduce else "none")
        sample_size = sample["target"].size(0) if self.sentence_avg else sample.get("ntokens", target.shape[0])
        logging_output = {"loss": loss.data, "nll_loss": torch.tensor(0.0), "ntokens": sample.get("ntokens", target.shape[0]),
                          "nsentences": sample["target"].size(0), "sample_size": sample_size}
        if self.report_accuracy:
            preds = (torch.sigmoid(logits) >= self.metric_threshold).to(target.dtype)
            logging_output["finetune/n_correct"] = float((preds == (target >= 0.5).to(target.dtype)).sum())
            logging_output["finetune/total"] = float(target.numel())
        return loss, sample_size, logging_output
