"""ExpandedModelCriterion (reference nn/criterions.py:388-502 on top of fairseq's ModelCriterion):
the model supplies the losses; the criterion sums them, copies ``log_keys`` and toggles ``can_sum``."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .registry import register_criterion


@dataclass
class ExpandedModelCriterionConfig:
    loss_weights: Dict[str, float] = field(default_factory=dict)
    log_keys: List[str] = field(default_factory=list)
    can_sum: bool = True


@register_criterion("expanded_model", dataclass=ExpandedModelCriterionConfig)
class ExpandedModelCriterion:
    def __init__(self, task=None, loss_weights=None, log_keys=None, can_sum=True):
        self.task = task
        self.loss_weights = loss_weights or {}
        self.log_keys = log_keys or []
        self.can_sum_original = can_sum
        self.can_sum = can_sum

    def __call__(self, model, sample, reduce=True):
        return self.forward(model, sample, reduce)

    def forward(self, model, sample, reduce=True):
        """criterions.py:405-411 + fairseq ModelCriterion.forward: loss = sum_k coef_k * losses[k].sum()"""
        if model.training:
            self.can_sum = self.can_sum_original
        net_output = model(**sample["net_input"])
        scaled = {}
        for lk, p in net_output["losses"].items():
            coef = 1.0 if not self.loss_weights else self.loss_weights.get(lk, 1.0)
            if coef != 0 and p is not None:
                scaled[lk] = coef * p.float().sum()
        loss = sum(scaled.values())
        sample_size = net_output.get("sample_size", sample["net_input"]["source"].shape[0])
        logging_output = {"loss": loss.data, "ntokens": sample_size, "nsentences": sample["net_input"]["source"].shape[0],
                          "sample_size": sample_size, "_world_size": 1}
        for lk in self.log_keys:
            if lk in net_output and net_output[lk] is not None:
                v = net_output[lk]
                logging_output[lk] = float(v) if not torch.is_tensor(v) or v.numel() == 1 else v
        for lk, v in scaled.items():
            logging_output[f"loss_{lk}"] = v.data
        if not model.training:
            self.can_sum = False
        return loss, sample_size, logging_output

    def logging_outputs_can_be_summed(self) -> bool:
        return self.can_sum

    @staticmethod
    def reduce_metrics(logging_outputs) -> None:
        """criterions.py:413-502: loss per sample, counts, every `loss_*` key per sample, every other logged key as the
        mean over workers (`_world_size` sums to the worker count), the `pretrain/*` counters as sums with the derived
        accuracy / precision / recall / F1 (they exist only with linear-eval labels during pretraining)."""
        from . import metrics
        item = lambda v: float(v.item()) if hasattr(v, "item") else float(v)
        tot = lambda key: item(sum(log.get(key, 0) for log in logging_outputs))
        loss_sum, ntokens, nsentences, sample_size = tot("loss"), tot("ntokens"), tot("nsentences"), tot("sample_size")
        metrics.log_scalar("loss", loss_sum / sample_size, sample_size, round=3)
        metrics.log_scalar("ntokens", ntokens)
        metrics.log_scalar("nsentences", nsentences)
        metrics.log_scalar("sample_size", sample_size)
        builtin = {"loss", "ntokens", "nsentences", "sample_size", "_world_size"}
        world_size = tot("_world_size")
        for k in logging_outputs[0]:
            if k in builtin or k.startswith("_"):
                continue
            val = tot(k)
            if k.startswith("loss_"):
                metrics.log_scalar(k, val / sample_size, sample_size, round=3)
            elif k.startswith("pretrain/"):
                metrics.log_scalar(k, val, round=3)
            else:
                metrics.log_scalar(k, val / world_size, round=3)
        if tot("pretrain/total") > 0:
            _log_prf(metrics, "pretrain")


def _log_prf(metrics, prefix):
    """accuracy / precision / recall / F1 (in per cent) from the summed n_correct, total, tp, fp, fn meters"""
    sr = metrics.safe_round
    s = lambda meters, k: meters[f"{prefix}/{k}"].sum
    nan = float("nan")
    metrics.log_derived(f"metrics/{prefix}/accuracy",
                        lambda m: sr(s(m, "n_correct") * 100.0 / s(m, "total"), 3) if s(m, "total") > 0 else nan)
    metrics.log_derived(f"metrics/{prefix}/precision",
                        lambda m: sr(s(m, "tp") * 100.0 / (s(m, "tp") + s(m, "fp")), 3) if s(m, "tp") + s(m, "fp") > 0 else nan)
    metrics.log_derived(f"metrics/{prefix}/recall",
                        lambda m: sr(s(m, "tp") * 100.0 / (s(m, "tp") + s(m, "fn")), 3) if s(m, "tp") + s(m, "fn") > 0 else nan)
    metrics.log_derived(f"metrics/{prefix}/f1",
                        lambda m: sr(s(m, "tp") * 200.0 / (2 * s(m, "tp") + s(m, "fn") + s(m, "fp")), 3)
                        if 2 * s(m, "tp") + s(m, "fn") + s(m, "fp") > 0 else nan)
