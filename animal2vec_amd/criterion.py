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
