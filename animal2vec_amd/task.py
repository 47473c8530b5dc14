"""Registration surface of the reference task (nn/audio_tasks.py:41-188): the config fields that reach the model and
the dataset, `load_dataset(split)` serving `<data>/<split>.tsv` through data.AudioLabelDataset (the host-side
restatement of FileAudioLabelDataset, same sample / collater format), `reduce_metrics`, and a synthetic-waveform sample
source with the collater's output format {"id", "net_input": {"source"}} (audio_tasks.py:433-469)."""
from __future__ import annotations

import ast
import os
import warnings
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .config import LARGE_LAYERS_STR, parse_conv_layers
from .registry import register_task


@dataclass
class AudioConfigCCAS:
    data: str = ""
    sample_rate: int = 8000
    normalize: bool = True
    enable_padding: bool = False
    with_labels: bool = False
    conv_feature_layers: str = LARGE_LAYERS_STR
    max_sample_size: int = 80000
    min_sample_size: int = 1
    min_label_size: int = 0
    unique_labels: str = "[]"
    use_focal_loss: bool = True
    do_focal_prediction: bool = True
    segmentation_metrics: bool = False
    verbose_tensorboard_logging: bool = False
    num_batch_buckets: int = 0
    collate_crop: str = "aligned"          # "reference": FileAudioLabelDataset's per-column label crops (data.py collater)


@register_task("audio_ccas", dataclass=AudioConfigCCAS)
class AudioTaskCCAS:
    def __init__(self, cfg: AudioConfigCCAS):
        self.cfg = cfg
        self.datasets: Dict[str, object] = {}
        # audio_tasks.py:96-107: label-side switches only count with with_labels
        wl = bool(getattr(cfg, "with_labels", False))
        ul = getattr(cfg, "unique_labels", None)
        self.unique_labels = (list(ast.literal_eval(ul)) if isinstance(ul, str) else list(ul)) if (ul and wl) else None
        self.use_focal_loss = getattr(cfg, "use_focal_loss", None) if wl else None
        self.segmentation_metrics = getattr(cfg, "segmentation_metrics", None) if wl else None

    # ------------------------------------------------------------------ datasets (audio_tasks.py:109-152)
    def load_dataset(self, split: str, task_cfg=None, **kwargs):
        """`<cfg.data>/<split>.tsv` -> self.datasets[split] (same constructor arguments as the reference passes to
        FileAudioLabelDataset; padding / bucketing are off in every shipped config and not rebuilt)."""
        from . import data as D
        c = task_cfg or self.cfg
        get = (lambda k, d=None: c.get(k, d)) if isinstance(c, dict) else (lambda k, d=None: getattr(c, k, d))
        if get("enable_padding", False):
            raise NotImplementedError("enable_padding is false in every shipped config (and broken upstream, audio.py:168)")
        manifest = os.path.join(self.cfg.data, f"{split}.tsv")
        wl = bool(get("with_labels", False))
        self.datasets[split] = D.AudioLabelDataset(
            manifest, sample_rate=get("sample_rate", self.cfg.sample_rate), max_sample_size=get("max_sample_size"),
            min_sample_size=get("min_sample_size", 0), normalize=get("normalize", False), return_labels=wl,
            unique_labels=self.unique_labels, use_focal_loss=bool(self.use_focal_loss) if self.use_focal_loss is not None else True,
            min_label_size=get("min_label_size", 0), conv_feature_layers=parse_conv_layers(get("conv_feature_layers")),
            do_focal_prediction=get("do_focal_prediction", True), require_label_files=True,
            crop=get("collate_crop", "aligned"))
        return self.datasets[split]

    def dataset(self, split: str):
        if split not in self.datasets:
            raise KeyError("Dataset not loaded: " + split)
        return self.datasets[split]

    def max_positions(self):
        import sys
        return sys.maxsize, sys.maxsize

    def reduce_metrics(self, logging_outputs, criterion):
        """audio_tasks.py:154-188: words / sentences per batch, then the criterion's own aggregation."""
        from . import metrics
        if not any("ntokens" in log for log in logging_outputs):
            warnings.warn("ntokens not found in Criterion logging outputs, cannot log wpb or wps")
        else:
            ntokens = sum(float(log.get("ntokens", 0)) for log in logging_outputs)
            metrics.log_scalar("misc/wpb", ntokens, priority=180, round=1)
            metrics.log_speed("misc/wps", ntokens, priority=90, round=1)
        if not any("nsentences" in log for log in logging_outputs):
            warnings.warn("nsentences not found in Criterion logging outputs, cannot log bsz")
        else:
            metrics.log_scalar("misc/bsz", sum(float(log.get("nsentences", 0)) for log in logging_outputs), priority=190, round=1)
        criterion.__class__.reduce_metrics(logging_outputs)

    @classmethod
    def setup_task(cls, cfg, **kwargs):
        return cls(cfg)

    def build_model(self, model_cfg, **kw):
        from .registry import MODEL_REGISTRY
        return MODEL_REGISTRY["data2vec_multi"].build_model(model_cfg, task=self, **kw)

    def build_criterion(self, cfg=None):
        from .criterion import ExpandedModelCriterion
        cfg = cfg or {}
        return ExpandedModelCriterion(self, cfg.get("loss_weights"), cfg.get("log_keys"), cfg.get("can_sum", True))

    def synthetic_batch(self, batch_size: int, seed: int, device="cpu"):
        """Waveforms as the collater hands them over: (B,L) fp32, per-clip zero-mean / unit-variance
        (`normalize: true`, audio_tasks.py + fairseq RawAudioDataset.postprocess)."""
        g = torch.Generator().manual_seed(seed)
        L = self.cfg.max_sample_size
        src = torch.randn(batch_size, L, generator=g)
        src = torch.nn.functional.layer_norm(src, (L,))
        return {"id": torch.arange(batch_size), "net_input": {"source": src.to(device)}}
