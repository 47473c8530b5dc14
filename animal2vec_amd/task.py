"""Registration surface of the reference task (nn/audio_tasks.py:41-188).  Only what the hot path needs:
the config fields that reach the model and a synthetic-waveform sample source with the collater's output
format {"id", "net_input": {"source"}} (audio_tasks.py:433-469).  Manifest / wav / h5 I/O is a 'next' row."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .config import LARGE_LAYERS_STR
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
    unique_labels: str = "[]"
    verbose_tensorboard_logging: bool = False


@register_task("audio_ccas", dataclass=AudioConfigCCAS)
class AudioTaskCCAS:
    def __init__(self, cfg: AudioConfigCCAS):
        self.cfg = cfg

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
