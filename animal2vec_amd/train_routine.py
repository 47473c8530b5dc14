"""`animal2vec_audio_main(cfg)` (reference nn/audio_train_routine.py:48): the entry the reference's
animal2vec_train.py hands control to.  Only the hot loop is rebuilt: setup task/model, then optimizer
steps on the native trainer with synthetic waveforms of the configured shape.  Checkpoint policy,
validation, logging back ends and real-data loading are the reference's control plane (out of scope)."""
from __future__ import annotations

import os
import time
from typing import Optional

import torch
import torch.distributed as dist
import yaml

from .config import D2vAudioConfig, D2vDecoderConfig, Data2VecMultiConfig
from .masking import compute_mask_indices, make_maskinfo
from .trainer import OptimConfig, PretrainTrainer


def config_from_yaml(path: str):
    """Map the reference's hydra YAML (configs/MeerKAT/a2v_large_pretrain_best.yaml) onto the dataclasses."""
    y = yaml.safe_load(open(path))
    m, t, o = y.get("model", {}), y.get("task", {}), y.get("optimization", {})
    au = dict(m.get("modalities", {}).get("audio", {}))
    dec = D2vDecoderConfig(**{k: v for k, v in au.pop("decoder", {}).items() if k in D2vDecoderConfig.__dataclass_fields__})
    au = {k: v for k, v in au.items() if k in D2vAudioConfig.__dataclass_fields__}
    audio = D2vAudioConfig(decoder=dec, conv_feature_layers=t.get("conv_feature_layers", D2vAudioConfig().conv_feature_layers),
                           sample_rate=t.get("sample_rate", 8000), **au)
    mk = {k: v for k, v in m.items() if k in Data2VecMultiConfig.__dataclass_fields__ and k != "audio"}
    cfg = Data2VecMultiConfig(audio=audio, **mk)
    g = y.get("optimizer", {}).get("groups", {}).get("default", {})
    oc = OptimConfig(lr=g.get("lr_float", 1e-4), adam_betas=tuple(g.get("optimizer", {}).get("adam_betas", (0.9, 0.98))),
                     adam_eps=float(g.get("optimizer", {}).get("adam_eps", 1e-6)),
                     weight_decay=g.get("optimizer", {}).get("weight_decay", 0.01), clip_norm=o.get("clip_norm", 1.0),
                     warmup_updates=g.get("lr_scheduler", {}).get("warmup_updates", 10000),
                     max_update=o.get("max_update", 384230))
    run = dict(max_tokens=y.get("dataset", {}).get("max_tokens", 408000), update_freq=(o.get("update_freq") or [1])[0],
               sample_rate=t.get("sample_rate", 8000))
    return cfg, oc, run


def animal2vec_audio_main(cfg, optim: Optional[OptimConfig] = None, *, updates: int = 2, clip_samples: int = 80000,
                          clips_per_micro_batch: int = 5, update_freq: int = 5, dtype=torch.bfloat16, log=print):
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    tr = PretrainTrainer(cfg, optim, dtype=dtype, device=torch.device("cuda", local), seed=cfg.seed)
    rng = __import__("numpy").random.default_rng(cfg.seed + rank)
    T = cfg.frame_counts(clip_samples)[-1]
    for u in range(updates):
        mbs = []
        for j in range(update_freq):
            g = torch.Generator().manual_seed(1000 * u + 10 * j + rank)
            src = torch.nn.functional.layer_norm(torch.randn(clips_per_micro_batch, clip_samples, generator=g), (clip_samples,))
            m = compute_mask_indices((clips_per_micro_batch * cfg.clone_batch, T), cfg.audio.mask_prob, cfg.audio.mask_length, rng)
            mu, keep, restore = make_maskinfo(torch.from_numpy(m).cuda())
            mbs.append(dict(source=src.cuda(), mask_u8=mu, ids_keep=keep, ids_restore=restore, noise=None))
        t0 = time.perf_counter()
        tr.train_step(mbs)
        met = tr.metrics()
        if rank == 0:
            log({"num_updates": tr.num_updates, "wall_s": round(time.perf_counter() - t0, 3), **{k: round(v, 5) for k, v in met.items()}})
    return tr
