"""`animal2vec_audio_main(cfg)` (reference nn/audio_train_routine.py:48): the entry the reference's
animal2vec_train.py hands control to.  Only the hot loop is rebuilt: setup task/model, then optimizer
steps on the native trainer -- on synthetic waveforms of the configured shape, or on a manifest of real clips
through data.py (manifest TSV -> wav -> labels -> prefetch) -- and fairseq-style checkpoints through checkpoint.py.
Checkpoint *policy*, validation and logging back ends are the reference's control plane (out of scope)."""
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


def _dist_setup():
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, local, world


def _real_batches(manifest, cfg, *, max_tokens, rank, world, labels, unique_labels=None, seed=0, device=None):
    """one epoch of device batches from a manifest (data.py), sharded over the data-parallel ranks"""
    from . import data as D
    ds = D.AudioLabelDataset(manifest, sample_rate=cfg.audio.sample_rate, conv_feature_layers=cfg.conv_layers,
                             unique_labels=unique_labels, return_labels=labels, normalize=True, min_sample_size=0)
    batches = D.batch_by_size(ds.sizes, max_tokens, shuffle_seed=seed, shard=(rank, world))
    return D.DevicePrefetcher(ds, batches, device=device, depth=2, seed=seed + rank)


def animal2vec_audio_main(cfg, optim: Optional[OptimConfig] = None, *, updates: int = 2, clip_samples: int = 80000,
                          clips_per_micro_batch: int = 5, update_freq: int = 5, dtype=torch.bfloat16, log=print,
                          data: Optional[str] = None, max_tokens: Optional[int] = None, restore_file: Optional[str] = None,
                          save_path: Optional[str] = None):
    """data=None: synthetic waveforms; data=<manifest.tsv>: real clips (sources only -- pretraining is label-free)."""
    rank, local, world = _dist_setup()
    dev = torch.device("cuda", local)
    tr = PretrainTrainer(cfg, optim, dtype=dtype, device=dev, seed=cfg.seed)
    if restore_file:
        from . import checkpoint as ck
        state = ck.load_checkpoint(restore_file)
        ck.load_pretrained(tr.store, state)
        tr.engine.refresh_student_packs()
        tr.engine.refresh_teacher_packs()
        tr.num_updates = int(state.get("optimizer_history", [{}])[-1].get("num_updates", 0))
        if isinstance(state.get("last_optimizer_state"), dict) and "adam_m" in state["last_optimizer_state"]:
            tr.load_optimizer_state(state["last_optimizer_state"])          # exact resume (moments, counters, RNG)
    rng = __import__("numpy").random.default_rng(cfg.seed + rank)
    real = None
    for u in range(updates):
        mbs = []
        for j in range(update_freq):
            if data is not None:
                if real is None:
                    real = iter(_real_batches(data, cfg, max_tokens=max_tokens or clips_per_micro_batch * clip_samples, rank=rank,
                                              world=world, labels=False, seed=cfg.seed + u, device=dev))
                b = next(real, None)
                if b is None:                       # epoch boundary
                    real = None
                    break
                src = b["source"]
            else:
                g = torch.Generator().manual_seed(1000 * u + 10 * j + rank)
                src = torch.nn.functional.layer_norm(torch.randn(clips_per_micro_batch, clip_samples, generator=g),
                                                     (clip_samples,)).cuda()
            T = cfg.frame_counts(src.shape[1])[-1]
            m = compute_mask_indices((src.shape[0] * cfg.clone_batch, T), cfg.audio.mask_prob, cfg.audio.mask_length, rng)
            mu, keep, restore = make_maskinfo(torch.from_numpy(m).cuda())
            mbs.append(dict(source=src, mask_u8=mu, ids_keep=keep, ids_restore=restore, noise=None))
        if not mbs:
            continue
        t0 = time.perf_counter()
        tr.train_step(mbs)
        met = tr.metrics()
        if rank == 0:
            log({"num_updates": tr.num_updates, "wall_s": round(time.perf_counter() - t0, 3), **{k: round(v, 5) for k, v in met.items()}})
    if save_path and rank == 0:
        from . import checkpoint as ck
        ck.save_checkpoint(save_path, tr.store, cfg={"model": {"_name": "data2vec_multi"}}, num_updates=tr.num_updates,
                           optimizer_state=tr.optimizer_state())
    return tr


def animal2vec_finetune_main(cfg, fcfg=None, *, data: str, unique_labels, pretrained: Optional[str] = None, load_ema: bool = False,
                             updates: int = 2, update_freq: int = 1, max_tokens: int = 426_667, dtype=torch.bfloat16, log=print,
                             save_path: Optional[str] = None):
    """The finetune hot loop (configs/MeerKAT/finetune_mixup_100.yaml): pretrained encoder from a fairseq-style
    checkpoint, labelled clips from a manifest, FinetuneTrainer steps, finetuned checkpoint out."""
    from . import checkpoint as ck
    from .finetune import FinetuneConfig, FinetuneTrainer
    from .params import ParamStore
    rank, local, world = _dist_setup()
    dev = torch.device("cuda", local)
    fcfg = fcfg or FinetuneConfig(n_labels=len(unique_labels))
    assert fcfg.n_labels == len(unique_labels)
    store = ParamStore(cfg, device=dev)
    if pretrained:
        ck.load_for_finetune(store, ck.load_checkpoint(pretrained), load_ema=load_ema)
    else:
        store.init_reference(cfg.seed)
    tr = FinetuneTrainer(cfg, fcfg, store=store, dtype=dtype, seed=cfg.seed)
    tr.engine.refresh_student_packs()
    real, epoch = None, 0
    for u in range(updates):
        mbs = []
        while len(mbs) < update_freq:
            if real is None:
                real = iter(_real_batches(data, cfg, max_tokens=max_tokens, rank=rank, world=world, labels=True,
                                          unique_labels=unique_labels, seed=cfg.seed + epoch, device=dev))
                epoch += 1
            b = next(real, None)
            if b is None:
                real = None
                if mbs:
                    break
                continue
            assert b["target"].shape[1] == cfg.frame_counts(b["source"].shape[1])[-1], "label frame rate != model frame rate"
            mbs.append(dict(source=b["source"], target=b["target"]))
        t0 = time.perf_counter()
        tr.train_step(mbs)
        if rank == 0:
            met = tr.metrics()
            log({"num_updates": tr.num_updates, "wall_s": round(time.perf_counter() - t0, 3), "frozen": tr.frozen,
                 **{k: (round(v, 6) if isinstance(v, float) else v) for k, v in met.items()}})
    if save_path and rank == 0:
        ck.save_checkpoint(save_path, tr.store, head=tr.head, cfg={"model": {"_name": "wav2vec_ccas_finetune"}},
                           num_updates=tr.num_updates)
    return tr
