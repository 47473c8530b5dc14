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
    if "with_labels" in t:                         # (model.with_labels = II("task.with_labels"), data2vec2.py:146)
        mk.setdefault("with_labels", bool(t["with_labels"]))
    cfg = Data2VecMultiConfig(audio=audio, **mk)   # raises on reference switches that are not built (config.UNBUILT)
    g = y.get("optimizer", {}).get("groups", {}).get("default", {})
    oc = OptimConfig(lr=g.get("lr_float", 1e-4), adam_betas=tuple(g.get("optimizer", {}).get("adam_betas", (0.9, 0.98))),
                     adam_eps=float(g.get("optimizer", {}).get("adam_eps", 1e-6)),
                     weight_decay=g.get("optimizer", {}).get("weight_decay", 0.01), clip_norm=o.get("clip_norm", 1.0),
                     warmup_updates=g.get("lr_scheduler", {}).get("warmup_updates", 10000),
                     max_update=o.get("max_update", 384230))
    run = dict(max_tokens=y.get("dataset", {}).get("max_tokens", 408000), update_freq=(o.get("update_freq") or [1])[0],
               sample_rate=t.get("sample_rate", 8000))
    return cfg, oc, run


def finetune_config_from_yaml(path: str):
    """Map a reference finetune recipe (configs/MeerKAT/finetune_mixup_{001,025,100}.yaml) onto FinetuneConfig plus the
    run-level settings around it.  Every `model:` / `optimization:` / `optimizer:` / `lr_scheduler:` field the finetune
    hot path reads is taken from the file (wav2vec2.py:94-131 push the model fields into the encoder config); fields of
    the reference's control plane (checkpoint policy, validation cadence, tensorboard) are returned untouched under
    run["yaml"] for whoever drives the loop."""
    from .finetune import FinetuneConfig
    y = yaml.safe_load(open(path))
    m, t, o = y.get("model", {}), y.get("task", {}), y.get("optimization", {})
    opt, sch, crit = y.get("optimizer", {}), y.get("lr_scheduler", {}), y.get("criterion", {})
    labels = t.get("unique_labels", "[]")
    labels = list(__import__("ast").literal_eval(labels)) if isinstance(labels, str) else list(labels)
    betas = opt.get("adam_betas", (0.9, 0.98))
    if isinstance(betas, str):
        betas = tuple(float(v) for v in betas.strip("()[] ").split(","))
    d = FinetuneConfig()
    pick = lambda src, k, cast, default: cast(src[k]) if k in src and src[k] is not None else default
    f = FinetuneConfig(
        n_labels=len(labels) or d.n_labels,
        apply_mask=pick(m, "apply_mask", bool, d.apply_mask), mask_prob=pick(m, "mask_prob", float, d.mask_prob),
        mask_length=pick(m, "mask_length", int, d.mask_length),
        mask_channel_prob=pick(m, "mask_channel_prob", float, d.mask_channel_prob),
        mask_channel_length=pick(m, "mask_channel_length", int, d.mask_channel_length),
        zero_mask=pick(m, "zero_mask", bool, d.zero_mask), dropout=pick(m, "dropout", float, d.dropout),
        activation_dropout=pick(m, "activation_dropout", float, d.activation_dropout),
        attention_dropout=pick(m, "attention_dropout", float, d.attention_dropout),
        dropout_input=pick(m, "dropout_input", float, d.dropout_input), final_dropout=pick(m, "final_dropout", float, d.final_dropout),
        layerdrop=pick(m, "layerdrop", float, d.layerdrop),
        freeze_finetune_updates=pick(m, "freeze_finetune_updates", int, d.freeze_finetune_updates),
        layer_decay=pick(m, "layer_decay", float, d.layer_decay), target_mixup=pick(m, "target_mixup", bool, d.target_mixup),
        source_mixup=pick(m, "source_mixup", float, d.source_mixup), mixup_prob=pick(m, "mixup_prob", float, d.mixup_prob),
        same_mixup=pick(m, "same_mixup", bool, d.same_mixup),
        mixing_window_length=pick(m, "mixing_window_length", float, d.mixing_window_length),
        lr=float((o.get("lr") or [d.lr])[0]), adam_betas=tuple(betas), adam_eps=float(opt.get("adam_eps", d.adam_eps)),
        weight_decay=float(opt.get("weight_decay", d.weight_decay)), clip_norm=float(o.get("clip_norm", d.clip_norm)),
        warmup_updates=pick(sch, "warmup_updates", int, d.warmup_updates),
        warmup_init_lr=pick(sch, "warmup_init_lr", float, d.warmup_init_lr), min_lr=pick(sch, "min_lr", float, d.min_lr),
        max_update=pick(o, "max_update", int, d.max_update))
    if not f.same_mixup:
        raise NotImplementedError("same_mixup: false (one mixing strength per clip) is not used by any shipped recipe")
    if m.get("gain_mode", "A_weighting") != "A_weighting":
        raise NotImplementedError(f"gain_mode {m.get('gain_mode')!r}: only A_weighting (every shipped recipe) is built")
    if float(m.get("feature_grad_mult", 0.0)) != 0.0:
        raise NotImplementedError("feature_grad_mult != 0: the finetune path keeps the conv feature extractor frozen")
    run = dict(unique_labels=labels, max_tokens=int(y.get("dataset", {}).get("max_tokens", 426_667)),
               update_freq=int((o.get("update_freq") or [1])[0]), sample_rate=int(t.get("sample_rate", 8000)),
               average_top_k_layers=m.get("average_top_k_layers"), min_label_size=int(t.get("min_label_size", 0)),
               min_sample_size=int(t.get("min_sample_size", 0)), max_sample_size=t.get("max_sample_size"),
               train_subset=y.get("dataset", {}).get("train_subset", "train"),
               metric_threshold=float(crit.get("metric_threshold", 0.5)), use_focal_loss=bool(crit.get("use_focal_loss", True)),
               conv_feature_layers=t.get("conv_feature_layers"), yaml=y)
    return f, run


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
    u0 = 0
    if restore_file:
        los = state.get("last_optimizer_state")
        if isinstance(los, dict) and "adam_m" in los:
            # exact resume: moments and counters.  Every random draw of a step (span masks, mixup, mask-token noise, dropout)
            # is a function of (seed, num_updates, micro-batch, rank) -- no RNG state travels, each rank re-derives its own
            tr.load_optimizer_state(los)
            u0 = int((los.get("extra") or {}).get("update_cursor", 0))
    real = None
    primed = False
    for u in range(u0, u0 + updates):
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
            if not primed:                                  # size the allocator once per run (trainer.prime_memory)
                tr.prime_memory(src.shape[0], src.shape[1])
                primed = True
            mbs.append(dict(source=src, noise=None))        # span masks: sampled by the step itself, on the device (masking.py)
        if not mbs:
            continue
        t0 = time.perf_counter()
        tr.train_step(mbs)
        met = tr.metrics()
        if rank == 0:
            log({"num_updates": tr.num_updates, "wall_s": round(time.perf_counter() - t0, 3), **{k: round(v, 5) for k, v in met.items()}})
    if save_path and rank == 0:
        from . import checkpoint as ck
        # (synthetic batches are a function of the update index; a manifest epoch restarts at its first batch on resume)
        ck.save_checkpoint(save_path, tr.store, cfg={"model": {"_name": "data2vec_multi"}}, num_updates=tr.num_updates,
                           optimizer_state=tr.optimizer_state(extra={"update_cursor": u0 + updates}))
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
