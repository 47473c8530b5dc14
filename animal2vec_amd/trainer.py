"""Native data-parallel pretraining step: micro-batch fwd/bwd (engine.py) -> bucketed RCCL all-reduce of
the flat gradient buffer overlapped with backward -> clip + Adam (one launch) -> EMA teacher (one
launch) -> operand re-pack.  Replaces, for the hot path only, what fairseq's Trainer.train_step +
LegacyDDP + EMAModule do around the reference model (nn/audio_train_routine.py:330-334,
nn/data2vec2.py:386-410); no control plane (checkpoint policy, logging, LR schedules beyond
warm-up/cosine) is rebuilt here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch
import torch.distributed as dist

from . import ops
from .config import Data2VecMultiConfig
from .ddp import GradSync, grad_scale_factor, make_buckets
from .engine import Engine
from .masking import DeviceMaskSampler, mask_seed
from .params import P, ParamStore, bucket_starts


@dataclass
class OptimConfig:
    """configs/MeerKAT/a2v_large_pretrain_best.yaml:61-81"""
    lr: float = 1e-4
    adam_betas: Tuple[float, float] = (0.9, 0.98)
    adam_eps: float = 1e-6
    weight_decay: float = 0.01
    clip_norm: float = 1.0
    warmup_updates: int = 10000
    max_update: int = 384230


def get_annealed_rate(start, end, curr_step, total_steps):
    """reference nn/modalities/base.py:492-497"""
    if curr_step >= total_steps:
        return end
    r = end - start
    return end - r * (1 - curr_step / total_steps)


class PretrainTrainer:
    def __init__(self, cfg: Data2VecMultiConfig, optim: Optional[OptimConfig] = None, dtype=torch.bfloat16,
                 device="cuda", seed: int = 0, process_group=None, dropout: bool = True, mixup: bool = True, x3: bool = False,
                 grad_wire=None):
        self.cfg, self.optim = cfg, optim or OptimConfig()
        if not cfg.audio.encoder_zero_mask:
            # (the features-only / finetune paths replace masked frames by noise when this is off, model._features_only; the
            # pretraining engine zero-masks through the row mask of its first pos-conv layer and has no noise form)
            raise NotImplementedError("pretraining with modalities.audio.encoder_zero_mask=False is not built "
                                      "(the shipped recipe zero-masks, a2v_large_pretrain_best.yaml)")
        if cfg.layerdrop > 0 or cfg.audio.prenet_layerdrop > 0:
            # (the features-only / finetune paths draw layerdrop, model._features_only and finetune.make_aug; the pretraining
            #  forward + hand-written backward schedule runs every block)
            raise NotImplementedError(f"pretraining with layerdrop={cfg.layerdrop} / prenet_layerdrop={cfg.audio.prenet_layerdrop}: "
                                      "block dropping in the pretraining step (data2vec2.py:651-655, modules.py:89-94) is not "
                                      "built; every shipped pretraining recipe uses 0")
        self.store = ParamStore(cfg, device=device)
        self.store.init_reference(seed)
        self.engine = Engine(cfg, self.store, dtype=dtype, x3=x3)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.num_updates = 0
        self.ema_decay = cfg.ema_decay
        self.dropout = dropout
        self.mixup = ops.Mixup(cfg.audio.sample_rate, cfg.mixing_window_length, self.store.device) \
            if (cfg.source_mixup >= 0 and cfg.mixup_prob > 0 and mixup) else None
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.seed = int(seed)
        # Every random draw of a step -- span masks, mixup strength / permutation / per-clip bernoulli, mask-token noise,
        # dropout -- is a pure function of (seed, num_updates, micro-batch index, rank): nothing to checkpoint, and a
        # resumed run draws on every rank what the uninterrupted one would have.  The device generator is re-seeded per
        # micro-batch (a host-side counter write; the permutation itself is drawn on the device: a pageable
        # host->device copy per micro-batch would make the host wait for the stream to drain).
        self._perm_gen = torch.Generator(device=self.store.device)
        a = cfg.audio
        self.mask_sampler = None
        if self.store.device.type == "cuda":
            self.mask_sampler = DeviceMaskSampler(self.store.device, mask_prob=a.mask_prob, mask_length=a.mask_length,
                                                  mask_dropout=a.mask_dropout, add_masks=a.add_masks)
        self.sync = GradSync(self.store.grad, self._make_buckets(), process_group, wire_dtype=grad_wire)     # (ddp.GradSync: wire dtype)
        self.engine.on_grad_ready = self.sync.on_ready
        dev = self.store.device
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._gscale = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last = {}

    # ------------------------------------------------------------------ gradient buckets
    def _make_buckets(self) -> Dict[str, Tuple[int, int]]:
        """Contiguous slices of the flat gradient buffer, keyed by the engine's grad-ready events."""
        bounds = bucket_starts(self.cfg, self.store.specs)
        return make_buckets(bounds, self.store.total)

    # ------------------------------------------------------------------ schedules
    def lr_at(self, step: int) -> float:
        o = self.optim
        if step < o.warmup_updates:
            return o.lr * step / max(1, o.warmup_updates)
        t = (step - o.warmup_updates) / max(1, o.max_update - o.warmup_updates)
        return 0.5 * o.lr * (1 + math.cos(math.pi * min(1.0, t)))

    def ema_decay_at(self, num_updates: int) -> float:
        """data2vec2.py:395-406"""
        c = self.cfg
        if c.ema_decay == c.ema_end_decay:
            return c.ema_decay
        if num_updates >= c.ema_anneal_end_step:
            return c.ema_end_decay
        return get_annealed_rate(c.ema_decay, c.ema_end_decay, num_updates, c.ema_anneal_end_step)

    # ------------------------------------------------------------------ one optimizer step
    def train_step(self, micro_batches: Sequence[dict]) -> None:
        """micro_batches: dicts with source (B,L) fp32 and, optionally, mask_u8 / ids_keep / ids_restore (a caller-supplied
        `precomputed_mask`; absent: the step samples its span masks itself, on the device, as the reference's forward does
        on the host -- base.py:401-455) and `noise` (N,T-K,D) fp32 (absent / None: N(0, mask_noise_std) is drawn on the
        device, inside the decoder-input kernel)."""
        st, eng, o = self.store, self.engine, self.optim
        st.grad.zero_()
        eng.stats.zero_()
        nmb = len(micro_batches)
        masks = self._step_masks(micro_batches)
        for j, mb in enumerate(micro_batches):
            self.sync.arm(j == nmb - 1)
            noise = mb.get("noise")
            if noise is None:                 # drawn inside the decoder-input kernel: (std, seed), seed from the step counters
                noise = (self.cfg.audio.mask_noise_std,
                         ((self.seed * 7919 + 31) + (self.num_updates * 131 + j) * 8191 + self.rank) & 0x7FFFFFFF)
            dseed = ((self.num_updates * 131 + j) * 257 + self.rank + 1) if self.dropout else None
            src = mb["source"]
            if self.mixup is not None:        # BC-learning mixup, train only (data2vec2.py:536-598)
                cs = mask_seed(self.seed + 17, self.num_updates, j, self.rank)
                r = max(1e-6, self.cfg.source_mixup)
                r = r + (1.0 - r) * ((cs >> 11) / float(1 << 53))               # U[source_mixup, 1), one per micro-batch
                self._perm_gen.manual_seed(cs & 0x7FFFFFFFFFFFFFFF)
                perm = torch.rand(src.shape[0], device=src.device, generator=self._perm_gen).argsort().to(torch.int32)
                mm = None
                if self.cfg.mixup_prob < 1:            # bernoulli(mixup_prob) per clip (data2vec2.py:540-545)
                    mm = (torch.rand(src.shape[0], device=src.device, generator=self._perm_gen) < self.cfg.mixup_prob).to(torch.uint8)
                src = self.mixup(src, perm, r, mm)
            mk = masks[j] if masks[j] is not None else (mb["mask_u8"], mb["ids_keep"], mb["ids_restore"])
            eng.forward_backward(src, mk[0], mk[1], mk[2], noise, dropout_seed=dseed)
        if self.mask_sampler is not None:
            self.mask_sampler.end_step()
        self.sync.arm(False)
        D = self.cfg.embed_dim
        self.sync.finish(eng.stats)            # wait buckets; loss / count / variance sums in one packed call
        # grads <- sum over ranks; scale by 1/sample_size, clip to clip_norm (fairseq Trainer semantics)
        self._sumsq.zero_()
        ops.sumsq(st.grad, self._sumsq)
        scale, gnorm = grad_scale_factor(self._sumsq[0], eng.stats[4 * D + 1], o.clip_norm)
        self._gscale.copy_(scale.reshape(1))
        self.num_updates += 1
        ops.adam_step(st.flat, st.grad, st.adam_m, st.adam_v, lr=self.lr_at(self.num_updates), beta1=o.adam_betas[0],
                      beta2=o.adam_betas[1], eps=o.adam_eps, weight_decay=o.weight_decay, step=self.num_updates,
                      decay_mask=st.decay_mask, grad_scale_dev=self._gscale)
        eng.refresh_student_packs()
        # EMA teacher (set_num_updates, data2vec2.py:386-410)
        self.ema_decay = self.ema_decay_at(self.num_updates)
        if self.ema_decay < 1:
            if eng.tstream is not None:
                # the teacher's parameters and operand images are only read by the teacher forward, which runs on the
                # side stream: update them there, under the next step's front end / student prelude on the main stream
                main = torch.cuda.current_stream()
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(eng.tstream):
                    eng.tstream.wait_event(ev)
                    ops.ema_update(st.teacher, st.flat[st.sh_lo:st.sh_hi], self.ema_decay)
                    eng.refresh_teacher_packs()
            else:
                ops.ema_update(st.teacher, st.flat[st.sh_lo:st.sh_hi], self.ema_decay)
                eng.refresh_teacher_packs()
        self.last = dict(stats=eng.stats, gnorm=gnorm)

    def prime_memory(self, clips: int, samples: int) -> int:
        """The kept-frame count K of a step is drawn with its masks, and every activation of the student is K rows long per
        sequence: steps with a K the allocator has not seen yet grow its pools (hipMalloc of tens of GiB: seconds on a fresh
        process) and fragment them (169 -> 215 -> 236 GiB reserved over ten 64-clip steps, measured).  Two measures: the student's
        tensors of this shape are allocated at a fixed capacity K_cap from now on (ops.set_row_cap: allocation sizes stop depending on
        the draw), and one forward / backward at K_cap -- gradients and statistics discarded -- sizes the pools once.  Returns K_cap."""
        from .masking import make_maskinfo, seeded_keep_mask
        cfg, eng = self.cfg, self.engine
        T = cfg.frame_counts(samples)[-1]
        N = clips * cfg.clone_batch
        self.mask_sampler.take(N, T, mask_seed(self.seed + 1, 0, 0, self.rank))      # one draw of this shape: the level of K
        self.mask_sampler.end_step()
        k0 = self.mask_sampler.last_k
        k_cap = min(T - 1, k0 + max(12, k0 // 8))         # (T = 2000, 768 rows: 150 + 18; draws above it are allocated exactly)
        eng.k_cap[(N, T)] = k_cap                          # from here on the student's tensors of this shape are allocated at k_cap rows
        m, keep, restore = make_maskinfo(seeded_keep_mask(N, T, k_cap, 12345, device=self.store.device))
        src = torch.zeros(clips, samples, dtype=torch.float32, device=self.store.device)
        src[:, ::7] = 1.0
        eng.stats.zero_()
        eng.forward_backward(src, m, keep, restore, (cfg.audio.mask_noise_std, 1), dropout_seed=1 if self.dropout else None)
        self.store.grad.zero_()
        eng.stats.zero_()
        return k_cap

    def _step_masks(self, micro_batches):
        """(mask_u8, ids_keep, ids_restore) per micro-batch that brings no mask of its own: taken from the device sampler
        (requested during the previous step -- same shapes assumed, else drawn now), and the next step's requested."""
        need = [j for j, mb in enumerate(micro_batches) if mb.get("mask_u8") is None]
        out = [None] * len(micro_batches)
        if not need:
            return out
        if self.mask_sampler is None:
            raise RuntimeError("micro-batches without masks need the device sampler (a GPU): there is no host path")
        cb = self.cfg.clone_batch
        shapes = {}
        for j in need:
            B, L = micro_batches[j]["source"].shape
            shapes[j] = (B * cb, self.cfg.frame_counts(L)[-1])
            out[j] = self.mask_sampler.take(*shapes[j], mask_seed(self.seed, self.num_updates, j, self.rank))
        for j in need:
            self.mask_sampler.request(*shapes[j], mask_seed(self.seed, self.num_updates + 1, j, self.rank))
        return out

    # ------------------------------------------------------------------ exact resume
    def optimizer_state(self, extra: Optional[dict] = None) -> dict:
        """What fairseq keeps under "last_optimizer_state" (Adam moments, step count) plus this trainer's own counters.  No
        random stream is part of it: masks, mixup draws, mask-token noise and dropout are functions of (seed, num_updates,
        micro-batch, rank).  `extra` carries what the caller owns (data cursor: train_routine.py).  Layout note: this is NOT fairseq's
        {"state": {param id: {exp_avg, exp_avg_sq, step}}, "param_groups"} dict -- the moments are the two flat fp32
        buffers in the flat parameter order -- so the reference can load the MODEL from these files but must start its
        optimizer afresh (`--reset-optimizer`)."""
        st = self.store
        out = {"adam_m": st.adam_m.detach().cpu(), "adam_v": st.adam_v.detach().cpu(), "num_updates": self.num_updates,
               "ema_decay": self.ema_decay, "seed": self.seed}
        if extra:
            out["extra"] = dict(extra)
        return out

    def load_optimizer_state(self, state: dict) -> None:
        st = self.store
        st.adam_m.copy_(state["adam_m"].to(st.device))
        st.adam_v.copy_(state["adam_v"].to(st.device))
        self.num_updates = int(state["num_updates"])
        self.ema_decay = float(state.get("ema_decay", self.ema_decay))
        self.seed = int(state.get("seed", self.seed))        # the draws are functions of (seed, counters): the seed travels

    # ------------------------------------------------------------------ logging-side scalars (sync)
    def metrics(self) -> Dict[str, float]:
        """loss / sample_size, pred_var, target_var (data2vec2.py:956-970,1095-1110) -- host sync.  Also where the
        representation-collapse guard of the reference's forward runs (data2vec2.py:972-988: abort when target_var / pred_var
        fall under their minima after 5000 updates): the step itself never synchronises, so a loop that wants the guard must
        call metrics() (animal2vec_audio_main does, every update); a loop that never does -- bench.py -- runs without it."""
        D = self.cfg.embed_dim
        s = self.last["stats"].cpu()
        n = float(s[4 * D + 1])

        def var(off):
            a, b = s[off:off + D], s[off + D:off + 2 * D]
            return float(torch.sqrt((b - a * a / n) / (n - 1) + 1e-6).mean())

        out = dict(loss=float(s[4 * D]) / n, sample_size=n, pred_var=var(0), target_var=var(2 * D),
                   gnorm=float(self.last["gnorm"]), ema_decay=self.ema_decay * 1000)
        # representation-collapse guard (data2vec2.py:972-988); evaluated here because this is where the variances
        # reach the host -- the step itself never synchronises
        c = self.cfg
        if self.num_updates > 5000:
            if out["target_var"] < c.min_target_var:
                raise Exception(f"target var is {out['target_var']} < {c.min_target_var}, exiting (AUDIO)")
            if out["pred_var"] < c.min_pred_var:
                raise Exception(f"pred_var is {out['pred_var']} < {c.min_pred_var}, exiting (AUDIO)")
        return out
