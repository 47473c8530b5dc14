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
                  gscale: float = 1.0, top_k: Optional[int] = None, aug: Optional[dict] = None, head_only: bool = False):
    """One micro-batch of the finetune objective: returns (logits (B,T,C), loss_sum so far as a device scalar).
    Gradients accumulate into engine.store.grad (encoder, project_features) and head.gw / head.gb.
    aug: train-form options (Engine.finetune_forward); head_only: classifier-only update (freeze_finetune_updates)."""
    logits = engine.finetune_forward_backward(source, target, head, backward=backward, gscale=gscale, top_k=top_k, aug=aug,
                                              head_only=head_only)
    return logits, head.loss_sum


# ---------------------------------------------------------------------------------------------- train form
import math  # noqa: E402
from dataclasses import dataclass  # noqa: E402
from typing import Dict, List, Sequence, Tuple  # noqa: E402

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from .ddp import GradSync, make_buckets  # noqa: E402
from .masking import mask_seed  # noqa: E402
from .params import ALIGN, P, bucket_starts  # noqa: E402


@dataclass
class FinetuneConfig:
    """configs/MeerKAT/finetune_mixup_100.yaml:60-108 (model / optimization / optimizer / lr_scheduler sections)"""
    n_labels: int = 12
    apply_mask: bool = True
    mask_prob: float = 0.825
    mask_length: int = 4
    mask_channel_prob: float = 0.5
    mask_channel_length: int = 64
    zero_mask: bool = False                # False: masked frames become N(0, mask_noise_std) (wav2vec2.py:121,127)
    dropout: float = 0.1                   # -> encoder_dropout, prenet_dropout, post_mlp_drop (wav2vec2.py:116-126)
    activation_dropout: float = 0.1
    attention_dropout: float = 0.2
    dropout_input: float = 0.0
    final_dropout: float = 0.0
    layerdrop: float = 0.1                 # -> layerdrop and prenet_layerdrop
    freeze_finetune_updates: int = 10000
    layer_decay: float = 1.0               # wav2vec2.py:214-236
    # weight_decay_scale 0 (biases, 1-D, alibi, p-swish) and the layer_decay lr scales are both `optim_overrides`
    # (data2vec2.py:318-320, wav2vec2.py:230-236), which only fairseq's `composite` optimizer reads.  The shipped finetune
    # recipes train with plain `adam` (finetune_mixup_100.yaml: optimizer._name adam), which ignores BOTH: False = that.
    optim_overrides: bool = False
    target_mixup: bool = True
    source_mixup: float = 0.5
    mixup_prob: float = 1.0                # < 1: bernoulli per clip (wav2vec2.py:366-373)
    same_mixup: bool = True
    mixing_window_length: float = 0.05
    # The wrapped pretraining model keeps ITS OWN source_mixup / mixup_prob (they are not in arg_overrides,
    # wav2vec2.py:94-131), so in train() mode data2vec2.py:536-598 mixes the already mixed sources a second time
    # (targets untouched).  True reproduces that; it uses the encoder config's source_mixup.
    inner_source_mixup: bool = True
    # optimizer / schedule
    lr: float = 3e-5
    adam_betas: Tuple[float, float] = (0.9, 0.98)
    adam_eps: float = 1e-8
    weight_decay: float = 0.0
    clip_norm: float = 0.0
    warmup_updates: int = 2000
    warmup_init_lr: float = 1e-10
    min_lr: float = 5e-6
    max_update: int = 30000


def cosine_lr(cfg: FinetuneConfig, num_updates: int) -> float:
    """fairseq CosineLRSchedule (lr_period_updates <= 0 -> period = max_update - warmup, t_mult 1, lr_shrink 0.1);
    third party, restated from its published semantics."""
    if num_updates < cfg.warmup_updates:
        return cfg.warmup_init_lr + num_updates * (cfg.lr - cfg.warmup_init_lr) / cfg.warmup_updates
    period = max(1, cfg.max_update - cfg.warmup_updates)
    curr = num_updates - cfg.warmup_updates
    i = curr // period
    t_curr = curr - period * i
    shrink = 0.1 ** i
    lo, hi = cfg.min_lr * shrink, cfg.lr * shrink
    return lo + 0.5 * (hi - lo) * (1 + math.cos(math.pi * t_curr / period))


def layer_lr_scales(n_blocks: int, layer_decay: float) -> List[float]:
    """wav2vec2.py:214-224: block i (prenet blocks first, 0-based) trains with lr * layer_decay ** (n_blocks + 1 - (i + 1))."""
    num_layers = n_blocks + 1
    scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    return [scales[i + 1] for i in range(n_blocks)]


def _to_device(arr: np.ndarray, device) -> torch.Tensor:
    """host array -> device through pinned memory (a pageable copy would synchronise the host with the stream)"""
    t = torch.from_numpy(arr)
    if torch.device(device).type != "cuda":
        return t
    return t.pin_memory().to(device, non_blocking=True)


def span_lengths(fcfg, T: int, D: int):
    """The time / channel span lengths the device sampler is asked for: both laws are checked -- span length 1 selects
    another law in the reference (random_masking) and is refused, like the switches masking.check_span_law names -- and a span that
    does not fit the axis (a clip shorter than mask_length frames, mask_channel_length >= embed_dim) is clamped to axis - 1 as
    fairseq's compute_mask_indices does (`min(mask_length, sz - 1)` per row), instead of failing inside the kernel mid-training."""
    from .masking import check_span_law
    t_len, c_len = int(fcfg.mask_length), int(fcfg.mask_channel_length)
    if fcfg.apply_mask and fcfg.mask_prob > 0:
        check_span_law(mask_length=t_len)
        if T < 3:
            raise ValueError(f"time masking needs at least 3 frames (got T = {T})")
        t_len = min(t_len, T - 1)
    if fcfg.apply_mask and fcfg.mask_channel_prob > 0:
        check_span_law(mask_length=c_len)
        c_len = min(c_len, D - 1)
    return t_len, c_len


def make_aug(fcfg: FinetuneConfig, B: int, T: int, D: int, n_prenet: int, n_blocks: int, rng: np.random.Generator,
             noise_std: float, device="cuda", train_dropout: bool = True, mask_seeds=None) -> dict:
    """The per-micro-batch random draws of the train form (what Engine.finetune_forward takes as `aug`).
    The time masks (B, T) and channel masks (B, D) of apply_mask (reference nn/modalities/base.py:370-425, 457-485) come from the
    DEVICE sampler (a2v_span_mask: the law of fairseq's compute_mask_indices with require_same_masks, min_masks 1 over frames / 0
    over channels), two launches each on the current stream -- no host loop, no copy.  mask_seeds = (time, channel) 64-bit
    seeds, a pure function of (run seed, update, micro-batch, rank) at the callers (masking.mask_seed), so a resumed run
    repeats them; None draws them from `rng`."""
    aug: Dict[str, object] = {}
    if mask_seeds is None:
        mask_seeds = (int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 62)))
    t_len, c_len = span_lengths(fcfg, T, D)
    if fcfg.apply_mask and fcfg.mask_prob > 0:
        aug["mask_t"] = ops.span_mask(B, T, mask_prob=fcfg.mask_prob, mask_length=t_len, seed=mask_seeds[0],
                                      min_masks=1, device=device)[0]
        aug["noise_std"] = 0.0 if fcfg.zero_mask else noise_std
        aug["noise_seed"] = int(rng.integers(0, 2 ** 31))
    if fcfg.apply_mask and fcfg.mask_channel_prob > 0:
        aug["mask_c"] = ops.span_mask(B, D, mask_prob=fcfg.mask_channel_prob, mask_length=c_len,
                                      seed=mask_seeds[1], min_masks=0, device=device)[0]
    aug["active"] = [bool(fcfg.layerdrop == 0 or rng.random() > fcfg.layerdrop) for _ in range(n_blocks)]
    if not any(aug["active"][n_prenet:]):          # the reference divides by zero layer results here; keep one block
        aug["active"][-1] = True
    if train_dropout and max(fcfg.dropout, fcfg.activation_dropout, fcfg.attention_dropout, fcfg.dropout_input,
                             fcfg.final_dropout) > 0:
        aug.update(drop_seed=int(rng.integers(0, 2 ** 31)), p_drop=fcfg.dropout, p_act=fcfg.activation_dropout,
                   p_att=fcfg.attention_dropout, p_in=fcfg.dropout_input, p_final=fcfg.final_dropout)
    return aug


def bc_mix(mixup_op, f: "FinetuneConfig", model_cfg, src, tgt, rng: np.random.Generator, perm_gen):
    """BC mixup of sources (wav2vec2.py:362-424) and, with target_mixup, of the targets (:426-432), then the
    wrapped model's own second source mixup (see FinetuneConfig.inner_source_mixup)."""
    if mixup_op is None:
        return src, tgt
    B = src.shape[0]
    r = float(rng.uniform(max(1e-6, f.source_mixup), 1.0))
    perm = torch.rand(B, device=src.device, generator=perm_gen).argsort().to(torch.int32)
    mm = None
    if f.mixup_prob < 1:                  # bernoulli(mixup_prob) per clip (wav2vec2.py:366-373)
        mm = (torch.rand(B, device=src.device, generator=perm_gen) < f.mixup_prob).to(torch.uint8)
    src = mixup_op(src, perm, r, mm)
    if f.target_mixup and tgt is not None:
        mixed_t = tgt * r + (1 - r) * tgt[perm.long()]
        tgt = mixed_t if mm is None else torch.where(mm.bool().view(-1, 1, 1), mixed_t, tgt)
    if f.inner_source_mixup and model_cfg.source_mixup >= 0 and model_cfg.mixup_prob > 0:
        r2 = float(rng.uniform(max(1e-6, model_cfg.source_mixup), 1.0))
        perm2 = torch.rand(B, device=src.device, generator=perm_gen).argsort().to(torch.int32)
        mm2 = None
        if model_cfg.mixup_prob < 1:
            mm2 = (torch.rand(B, device=src.device, generator=perm_gen) < model_cfg.mixup_prob).to(torch.uint8)
        src = mixup_op(src, perm2, r2, mm2)
    return src, tgt


class FinetuneTrainer:
    """Data-parallel finetune step on the native engine: train-form forward/backward per micro-batch (BC mixup of
    sources and targets, masking, layerdrop, dropouts), bucketed gradient all-reduce overlapped with backward, Adam
    with per-block lr scales (layer_decay) on the encoder slice and on the head, classifier-only updates while
    num_updates < freeze_finetune_updates.  Replaces fairseq Trainer + Wav2VecCcasFinetune + FinetuneCrossEntropyCriterion
    around the hot path (wav2vec2.py:57-482, criterions.py:137-277); no control plane."""

    def __init__(self, model_cfg, fcfg: Optional[FinetuneConfig] = None, *, store=None, dtype=torch.bfloat16, device="cuda",
                 seed: int = 0, process_group=None):
        from .params import ParamStore
        self.cfg, self.fcfg = model_cfg, fcfg or FinetuneConfig()
        self.store = store if store is not None else ParamStore(model_cfg, device=device)
        if store is None:
            self.store.init_reference(seed)
        self.engine = Engine(model_cfg, self.store, dtype=dtype)
        self.head = FinetuneHead(model_cfg.embed_dim, self.fcfg.n_labels, dtype=dtype, device=self.store.device, seed=seed)
        dev = self.store.device
        self.head_m = (torch.zeros_like(self.head.w), torch.zeros_like(self.head.b))
        self.head_v = (torch.zeros_like(self.head.w), torch.zeros_like(self.head.b))
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.num_updates = 0
        self.seed = int(seed)
        self.rng = np.random.default_rng(seed * 1000 + 17 + self.rank)
        self._perm_gen = torch.Generator(device=dev)            # device-side permutations: no pageable H2D copy per micro-batch
        self._perm_gen.manual_seed(seed * 7919 + 29 + self.rank)
        self.mixup = ops.Mixup(model_cfg.audio.sample_rate, self.fcfg.mixing_window_length, dev) \
            if self.fcfg.source_mixup >= 0 and self.fcfg.mixup_prob > 0 else None
        # trainable slice of the flat buffer: transformer blocks .. project_features (the decoder is removed and the
        # conv feature extractor frozen, wav2vec2.py:176, finetune_mixup_100.yaml:85)
        by = self.store.by_name
        nl = len(model_cfg.conv_layers)
        self.lo = self.store.sh_lo
        self.hi = by[P + f"local_encoder.conv_layers.{nl - 1}.2.1.weight"].offset
        assert self.lo % ALIGN == 0 and self.hi % ALIGN == 0
        buckets = {k: (max(lo, self.lo), min(hi, self.hi)) for k, (lo, hi) in
                   make_buckets(bucket_starts(model_cfg, self.store.specs), self.store.total).items()}
        self.sync = GradSync(self.store.grad, {k: v for k, v in buckets.items() if v[1] > v[0]}, process_group)
        self.engine.on_grad_ready = self._on_ready
        self.lr_scale = self._lr_scale_table()
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.last: Dict[str, float] = {}

    def _on_ready(self, event):
        if event in self.sync.buckets:
            self.sync.on_ready(event)

    def _lr_scale_table(self):
        f = self.fcfg
        if f.layer_decay >= 1:
            return None
        a = self.cfg.audio
        n_blocks = a.prenet_depth + self.cfg.depth
        sc = layer_lr_scales(n_blocks, f.layer_decay)
        tab = np.ones((self.hi - self.lo) // ALIGN, dtype=np.float32)
        for s in self.store.specs:
            if not (self.lo <= s.offset < self.hi):
                continue
            blk = None
            if s.name.startswith(P + "context_encoder.blocks."):
                blk = int(s.name[len(P + "context_encoder.blocks."):].split(".")[0])
            elif s.name.startswith("blocks."):
                blk = a.prenet_depth + int(s.name.split(".")[1])
            if blk is not None:
                tab[(s.offset - self.lo) // ALIGN:(s.offset - self.lo + s.numel + ALIGN - 1) // ALIGN] = sc[blk]
        return torch.from_numpy(tab).to(self.store.device)

    @property
    def frozen(self) -> bool:
        """wav2vec2.py:440: ft = freeze_finetune_updates <= num_updates"""
        return self.num_updates < self.fcfg.freeze_finetune_updates

    def _mix(self, src, tgt):
        return bc_mix(self.mixup, self.fcfg, self.cfg, src, tgt, self.rng, self._perm_gen)

    def train_step(self, micro_batches: Sequence[dict]) -> None:
        """micro_batches: dicts with source (B,L) fp32 and target (B,T,C) fp32 multi-hot, on the device."""
        st, eng, f, head = self.store, self.engine, self.fcfg, self.head
        st.grad[self.lo:self.hi].zero_()
        head.zero_grad()
        frozen = self.frozen
        a = self.cfg.audio
        n_blocks = a.prenet_depth + self.cfg.depth
        ntok = 0
        nmb = len(micro_batches)
        for j, mb in enumerate(micro_batches):
            self.sync.arm(j == nmb - 1 and not frozen)
            src, tgt = self._mix(mb["source"], mb["target"])
            B, T = tgt.shape[0], tgt.shape[1]
            seeds = (mask_seed(self.seed, self.num_updates, 2 * j, self.rank), mask_seed(self.seed, self.num_updates, 2 * j + 1, self.rank))
            aug = make_aug(f, B, T, self.cfg.embed_dim, a.prenet_depth, n_blocks, self.rng, a.mask_noise_std, device=st.device,
                           mask_seeds=seeds)
            eng.finetune_forward_backward(src, tgt, head, aug=aug, head_only=frozen)
            ntok += B * T
        self.sync.arm(False)
        # ---- reduce: encoder buckets (already in flight), then the head and the scalars in one small call
        packed = torch.cat([head.gw.view(-1), head.gb.view(-1), head.loss_sum.float().view(-1),
                            torch.full((1,), float(ntok), dtype=torch.float32, device=st.device)])   # fill kernel, no H2D copy
        if self.sync.active:
            if not frozen:
                self.sync.finish(None)
            dist.all_reduce(packed, group=self.pg)
            ngw = head.gw.numel()
            head.gw.copy_(packed[:ngw].view_as(head.gw))
            head.gb.copy_(packed[ngw:ngw + head.gb.numel()])
        loss_sum, sample_size = packed[-2], packed[-1]
        inv = 1.0 / sample_size
        gw, gb = head.grad_w * inv, head.grad_b * inv
        scale = inv.reshape(1).float()
        gnorm = None
        if f.clip_norm > 0:
            self._sumsq.zero_()
            if not frozen:
                ops.sumsq(st.grad[self.lo:self.hi], self._sumsq)
            gnorm = torch.sqrt(self._sumsq[0].float() * inv * inv + (gw * gw).sum() + (gb * gb).sum())
            clip = torch.clamp(f.clip_norm / (gnorm + 1e-6), max=1.0)
            scale = scale * clip
            gw, gb = gw * clip, gb * clip
        self.num_updates += 1
        lr = cosine_lr(f, self.num_updates)
        b1, b2 = f.adam_betas
        if not frozen:
            # fairseq Adam keeps a step count per parameter that starts with its first gradient: the encoder's
            # bias correction starts when it is unfrozen
            enc_step = self.num_updates - f.freeze_finetune_updates
            ops.adam_step(st.flat[self.lo:self.hi], st.grad[self.lo:self.hi], st.adam_m[self.lo:self.hi],
                          st.adam_v[self.lo:self.hi], lr=lr, beta1=b1, beta2=b2, eps=f.adam_eps, weight_decay=f.weight_decay,
                          step=max(1, enc_step),
                          # plain fairseq `adam` ignores optim_overrides: weight decay on every parameter, no layer_decay scales
                          decay_mask=st.decay_mask[self.lo // ALIGN:self.hi // ALIGN] if f.optim_overrides else None,
                          grad_scale_dev=scale, lr_scale=self.lr_scale if f.optim_overrides else None)
            eng.refresh_student_packs()
        for prm, gr, m, v in ((head.w, gw, self.head_m[0], self.head_v[0]), (head.b, gb, self.head_m[1], self.head_v[1])):
            ops.adam_step(prm.view(-1), gr.contiguous().view(-1), m.view(-1), v.view(-1), lr=lr, beta1=b1, beta2=b2,
                          eps=f.adam_eps, weight_decay=f.weight_decay, step=self.num_updates)
        head.refresh()
        self.last = dict(loss_sum=loss_sum, sample_size=sample_size, lr=lr, gnorm=gnorm)

    def metrics(self) -> Dict[str, float]:
        """loss / sample_size as fairseq logs it (criterions.py:279-295) -- host sync"""
        n = float(self.last["sample_size"])
        return dict(loss=float(self.last["loss_sum"]) / n / math.log(2), sample_size=n, lr=self.last["lr"],
                    gnorm=float(self.last["gnorm"]) if self.last["gnorm"] is not None else None)


# ---------------------------------------------------------------------------------------------- drop-in surface
import torch.nn as tnn  # noqa: E402

from .registry import register_criterion, register_model  # noqa: E402


class _FinetuneLogits(torch.autograd.Function):
    """autograd bridge: forward runs the HIP forward (keeping its saved state in the engine), backward receives
    d loss / d logits from whatever criterion sits on top and runs the hand-written backward into the flat gradient
    buffer (encoder) and the head's gradient buffers."""

    @staticmethod
    def forward(ctx, anchor, model, source, aug=None, head_only=False):
        ctx.model, ctx.head_only = model, head_only
        eng, head = model.encoder.engine, model.head
        logits = eng.finetune_forward(source, head, save=True, aug=aug)
        B = source.shape[0]
        return logits.view(B, logits.shape[0] // B, -1)[..., : head.n_labels].float()

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        eng, head = model.encoder.engine, model.head
        B, T, C = dlogits.shape
        dl = torch.zeros(B * T, head.Cp, dtype=eng.dtype, device=dlogits.device)
        dl[:, :C] = dlogits.reshape(B * T, C).to(eng.dtype)
        if not ctx.head_only:
            model.encoder.attach_grads()                      # the optimizer may have set p.grad = None
        eng.finetune_backward(dl, head, head_only=ctx.head_only)
        model._publish_head_grads()
        return torch.zeros_like(model._anchor), None, None, None, None


@register_model("wav2vec_ccas_finetune")
class Wav2VecCcasFinetune(tnn.Module):
    """Call surface of the reference's finetune model (wav2vec2.py:57-90,430-482): forward(source, padding_mask, target)
    -> {"encoder_out": logits (B,T,C), "padding_mask", "layer_results", "target"}; get_logits / get_targets as the
    criterion uses them.  `encoder` is a Data2VecMultiModel of this package (its parameters' .grad are views of the
    flat gradient buffer); `proj.weight/.bias` mirror the head."""

    def __init__(self, encoder, n_labels: int, seed: int = 0, use_focal_loss: bool = True, fcfg: Optional["FinetuneConfig"] = None):
        """fcfg=None: deterministic form (no mixup, mask, layerdrop or dropout even in train mode; the parity tests use
        it).  With a FinetuneConfig, train mode runs the reference's train form: BC mixup of sources and targets
        (wav2vec2.py:362-432), apply_mask (mask = apply_mask and training, :433-437), layerdrop and dropouts pushed into
        the encoder (:94-131), classifier-only gradients while num_updates < freeze_finetune_updates (:440-445)."""
        super().__init__()
        self.encoder = encoder
        self.cfg_use_focal_loss = use_focal_loss
        self.fcfg = fcfg
        self.num_updates = 0
        cfg = encoder.cfg
        dev = encoder.store.device
        self.head = FinetuneHead(cfg.embed_dim, n_labels, dtype=encoder.engine.dtype, device=dev, seed=seed)
        self.proj = tnn.Linear(cfg.embed_dim, n_labels).to(dev)
        with torch.no_grad():
            self.proj.weight.copy_(self.head.w)
            self.proj.bias.copy_(self.head.b)
        self._anchor = tnn.Parameter(torch.zeros(1, device=dev))
        self._rng = np.random.default_rng(seed * 1000 + 23)
        self._perm_gen = None
        self._mixup_op = None
        if fcfg is not None and fcfg.source_mixup >= 0 and fcfg.mixup_prob > 0 and torch.device(dev).type == "cuda":
            self._mixup_op = ops.Mixup(cfg.audio.sample_rate, fcfg.mixing_window_length, dev)

    def set_num_updates(self, num_updates):
        """wav2vec2.py:84-87 (fairseq calls it after every optimizer step): also the moment to rebuild the operand
        images of whatever the optimizer just changed."""
        self.num_updates = num_updates
        self._aug_draws = 0
        self.sync_head()
        self.encoder.engine.refresh_student_packs()

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
            aug, head_only, f = None, False, self.fcfg
            if f is not None:
                cfg = self.encoder.cfg
                a = cfg.audio
                if self._perm_gen is None:
                    self._perm_gen = torch.Generator(device=source.device)
                    self._perm_gen.manual_seed(cfg.seed * 7919 + 43)
                tgt = target.to(source.device, torch.float32) if target is not None else None
                source, target = bc_mix(self._mixup_op, f, cfg, source, tgt, self._rng, self._perm_gen)
                T = cfg.frame_counts(source.shape[1])[-1]
                self._aug_draws = getattr(self, "_aug_draws", 0) + 1            # micro-batches since the last set_num_updates
                rank = dist.get_rank() if dist.is_initialized() else 0         # (resolved at the draw: the group may be set up late)
                seeds = (mask_seed(cfg.seed, self.num_updates, 2 * self._aug_draws, rank),
                         mask_seed(cfg.seed, self.num_updates, 2 * self._aug_draws + 1, rank))
                aug = make_aug(f, source.shape[0], T, cfg.embed_dim, a.prenet_depth, a.prenet_depth + cfg.depth, self._rng,
                               a.mask_noise_std, device=source.device, mask_seeds=seeds)
                head_only = self.num_updates < f.freeze_finetune_updates
            logits = _FinetuneLogits.apply(self._anchor, self, source, aug, head_only)
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


def label_smoothed_nll_loss(lprobs, target, epsilon: float, ignore_index=None, reduce=True):
    """fairseq.criterions.label_smoothed_cross_entropy.label_smoothed_nll_loss (third party, restated): lprobs (M, C)
    log-probabilities, target (M,) class indices; eps_i = eps / (C - 1); loss = (1 - eps - eps_i) nll + eps_i smooth."""
    if target.dim() == lprobs.dim() - 1:
        target = target.unsqueeze(-1)
    nll = -lprobs.gather(dim=-1, index=target)
    smooth = -lprobs.sum(dim=-1, keepdim=True)
    if ignore_index is not None:
        pad = target.eq(ignore_index)
        nll = nll.masked_fill(pad, 0.0)
        smooth = smooth.masked_fill(pad, 0.0)
    else:
        nll, smooth = nll.squeeze(-1), smooth.squeeze(-1)
    if reduce:
        nll, smooth = nll.sum(), smooth.sum()
    eps_i = epsilon / (lprobs.size(-1) - 1)
    return (1.0 - epsilon - eps_i) * nll + eps_i * smooth, nll


def confusion(preds, target):
    """tp, fp, tn, fn counts of binary predictions against binary targets (nn/utils.py confusion, as logged by
    criterions.py:219-231)"""
    p, t = preds.bool(), target.bool()
    return (int((p & t).sum()), int((p & ~t).sum()), int((~p & ~t).sum()), int((~p & t).sum()))


@register_criterion("finetunecriterion")
class FinetuneCrossEntropyCriterion:
    """The reference criterion (criterions.py:137-277) on the autograd surface: with use_focal_loss (every shipped
    finetune config) loss = sum of sigmoid_focal_loss(logits, multi-hot target); otherwise fairseq's label-smoothed
    cross entropy over single-label targets.  sample_size = ntokens (sentence_avg false).  With report_accuracy the
    logging output carries accuracy and confusion counts at metric_threshold (segment-level scores are evaluation
    tooling and not rebuilt)."""

    def __init__(self, task=None, use_focal_loss=True, sentence_avg=False, metric_threshold=0.5, report_accuracy=False,
                 label_smoothing=0.0, padding_idx=None):
        self.task, self.sentence_avg, self.metric_threshold, self.report_accuracy = task, sentence_avg, metric_threshold, report_accuracy
        self.use_focal_loss, self.eps, self.padding_idx = use_focal_loss, label_smoothing, padding_idx

    def __call__(self, model, sample, reduce=True):
        return self.forward(model, sample, reduce)

    def forward(self, model, sample, reduce=True):
        net_output = model(**sample["net_input"])
        logits = model.get_logits(net_output)
        target = model.get_targets(sample, net_output).to(logits.device)
        if self.use_focal_loss:
            loss = sigmoid_focal_loss(logits, target, reduction="sum" if reduce else "none")
            nll_loss = torch.tensor(0.0)
        else:
            lprobs = torch.log_softmax(logits.float(), dim=-1)
            loss, nll_loss = label_smoothed_nll_loss(lprobs, target.reshape(-1).long(), self.eps, ignore_index=self.padding_idx,
                                                     reduce=reduce)
        sample_size = sample["target"].size(0) if self.sentence_avg else sample.get("ntokens", target.shape[0])
        logging_output = {"loss": loss.data, "nll_loss": nll_loss.data, "ntokens": sample.get("ntokens", target.shape[0]),
                          "nsentences": sample["target"].size(0), "sample_size": sample_size}
        if self.report_accuracy:
            with torch.no_grad():
                if self.use_focal_loss:
                    preds = (torch.sigmoid(logits) >= self.metric_threshold)
                    tgt = target >= 0.5
                    logging_output["finetune/n_correct"] = float((preds == tgt).sum())
                    logging_output["finetune/total"] = float(target.numel())
                else:
                    t1 = target.reshape(-1).long()
                    keep = t1.ne(self.padding_idx) if self.padding_idx is not None else torch.ones_like(t1, dtype=torch.bool)
                    logging_output["finetune/n_correct"] = float((logits.argmax(-1)[keep] == t1[keep]).sum())
                    logging_output["finetune/total"] = float(keep.sum())
                    preds = torch.softmax(logits.float(), -1) >= self.metric_threshold
                    tgt = torch.nn.functional.one_hot(t1, logits.size(-1)).bool()
                tp, fp, tn, fn = confusion(preds, tgt)
                logging_output.update({"finetune/tp": tp, "finetune/fp": fp, "finetune/tn": tn, "finetune/fn": fn})
        return loss, sample_size, logging_output

    @classmethod
    def reduce_metrics(cls, logging_outputs) -> None:
        """criterions.py:279-380: loss in bits per sample, nll / perplexity, every other logged key as the mean over the
        logging outputs, and -- when accuracy counts were logged -- the derived accuracy / precision / recall / F1."""
        from . import metrics
        from .criterion import _log_prf
        item = lambda v: float(v.item()) if hasattr(v, "item") else float(v)
        tot = lambda key: item(sum(log.get(key, 0) for log in logging_outputs))
        loss_sum, nll_sum, ntokens, sample_size, nsentences = tot("loss"), tot("nll_loss"), tot("ntokens"), tot("sample_size"), tot("nsentences")
        metrics.log_scalar("misc/ntokens", ntokens)
        metrics.log_scalar("misc/nsentences", nsentences)
        metrics.log_scalar("misc/sample_size", sample_size)
        metrics.log_scalar("loss", loss_sum / sample_size / math.log(2), sample_size, round=3)
        metrics.log_scalar("misc/nll_loss", nll_sum / ntokens / math.log(2), ntokens, round=3)
        metrics.log_derived("misc/perplexity", lambda m: metrics.safe_round(2 ** min(m["misc/nll_loss"].avg, 100.0), 2))
        metrics.log_scalar("finetune/total", tot("count"))
        builtin = {"loss", "nll_loss", "ntokens", "nsentences", "sample_size", "total"}
        for k in logging_outputs[0]:
            if k in builtin or k.startswith("_"):
                continue
            val = tot(k)
            if k.startswith("loss"):
                metrics.log_scalar("individual_losses/" + k, val / (sample_size or 1) / math.log(2), sample_size, round=3)
            else:
                metrics.log_scalar(k, val / len(logging_outputs), round=3)
        if tot("finetune/total") > 0:
            _log_prf(metrics, "finetune")

    def logging_outputs_can_be_summed(self) -> bool:
        return True
