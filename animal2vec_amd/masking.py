"""Mask sampling and bookkeeping of the reference (nn/modalities/base.py:370-455) for the pretraining path.

In the step the masks come from the DEVICE sampler (`DeviceMaskSampler` -> a2v_span_mask, csrc/a2v_mask.hip): one launch
pair per micro-batch on a side stream, one step ahead of its use, K read back through pinned memory -- no host loop, no
pageable copy, no argsort.  The finetune path's time / channel masks (finetune.make_aug, model._features_only) are drawn by
the same kernel (ops.span_mask; round 4).  `compute_mask_indices` below is the host statement of the same law, kept for the
CPU tests of the host logic only -- no step path calls it; `make_maskinfo` serves caller-supplied masks (`precomputed_mask`)."""
from __future__ import annotations

from collections import deque
from typing import Deque, Optional, Tuple

import numpy as np
import torch


def make_maskinfo(mask: torch.Tensor):
    """mask (N,T), 1 = masked, equal count per row (the reference assumes it, base.py:437).
    Returns (mask_u8, ids_keep (N,K) int32, ids_restore (N,T) int32).  Ties of the reference's
    (unstable) argsort are resolved in ascending frame order; decoder mask-token noise row j belongs
    to the j-th masked frame in ascending order.  Device masks go through a2v_maskinfo (one launch,
    no sort); host masks (tests of the host logic) through two stable argsorts."""
    m = mask.to(torch.uint8).contiguous()
    if m.is_cuda:
        from . import ops
        K = m.shape[1] - int(m[0].sum())
        bad = torch.zeros(1, dtype=torch.int32, device=m.device)
        keep, restore = ops.maskinfo(m, K, rows_bad=bad)
        if int(bad):
            raise ValueError(f"{int(bad)} mask rows do not keep {K} frames like row 0 (the reference requires equal counts, base.py:437)")
        return m, keep, restore
    ids_shuffle = torch.argsort(m, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    len_keep = m.shape[1] - int(m[0].sum())
    return m, ids_shuffle[:, :len_keep].to(torch.int32).contiguous(), ids_restore.to(torch.int32).contiguous()


def compute_mask_indices(shape, mask_prob: float, mask_length: int, rng: np.random.Generator,
                         min_masks: int = 1, require_same_masks: bool = True, mask_dropout: float = 0.0) -> np.ndarray:
    """Span masking with the distribution of fairseq's compute_mask_indices (static lengths, no
    overlap handling, equalised to the smallest masked count across rows) as called at base.py:401-413.
    The RNG stream of fairseq@920a548 itself is not reproducible here (parity unpinned);
    parity tests always inject ``precomputed_mask``."""
    bsz, all_sz = shape
    mask = np.zeros((bsz, all_sz), dtype=bool)
    idcs = []
    for _ in range(bsz):
        num_mask = int(mask_prob * all_sz / float(mask_length) + rng.random())
        num_mask = max(min_masks, num_mask)
        min_len = mask_length
        if all_sz - min_len <= num_mask:
            min_len = all_sz - num_mask - 1
        starts = rng.choice(all_sz - min_len, num_mask, replace=False)
        idx = np.asarray([s + off for s in starts for off in range(mask_length)])
        idcs.append(np.unique(idx[idx < all_sz]))
    target = min(len(m) for m in idcs)
    for i, idx in enumerate(idcs):
        if len(idx) > target and require_same_masks:
            idx = rng.choice(idx, target, replace=False)
        if mask_dropout > 0:
            drop = int(np.rint(len(idx) * mask_dropout))
            idx = rng.choice(idx, len(idx) - drop, replace=False)
        mask[i, idx] = True
    return mask


def seeded_keep_mask(n_rows: int, T: int, keep: int, seed: int, device="cpu") -> torch.Tensor:
    """Deterministic benchmark/test mask: per row a seeded randperm keeps `keep` frames."""
    g = torch.Generator().manual_seed(seed)
    mask = torch.ones(n_rows, T, dtype=torch.uint8)
    for r in range(n_rows):
        mask[r, torch.randperm(T, generator=g)[:keep]] = 0
    return mask.to(device)


def mask_seed(base_seed: int, update: int, micro_batch: int, rank: int) -> int:
    """64-bit seed of one micro-batch's span masks: a pure function of the run seed and the step counters, so a resumed
    run -- on any rank -- draws what the uninterrupted one would have (no RNG state to checkpoint)."""
    x = (int(base_seed) * 0x9E3779B97F4A7C15 + int(update) * 0xD1B54A32D192ED03 + int(micro_batch) * 0x8CB92BA72F3D8DD7
         + int(rank) * 0xAEF17502108EF2D9 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    return (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF


def check_span_law(*, mask_length: int, add_masks: bool = False, **other) -> None:
    """The device sampler implements ONE law: static spans of length >= 2, require_same_masks (rows cut down to the batch's smallest
    masked count), optional mask_dropout -- what every shipped recipe uses.  The reference switches to other laws for
    mask_length == 1 (random_masking, base.py:395-396) and add_masks (rows topped up instead of cut down, base.py:409), and
    has inverse_mask / keep_masked_pct / mask_prob_min / mask_prob_adjust on top: those are refused here instead of being
    sampled with the wrong law silently (there is no host path in the step any more)."""
    if add_masks:
        raise NotImplementedError("add_masks=True (base.py:409) is not implemented by the device span-mask sampler")
    if int(mask_length) == 1:
        raise NotImplementedError("mask_length == 1 selects random_masking in the reference (base.py:395-396); the device sampler "
                                  "implements the span law only")
    for k in ("inverse_mask", "keep_masked_pct", "mask_prob_min", "mask_prob_adjust"):
        if other.get(k):
            raise NotImplementedError(f"{k} is not implemented by the device span-mask sampler")


class DeviceMaskSampler:
    """Span masks + maskinfo on the device, one request ahead of their use.

    `request(N, T, seed)` launches a2v_span_mask on the sampler's own stream into a free slot and copies K
    (data dependent: T - the batch's smallest masked count) to pinned host memory; `take(N, T, seed)` returns that
    request's tensors -- or, when nothing matching was requested, launches now and waits; `end_step()` says that every
    consumer of the slots taken so far has been enqueued on the caller's stream, which makes them reusable.  The
    caller's stream waits for the sampler (event); a reused slot's launch waits for what the caller's stream held at
    request time, so a slot is never overwritten under a kernel that still reads it."""

    def __init__(self, device, *, mask_prob: float, mask_length: int, mask_dropout: float = 0.0, min_masks: int = 1,
                 add_masks: bool = False):
        check_span_law(mask_length=mask_length, add_masks=add_masks)
        self.device = torch.device(device)
        self.law = dict(mask_prob=float(mask_prob), mask_length=int(mask_length), mask_dropout=float(mask_dropout),
                        min_masks=int(min_masks))
        self.stream = torch.cuda.Stream(self.device)
        self._free = {}                   # (N, T) -> list of slots
        self._taken = []
        self._pending: Deque[Tuple[tuple, dict]] = deque()
        self.last_k: Optional[int] = None
        self.k_sum, self.k_count = 0, 0

    def _slot(self, N: int, T: int) -> dict:
        pool = self._free.setdefault((N, T), [])
        if pool:
            return pool.pop()
        dev = self.device
        return dict(bufs=(torch.empty(N, T, dtype=torch.uint8, device=dev), torch.empty(N * T, dtype=torch.int32, device=dev),
                          torch.empty(N, T, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev),
                          torch.empty(N, dtype=torch.int32, device=dev)),
                    khost=torch.empty(1, dtype=torch.int32).pin_memory(), done=torch.cuda.Event(), N=N, T=T)

    def request(self, N: int, T: int, seed: int) -> None:
        from . import ops
        slot = self._slot(N, T)
        main = torch.cuda.current_stream(self.device)
        guard = torch.cuda.Event()
        guard.record(main)                # what is enqueued so far may still read this slot's previous contents
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(guard)
            ops.span_mask(N, T, seed=seed, bufs=slot["bufs"], **self.law)
            slot["khost"].copy_(slot["bufs"][3], non_blocking=True)
            slot["done"].record(self.stream)
        self._pending.append(((N, T, int(seed)), slot))

    def take(self, N: int, T: int, seed: int):
        """-> (mask_u8 (N,T), ids_keep (N,K) int32, ids_restore (N,T) int32); K is on the host afterwards (`last_k`)."""
        key = (N, T, int(seed))
        slot = None
        while self._pending:
            k, sl = self._pending.popleft()
            if k == key:
                slot = sl
                break
            self._free.setdefault((sl["N"], sl["T"]), []).append(sl)      # a stale request (shape / counters changed)
        if slot is None:
            self.request(N, T, seed)
            _, slot = self._pending.pop()
        slot["done"].synchronize()        # host: K has landed (normally the request ran a step ago)
        torch.cuda.current_stream(self.device).wait_event(slot["done"])
        K = int(slot["khost"][0])
        self.last_k = K
        self.k_sum += K
        self.k_count += 1
        self._taken.append(slot)
        mask, keep, restore = slot["bufs"][0], slot["bufs"][1], slot["bufs"][2]
        return mask, keep[:N * K].view(N, K), restore

    def end_step(self) -> None:
        for sl in self._taken:
            self._free.setdefault((sl["N"], sl["T"]), []).append(sl)
        self._taken = []
