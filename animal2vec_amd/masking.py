"""Mask bookkeeping of the reference (nn/modalities/base.py:370-455) for the pretraining path."""
from __future__ import annotations

import numpy as np
import torch


def make_maskinfo(mask: torch.Tensor):
    """mask (N,T), 1 = masked, equal count per row (the reference assumes it, base.py:437).
    Returns (mask_u8, ids_keep (N,K) int32, ids_restore (N,T) int32).  Ties of the reference's
    (unstable) argsort are resolved in ascending frame order; decoder mask-token noise row j belongs
    to the j-th masked frame in ascending order."""
    m = mask.to(torch.uint8).contiguous()
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
