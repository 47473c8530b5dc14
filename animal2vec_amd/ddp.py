"""Gradient synchronisation for one-process-per-GPU data parallelism (torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).  The flat gradient buffer is ordered by backward-completion
(params.py), so each engine event releases one contiguous bucket whose all-reduce runs on the
collective stream while backward keeps issuing kernels: xGMI is point-to-point (7 links x ~153 GB/s per
GPU), a 1.26 GB fp32 gradient needs ~10 ms of ring time per step, which hides behind >100 ms of backward.
Replaces fairseq's LegacyDistributedDataParallel (single flat all-reduce AFTER backward, no overlap) and
the six small all-reduces of compute_var (reference nn/data2vec2.py:1098-1108) with one packed call.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def make_buckets(starts: Sequence[Tuple[str, int]], total: int) -> Dict[str, Tuple[int, int]]:
    """starts: (event, first offset) in flat order -> event -> [lo, hi) covering [0,total) exactly once."""
    out = {}
    for j, (ev, lo) in enumerate(starts):
        hi = starts[j + 1][1] if j + 1 < len(starts) else total
        assert 0 <= lo <= hi <= total, (ev, lo, hi)
        out[ev] = (lo, hi)
    return out


class GradSync:
    def __init__(self, grad: torch.Tensor, buckets: Dict[str, Tuple[int, int]], process_group=None, wire_dtype=None):
        """wire_dtype (None = the gradient's own fp32): the dtype the buckets travel in.  torch.bfloat16 halves the bytes on the
        links (632 MB instead of 1.26 GB for the large model) -- what the reference's fp16 recipe does with its fp16 gradients
        (fairseq FP16Optimizer: the all-reduce runs on the half-precision copies) -- at the price of two conversion passes over
        the flat buffer on the compute stream; accumulation over micro-batches stays fp32 in the flat buffer either way."""
        self.grad, self.buckets, self.pg = grad, buckets, process_group
        self.wire = None if wire_dtype in (None, grad.dtype) else wire_dtype
        self._stage = None              # the wire-dtype image of the flat gradient, allocated at the first reduction
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if inited else 1
        # A2V_DDP_FORCE=1: issue the collectives even at world size 1 (exercises the RCCL path on a one-GPU box)
        self.active = inited and (self.world > 1 or os.environ.get("A2V_DDP_FORCE") == "1")
        self.handles: List = []
        self.armed = False          # only the LAST micro-batch of an accumulation group reduces
        self.fired = set()

    def arm(self, on: bool):
        self.armed = on
        if on:
            self.fired = set()
            self.handles = []

    def on_ready(self, event: str):
        if not self.armed or not self.active:
            return
        lo, hi = self.buckets[event]
        assert event not in self.fired, f"bucket {event} released twice"
        self.fired.add(event)
        if hi > lo:
            self.handles.append(self._reduce(lo, hi))

    def _reduce(self, lo: int, hi: int):
        """-> (work handle, slice to convert back or None)"""
        if self.wire is None:
            return dist.all_reduce(self.grad[lo:hi], group=self.pg, async_op=True), None
        if self._stage is None:
            self._stage = torch.empty(self.grad.numel(), dtype=self.wire, device=self.grad.device)
        w = self._stage[lo:hi]
        w.copy_(self.grad[lo:hi])             # (current stream; the collective is ordered behind it)
        return dist.all_reduce(w, group=self.pg, async_op=True), (lo, hi)

    def finish(self, stats: Optional[torch.Tensor] = None):
        """Wait for every bucket; reduce the packed loss/count/variance sums in one call."""
        if not self.active:
            return
        missing = set(self.buckets) - self.fired
        for ev in missing:                       # never leave a slice un-reduced
            lo, hi = self.buckets[ev]
            if hi > lo:
                self.handles.append(self._reduce(lo, hi))
        for h, back in self.handles:
            h.wait()
            if back is not None:
                self.grad[back[0]:back[1]].copy_(self._stage[back[0]:back[1]])
        self.handles = []
        if stats is not None:
            dist.all_reduce(stats, group=self.pg)


def grad_scale_factor(sumsq: torch.Tensor, sample_size: torch.Tensor, clip_norm: float):
    """fairseq Trainer semantics: grads *= 1/sample_size, then clip the global norm to clip_norm.
    Returns (scale to apply to the raw summed grads, grad norm after the 1/sample_size scaling) as
    tensors on the inputs' device (no host sync)."""
    inv = (1.0 / sample_size).to(torch.float32)
    gnorm = torch.sqrt(sumsq).to(torch.float32) * inv
    if clip_norm > 0:
        clip = torch.clamp(clip_norm / (gnorm + 1e-6), max=1.0)
    else:
        clip = torch.ones_like(gnorm)
    return inv * clip, gnorm
