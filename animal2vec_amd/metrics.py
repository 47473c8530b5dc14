"""Meter table behind `reduce_metrics` (the reference logs through fairseq.logging.metrics: log_scalar / log_speed /
log_derived, criterions.py:279-380,413-502, audio_tasks.py:154-188).  If fairseq is importable the calls are forwarded
to it, so the numbers reach the reference's progress bar / tensorboard; otherwise they land in the small table below,
which keeps the same aggregation semantics (weighted average per key, sums for derived metrics)."""
from __future__ import annotations

import time
from typing import Callable, Dict


class _Meter:
    def __init__(self, round_=None, priority=10):
        self.sum, self.count, self.val, self.round, self.priority = 0.0, 0.0, 0.0, round_, priority

    def update(self, val, n=1):
        self.val = float(val)
        if n > 0:
            self.sum += float(val) * n
            self.count += n

    @property
    def avg(self):
        return self.sum / self.count if self.count > 0 else self.val

    @property
    def smoothed_value(self):
        v = self.avg
        return round(v, self.round) if self.round is not None else v


class _Speed(_Meter):
    def __init__(self, round_=None, priority=10):
        super().__init__(round_, priority)
        self.start = time.perf_counter()

    @property
    def avg(self):
        dt = time.perf_counter() - self.start
        return self.sum / dt if dt > 0 else 0.0


_METERS: Dict[str, _Meter] = {}
_DERIVED: Dict[str, Callable] = {}


def _fairseq():
    try:
        from fairseq.logging import metrics as m            # third party, absent in this image
        return m
    except Exception:
        return None


def reset():
    _METERS.clear()
    _DERIVED.clear()


def log_scalar(key, value, weight=1, priority=10, round=None):
    f = _fairseq()
    if f is not None:
        return f.log_scalar(key, value, weight, priority=priority, round=round)
    m = _METERS.get(key)
    if m is None:
        m = _METERS[key] = _Meter(round, priority)
    m.update(float(value), weight)


def log_speed(key, value, priority=30, round=None):
    f = _fairseq()
    if f is not None:
        return f.log_speed(key, value, priority=priority, round=round)
    m = _METERS.get(key)
    if not isinstance(m, _Speed):
        m = _METERS[key] = _Speed(round, priority)
    m.update(float(value), 1)


def log_derived(key, fn, priority=20):
    f = _fairseq()
    if f is not None:
        return f.log_derived(key, fn, priority)
    _DERIVED[key] = fn


def get_smoothed_values() -> Dict[str, float]:
    out = {k: m.smoothed_value for k, m in sorted(_METERS.items(), key=lambda kv: (kv[1].priority, kv[0]))}
    for k, fn in _DERIVED.items():
        out[k] = fn(_METERS)
    return out


def safe_round(number, ndigits):
    if hasattr(number, "item"):
        number = number.item()
    return round(float(number), ndigits)
