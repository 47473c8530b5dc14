"""fairseq-compatible registration surface (reference: @register_model("data2vec_multi") data2vec2.py:168,
@register_task("audio_ccas") audio_tasks.py:92, @register_criterion("expanded_model") criterions.py:388).
If fairseq is importable the decorators forward to it, so the classes also appear in fairseq's own
registries and `animal2vec_train.py` keeps working under the reference's trainer; otherwise the local
tables below are the registry."""
from __future__ import annotations

MODEL_REGISTRY, TASK_REGISTRY, CRITERION_REGISTRY = {}, {}, {}
MODEL_DATACLASS, TASK_DATACLASS, CRITERION_DATACLASS = {}, {}, {}


def _fairseq(kind):
    try:
        if kind == "model":
            from fairseq.models import register_model as f
        elif kind == "task":
            from fairseq.tasks import register_task as f
        else:
            from fairseq.criterions import register_criterion as f
        return f
    except Exception:
        return None


def _make(kind, table, dc_table):
    def register(name, dataclass=None):
        def deco(cls):
            if name in table and table[name] is not cls:
                raise ValueError(f"Cannot register duplicate {kind} ({name})")
            table[name] = cls
            dc_table[name] = dataclass
            f = _fairseq(kind)
            if f is not None:
                try:
                    return f(name, dataclass=dataclass)(cls)
                except Exception:
                    return cls
            return cls
        return deco
    return register


register_model = _make("model", MODEL_REGISTRY, MODEL_DATACLASS)
register_task = _make("task", TASK_REGISTRY, TASK_DATACLASS)
register_criterion = _make("criterion", CRITERION_REGISTRY, CRITERION_DATACLASS)
