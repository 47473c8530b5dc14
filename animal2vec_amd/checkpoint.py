"""Checkpoint wire format: fairseq-style ``.pt`` files as the reference writes and reads them --
``{"cfg": ..., "model": {<reference parameter names>: tensor, ..., "_ema": {...}}, "optimizer_history": [...],
"extra_state": {...}, "last_optimizer_state": {...}}`` (fairseq checkpoint_utils.save_state; model part:
nn/data2vec2.py:412-429, finetune loading: nn/wav2vec2.py:132-148,340-358, alibi upgrade: nn/modalities/base.py:152-157).

The files are ordinary torch pickles.  Published checkpoints carry omegaconf / argparse objects in "cfg" / "args";
omegaconf is not part of this image, so ``load_checkpoint`` unpickles with a lenient class resolver that turns
classes of missing modules into inert containers: tensors and plain containers load exactly, foreign config
objects come back as attribute bags that ``config_to_dict`` flattens.
"""
from __future__ import annotations

import io
import pickle
from typing import Dict, Optional

import torch

FT_PREFIX = "w2v_encoder.w2v_model."          # Wav2VecCcasFinetune.w2v_encoder (Wav2VecEncoderModOut) .w2v_model
FT_HEAD = "w2v_encoder.proj."


class _Inert:
    """Stand-in for an instance of a class whose module is not importable here."""

    def __init__(self, *a, **k):
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self._state = state

    def __reduce_ex__(self, protocol):          # never re-pickled as the foreign class
        return (dict, (config_to_dict(self),))


class _LenientUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (_Inert,), {"__module__": "a2v_inert." + module})


class _LenientPickle:
    """pickle_module for torch.load"""
    __name__ = "a2v_lenient_pickle"
    Unpickler = _LenientUnpickler
    load = staticmethod(lambda f, **kw: _LenientUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _LenientUnpickler(io.BytesIO(b), **kw).load())
    dump, dumps, Pickler = pickle.dump, pickle.dumps, pickle.Pickler
    PickleError, UnpicklingError, HIGHEST_PROTOCOL = pickle.PickleError, pickle.UnpicklingError, pickle.HIGHEST_PROTOCOL


def config_to_dict(obj):
    """Best-effort plain-container view of whatever sat under "cfg" / "args"."""
    if isinstance(obj, dict):
        return {k: config_to_dict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(config_to_dict(v) for v in obj)
    if isinstance(obj, _Inert) or hasattr(obj, "__dict__") and not isinstance(obj, (torch.Tensor, type)):
        d = {k: config_to_dict(v) for k, v in vars(obj).items() if not k.startswith("__")}
        inner = d.get("_content", d.get("_val"))          # omegaconf nodes keep their payload there
        return config_to_dict(inner) if inner is not None and set(d) <= {"_content", "_val", "_metadata", "_parent", "_flags_cache", "_args", "_kwargs"} else d
    return obj


def load_checkpoint(path, map_location="cpu") -> dict:
    """fairseq checkpoint_utils.load_checkpoint_to_cpu without fairseq: returns the state dict of the file."""
    state = torch.load(path, map_location=map_location, pickle_module=_LenientPickle, weights_only=False)
    if not isinstance(state, dict) or "model" not in state:
        raise ValueError(f"{path}: not a fairseq-style checkpoint (no 'model' entry)")
    return state


def save_checkpoint(path, store, *, cfg: Optional[dict] = None, num_updates: int = 0, optimizer_state: Optional[dict] = None,
                    extra_state: Optional[dict] = None, head=None) -> dict:
    """Write the pretraining model (or, with `head`, the finetuned classifier model) in the reference's layout.
    cfg: plain nested dict ({"model": {...}, "task": {...}, ...}); tensors are written on the CPU."""
    if head is None:
        model = pretrain_state_dict(store)
    else:
        model = finetune_state_dict(store, head)
    state = {
        "cfg": cfg or {},
        "args": None,
        "model": model,
        "optimizer_history": [{"criterion_name": "FinetuneCrossEntropyCriterion" if head is not None else "ExpandedModelCriterion",
                               "optimizer_name": "FairseqAdam", "lr_scheduler_state": {"best": None},
                               "num_updates": int(num_updates)}],
        "extra_state": dict(extra_state or {}),
        "last_optimizer_state": optimizer_state,
    }
    torch.save(state, path)
    return state


def _cpu(t):
    return t.detach().to("cpu").clone()


def pretrain_state_dict(store) -> Dict[str, object]:
    """data2vec2.py:412-418: module parameters under their reference names + "_ema" (fp32 teacher parameters)."""
    sd = store.state_dict()
    out = {k: _cpu(v) for k, v in sd.items() if k != "_ema"}
    out["_ema"] = {k: _cpu(v) for k, v in sd["_ema"].items()}
    return out


def load_pretrained(store, state: dict, *, strict: bool = True) -> None:
    """Resume / evaluate a pretraining checkpoint: student parameters and the EMA teacher."""
    model = dict(state["model"] if "model" in state else state)
    ema = model.pop("_ema", None)
    store.load_state_dict(model, ema, strict=strict)


def load_for_finetune(store, state: dict, *, load_ema: bool = False) -> None:
    """Wav2VecEncoderModOut.__init__ / load_model_weights (wav2vec2.py:185-192,340-358): "_ema" is dropped -- or, with
    load_ema, the teacher's parameters replace the student's ("encoder." + k is how the reference addresses them; the
    multi-modal model keeps plain names) -- and the pretraining-only modules (decoder) are not needed."""
    model = dict(state["model"] if "model" in state else state)
    ema = model.pop("_ema", None)
    for k in ("target_proj", "decoder"):
        model.pop(k, None)
    if load_ema:
        if ema is None:
            raise KeyError("load_ema: the checkpoint has no '_ema' entry")
        for k, v in ema.items():
            model[k] = v
    need = [s.name for s in store.specs if ".decoder." not in s.name]
    missing = [k for k in need if k not in model]
    if missing:
        raise KeyError(f"missing parameters: {missing[:5]}...")
    store.load_state_dict(model, None, strict=False)


def finetune_state_dict(store, head) -> Dict[str, torch.Tensor]:
    """State dict of Wav2VecCcasFinetune (wav2vec2.py:57-66): the encoder without its pretraining modules under
    "w2v_encoder.w2v_model.", the classifier under "w2v_encoder.proj."."""
    out = {}
    for s in sorted(store.specs, key=lambda s_: s_.name):
        if ".decoder." in s.name:
            continue
        out[FT_PREFIX + s.name] = _cpu(store.p[s.name])
    out[FT_HEAD + "weight"] = _cpu(head.w)
    out[FT_HEAD + "bias"] = _cpu(head.b)
    return out


def load_finetuned(store, head, state: dict) -> None:
    model = state["model"] if "model" in state else state
    enc = {k[len(FT_PREFIX):]: v for k, v in model.items() if k.startswith(FT_PREFIX)}
    need = [s.name for s in store.specs if ".decoder." not in s.name]
    missing = [k for k in need if k not in enc]
    if missing:
        raise KeyError(f"missing parameters: {missing[:5]}...")
    store.load_state_dict(enc, None, strict=False)
    head.load(model[FT_HEAD + "weight"].to(head.w.device), model[FT_HEAD + "bias"].to(head.b.device))
