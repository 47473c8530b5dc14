"""Container-only import shims for the *live* reference (``/root/reference``).

TEST INFRASTRUCTURE -- never imported by the product (``animal2vec_amd``).

The reference's ``nn`` package imports fairseq / timm / omegaconf / hydra / tensorflow /
skimage / intervaltree / h5py / soundfile, none of which exist in this image.  This module
installs a ``sys.meta_path`` finder that fabricates empty package-like modules for those
top-level names, then overwrites the handful of names that carry arithmetic with small
stand-ins that restate the published upstream semantics (fairseq @ 920a548, timm 0.6.12).
It is used only by ``oracle/gen_golden.py`` to (a) validate ``oracle/a2v_oracle.py`` against the
reference's own files and (b) dump the golden vectors under ``tests/golden/``.

Nothing here travels to the GPU box in a way that matters: ``/root/reference`` does not exist
there, and no test marked ``gpu`` imports this file.

Third-party pieces restated here are "parity unpinned" in the sense of SURVEY.md section 8c: the
reference has no tests that pin them, and their source is not under /root/reference.
"""
from __future__ import annotations

import abc
import contextlib
import enum
import importlib.abc
import importlib.machinery
import sys
import types
from dataclasses import dataclass

import torch
import torch.nn as tnn
import torch.nn.functional as F

_STUB_ROOTS = {
    "fairseq", "omegaconf", "hydra", "timm", "tensorflow", "skimage", "intervaltree",
    "h5py", "soundfile", "iopath", "librosa",
}


def _permissive(*a, **k):
    """Stands in for any third-party callable: usable as ``f(x)``, ``@f`` and ``@f(...)``."""
    if len(a) == 1 and not k and (isinstance(a[0], type) or callable(a[0])):
        return a[0]
    return _permissive


class _DummyMeta(abc.ABCMeta):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _permissive


class _StubModule(types.ModuleType):
    """A module on which every unknown attribute is a fresh dummy class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = _DummyMeta(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# ----------------------------------------------------------------------------- stand-ins
class TransposeLast(tnn.Module):
    # fairseq.modules.TransposeLast (note upstream's misspelt kwarg ``tranpose_dim``)
    def __init__(self, deconstruct_idx=None, tranpose_dim=-2):
        super().__init__()
        self.deconstruct_idx = deconstruct_idx
        self.tranpose_dim = tranpose_dim

    def forward(self, x):
        if self.deconstruct_idx is not None:
            x = x[self.deconstruct_idx]
        return x.transpose(self.tranpose_dim, -1)


class SamePad(tnn.Module):
    def __init__(self, kernel_size, causal=False):
        super().__init__()
        self.remove = kernel_size - 1 if causal else (1 if kernel_size % 2 == 0 else 0)

    def forward(self, x):
        if self.remove > 0:
            x = x[:, :, : -self.remove]
        return x


class Fp32LayerNorm(tnn.LayerNorm):
    def forward(self, input):
        out = F.layer_norm(
            input.float(), self.normalized_shape,
            self.weight.float() if self.weight is not None else None,
            self.bias.float() if self.bias is not None else None, self.eps)
        return out.type_as(input)


class Fp32GroupNorm(tnn.GroupNorm):
    def forward(self, input):
        out = F.group_norm(
            input.float(), self.num_groups,
            self.weight.float() if self.weight is not None else None,
            self.bias.float() if self.bias is not None else None, self.eps)
        return out.type_as(input)


def LayerNorm(normalized_shape, eps=1e-5, elementwise_affine=True, export=False):
    return tnn.LayerNorm(normalized_shape, eps, elementwise_affine)


class GradMultiply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.new(x)

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.scale, None


@dataclass
class EMAModuleConfig:
    ema_decay: float = 0.9999
    ema_fp32: bool = False
    add_missing_params: bool = True
    log_norms: bool = False


class EMAModule:
    """fairseq.modules.EMAModule (semantics of ``step``/``restore``/``fp32_params``)."""

    def __init__(self, model, config, copy_model=True, device=None, skip_keys=None):
        import copy
        if copy_model:
            model = copy.deepcopy(model)
            model.requires_grad_(False)
        self.model = model
        self.config = config
        self.decay = config.ema_decay
        self.skip_keys = skip_keys or set()
        self.add_missing_params = config.add_missing_params
        self.fp32_params = {}
        if config.ema_fp32:
            self.build_fp32_params()
        self.log_norms = config.log_norms
        self.logs = {}

    def build_fp32_params(self, state_dict=None):
        if state_dict is None:
            state_dict = self.model.state_dict()
        for k, v in state_dict.items():
            if k in self.fp32_params:
                self.fp32_params[k].copy_(v)
            else:
                self.fp32_params[k] = v.detach().clone().float() if torch.is_floating_point(v) else v.clone()

    def restore(self, state_dict, build_fp32_params=False):
        self.model.load_state_dict(state_dict, strict=False)
        if build_fp32_params:
            self.build_fp32_params(state_dict)

    def set_decay(self, decay, weight_decay=None):
        self.decay = decay
        if weight_decay is not None:
            self.weight_decay = weight_decay

    def get_decay(self):
        return self.decay

    @torch.no_grad()
    def step(self, new_model):
        decay = self.decay
        ema_state_dict = {}
        ema_params = self.fp32_params if self.config.ema_fp32 else self.model.state_dict()
        new_p = []
        ema_p = []
        for key, param in new_model.named_parameters():
            if isinstance(param, dict):
                continue
            if not self.add_missing_params and key not in ema_params:
                continue
            try:
                ema_param = ema_params[key]
            except KeyError:
                ema_param = param.float().clone() if param.ndim == 1 else __import__("copy").deepcopy(param)
                ema_params[key] = ema_param
            if param.shape != ema_param.shape:
                raise ValueError("incompatible tensor shapes between model param and ema param")
            if "version" in key:
                continue
            lr = 1 - decay
            if key in self.skip_keys or not param.requires_grad:
                ema_params[key].copy_(param.to(dtype=ema_param.dtype).data)
                ema_param = ema_params[key]
            else:
                if self.log_norms:
                    new_p.append(param)
                    ema_p.append(ema_param)
                ema_param.mul_(1 - lr)
                ema_param.add_(param.data.to(dtype=ema_param.dtype), alpha=lr)
            ema_state_dict[key] = ema_param
        for key, param in new_model.named_buffers():
            ema_state_dict[key] = param
        if self.log_norms:
            if new_p:
                self.logs["model_norm"] = torch.norm(torch.stack([torch.norm(p.float()) for p in new_p]))
                self.logs["ema_norm"] = torch.norm(torch.stack([torch.norm(p.float()) for p in ema_p]))
        self.restore(ema_state_dict, build_fp32_params=False)


def init_bert_params(module):
    if isinstance(module, tnn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, tnn.Embedding):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()


def index_put(tensor, indices, value):
    tensor[indices] = value
    return tensor


def compute_mask_indices(*a, **k):
    raise RuntimeError("oracle: fairseq.compute_mask_indices is not reproducible here; pass precomputed_mask")


class BaseFairseqModel(tnn.Module):
    def __init__(self):
        super().__init__()

    def set_num_updates(self, num_updates):
        pass

    @classmethod
    def build_model(cls, cfg, task=None):
        raise NotImplementedError


def _register(name, dataclass=None):
    def deco(cls):
        return cls
    return deco


@dataclass
class FairseqDataclass:
    pass


class Mlp(tnn.Module):
    # timm.models.vision_transformer.Mlp (0.6.12): fc1 -> act -> drop1 -> fc2 -> drop2
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=tnn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = tnn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = tnn.Dropout(drop)
        self.fc2 = tnn.Linear(hidden_features, out_features)
        self.drop2 = tnn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class DropPath(tnn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep


class Meter:
    pass


class TextCompressionLevel(enum.Enum):
    none = 0
    low = 1
    high = 2


class RawAudioDataset:
    """fairseq.data.audio.raw_audio_dataset.RawAudioDataset @ 920a548, the parts FileAudioLabelDataset uses (third
    party, restated from its published semantics: parity unpinned): field bookkeeping, `postprocess`
    (channel mean, sample-rate check, whole-clip layer norm) and `crop_to_max_size` (numpy GLOBAL RNG)."""

    def __init__(self, sample_rate, max_sample_size=None, min_sample_size=0, shuffle=True, pad=False, normalize=False,
                 feature_encoder_spec=None, **_unused):
        import sys as _sys
        self.sample_rate = sample_rate
        self.sizes = []
        self.max_sample_size = max_sample_size if max_sample_size is not None else _sys.maxsize
        self.min_sample_size = min_sample_size
        self.pad, self.shuffle, self.normalize = pad, shuffle, normalize

    def __len__(self):
        return len(self.sizes)

    def postprocess(self, feats, curr_sample_rate):
        if feats.dim() == 2:
            feats = feats.mean(-1)
        if curr_sample_rate != self.sample_rate:
            raise Exception(f"sample rate: {curr_sample_rate}, need {self.sample_rate}")
        assert feats.dim() == 1, feats.dim()
        if self.normalize:
            with torch.no_grad():
                feats = F.layer_norm(feats, feats.shape)
        return feats

    def crop_to_max_size(self, t, target_size, dim=0):
        import numpy as _np
        size = t.size(dim)
        diff = size - target_size
        if diff <= 0:
            return t
        start = _np.random.randint(0, diff + 1)
        end = size - diff + start
        slices = [slice(None)] * t.dim()
        slices[dim] = slice(start, end)
        return t[tuple(slices)]

    def set_bucket_info(self, num_buckets):
        self.num_buckets = num_buckets


class TextCompressor:
    def __init__(self, level=None, max_input_byte_length=2 ** 16):
        self.level = level

    def compress(self, text):
        return text

    def decompress(self, compressed):
        return compressed


def _sf_read(path_or_fp, dtype="float32"):
    """soundfile.read(dtype="float32") for PCM16 wav files: int16 / 32768 (libsndfile's normalisation)."""
    import wave as _wave

    import numpy as _np
    with _wave.open(path_or_fp, "rb") as w:
        nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    assert width == 2, "the fixture corpus is PCM16"
    x = _np.frombuffer(raw, dtype="<i2").astype(_np.float32) / 32768.0
    return (x.reshape(-1, nch) if nch > 1 else x), sr


class _H5File:
    """h5py.File stand-in over an .npz container (the fixture corpus stores the four label arrays that way under the
    .h5 name the reference looks for): `with h5py.File(p, "r") as f: list(f["start_frame_lbl"])`."""

    def __init__(self, path, mode="r"):
        import numpy as _np
        self._z = _np.load(path)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self._z.close()
        return False

    def __getitem__(self, k):
        return self._z[k]

    def __contains__(self, k):
        return k in self._z.files


def install(reference_root="/root/reference"):
    """Install the shims and return the imported reference ``nn`` package."""
    if "nn" in sys.modules and getattr(sys.modules["nn"], "__file__", "").startswith(reference_root):
        return sys.modules["nn"]
    sys.meta_path.insert(0, _StubFinder())
    import importlib

    def mod(name):
        return importlib.import_module(name)

    fm = mod("fairseq.modules")
    for n, v in dict(TransposeLast=TransposeLast, SamePad=SamePad, Fp32LayerNorm=Fp32LayerNorm,
                     Fp32GroupNorm=Fp32GroupNorm, LayerNorm=LayerNorm, GradMultiply=GradMultiply,
                     EMAModule=EMAModule, EMAModuleConfig=EMAModuleConfig).items():
        setattr(fm, n, v)
    mod("fairseq.modules.transformer_sentence_encoder").init_bert_params = init_bert_params
    mod("fairseq.utils").index_put = index_put
    mod("fairseq.data.data_utils").compute_mask_indices = compute_mask_indices
    fmodels = mod("fairseq.models")
    fmodels.BaseFairseqModel = BaseFairseqModel
    fmodels.register_model = _register
    fmodels.FairseqEncoder = type("FairseqEncoder", (tnn.Module,), {})
    mod("fairseq.tasks").register_task = _register
    mod("fairseq.tasks").FairseqTask = type("FairseqTask", (), {})
    mod("fairseq.criterions").register_criterion = _register
    mod("fairseq.dataclass").FairseqDataclass = FairseqDataclass
    mod("fairseq.logging.meters").Meter = Meter
    mod("fairseq.data.text_compressor").TextCompressionLevel = TextCompressionLevel
    mod("fairseq.data.text_compressor").TextCompressor = TextCompressor
    mod("fairseq.data.audio.raw_audio_dataset").RawAudioDataset = RawAudioDataset
    au = mod("fairseq.data.audio.audio_utils")
    au.parse_path = lambda p: (p, [])
    au.read_from_stored_zip = _permissive
    au.is_sf_audio_data = lambda b: True
    mod("soundfile").read = _sf_read
    mod("h5py").File = _H5File
    for path, names in {
        "fairseq.tasks.audio_pretraining": ["AudioPretrainingConfig"],
        "fairseq.criterions.model_criterion": ["ModelCriterionConfig"],
        "fairseq.criterions.label_smoothed_cross_entropy": ["LabelSmoothedCrossEntropyCriterionConfig"],
        "fairseq.models.wav2vec": ["Wav2Vec2CtcConfig", "Wav2Vec2Config"],
    }.items():
        m = mod(path)
        for n in names:
            setattr(m, n, dataclass(type(n, (), {})))
    oc = mod("omegaconf")
    oc.II = lambda x: None
    oc.MISSING = "???"
    oc.open_dict = lambda *_a, **_k: contextlib.nullcontext()
    tv = mod("timm.models.vision_transformer")
    tv.Mlp = Mlp
    tv.DropPath = DropPath

    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    import nn  # noqa: the reference package
    assert nn.__file__.startswith(reference_root), nn.__file__
    return nn


LARGE_LAYERS = "[(127, 63, 1)] +[(512, 10, 5)] + [(512, 3, 2)] * 3 + [(512, 3, 1)] + [(512, 2, 1)] * 2"


def build_reference_model(refnn, *, embed_dim, num_heads, depth, prenet_depth, decoder_dim,
                          decoder_layers, decoder_kernel=7, decoder_groups=16, clone_batch=12,
                          conv_feature_layers=LARGE_LAYERS, sample_rate=8000, conv_pos_groups=16,
                          conv_pos_width=95, conv_pos_depth=5, average_top_k_layers=None,
                          mask_noise_std=0.01, seed=0, ema_decay=0.9997, source_mixup=-1.0,
                          ema_anneal_end_step=300000, local_grad_mult=1.0, encoder_zero_mask=True):
    """Build the reference Data2VecMultiModel with the large YAML's switches and given dims."""
    from nn.modalities.audio import D2vAudioConfig
    from nn.modalities.modules import D2vDecoderConfig
    from nn.data2vec2 import Data2VecMultiConfig, D2vModalitiesConfig, Data2VecMultiModel
    from nn import Modality

    dec = D2vDecoderConfig(decoder_dim=decoder_dim, decoder_groups=decoder_groups,
                           decoder_kernel=decoder_kernel, decoder_layers=decoder_layers,
                           input_dropout=0.1)
    audio = D2vAudioConfig(
        type=Modality.AUDIO, conv_feature_layers=conv_feature_layers, sample_rate=sample_rate,
        sinc_input=True, apply_window_to_root=False, use_pswish=True, sinc_norm="layer_norm",
        conv_pos_depth=conv_pos_depth, conv_pos_width=conv_pos_width, conv_pos_groups=conv_pos_groups,
        prenet_depth=prenet_depth, mask_prob=1.5, mask_length=2, mask_prob_adjust=0.05,
        inverse_mask=False, mask_noise_std=mask_noise_std, mask_dropout=0, add_masks=False,
        ema_local_encoder=False, use_alibi_encoder=True, prenet_layerdrop=0, prenet_dropout=0.1,
        learned_alibi_scale=True, learned_alibi_scale_per_head=True, num_alibi_heads=num_heads,
        model_depth=depth, decoder=dec, local_grad_mult=local_grad_mult,
        encoder_zero_mask=encoder_zero_mask)
    cfg = Data2VecMultiConfig(
        loss_beta=0, loss_scale=None, depth=depth, embed_dim=embed_dim, num_heads=num_heads,
        clone_batch=clone_batch, ema_decay=ema_decay, ema_end_decay=1.0,
        ema_anneal_end_step=ema_anneal_end_step, ema_encoder_only=False,
        average_top_k_layers=average_top_k_layers or depth, instance_norm_target_layer=True,
        layer_norm_target_layer=False, layer_norm_targets=False, layerdrop=0, norm_eps=1e-5,
        supported_modality=Modality.AUDIO, target_mixup=False, source_mixup=source_mixup,
        mixup_prob=1.0, same_mixup=True, mixing_window_length=0.05, gain_mode="A_weighting",
        modalities=D2vModalitiesConfig(audio=audio), max_update=384230, seed=1,
        unique_labels="['a']", with_labels=False, use_focal_loss=True, sample_rate=sample_rate,
        conv_feature_layers=conv_feature_layers, segmentation_metrics=False,
        verbose_tensorboard_logging=False)
    torch.manual_seed(seed)
    model = Data2VecMultiModel.build_model(cfg, task=None)
    return model, cfg
