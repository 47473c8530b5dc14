"""Configuration of the pretraining hot path.  Field names follow the reference's dataclasses
(nn/data2vec2.py:56-165 Data2VecMultiConfig, nn/modalities/audio.py:29-51 D2vAudioConfig,
nn/modalities/base.py:30-72 D2vModalityConfig, nn/modalities/modules.py:34-47 D2vDecoderConfig)
restricted to the switches the shipped recipe (configs/MeerKAT/a2v_large_pretrain_best.yaml) uses.
"""
from __future__ import annotations

import ast
import math
from dataclasses import dataclass, field
from typing import ClassVar, Dict, List, Optional, Tuple

LARGE_LAYERS_STR = "[(127, 63, 1)] +[(512, 10, 5)] + [(512, 3, 2)] * 3 + [(512, 3, 1)] + [(512, 2, 1)] * 2"


def parse_conv_layers(spec) -> List[Tuple[int, int, int]]:
    """The reference eval()s this string (audio.py:68); parse the same grammar safely
    (list literals of int tuples combined with + and *)."""
    if not isinstance(spec, str):
        return [tuple(int(v) for v in t) for t in spec]

    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
            return ev(node.left) + ev(node.right)
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mult):
            l, r = ev(node.left), ev(node.right)
            return l * r
        if isinstance(node, ast.List):
            return [ev(e) for e in node.elts]
        if isinstance(node, ast.Tuple):
            return tuple(ev(e) for e in node.elts)
        if isinstance(node, ast.Constant) and isinstance(node.value, int):
            return node.value
        raise ValueError(f"unsupported syntax in conv_feature_layers: {ast.dump(node)}")

    out = ev(ast.parse(spec.strip(), mode="eval"))
    return [tuple(int(v) for v in t) for t in out]


@dataclass
class D2vDecoderConfig:
    decoder_dim: int = 768
    decoder_groups: int = 16
    decoder_kernel: int = 7
    decoder_layers: int = 4
    input_dropout: float = 0.1
    decoder_residual: bool = True


@dataclass
class D2vAudioConfig:
    conv_feature_layers: str = LARGE_LAYERS_STR
    sample_rate: int = 8000
    conv_pos_width: int = 95
    conv_pos_groups: int = 16
    conv_pos_depth: int = 5
    sinc_input: bool = True
    sinc_norm: str = "layer_norm"
    use_pswish: bool = True
    prenet_depth: int = 8
    prenet_dropout: float = 0.1
    mask_prob: float = 1.5
    mask_length: int = 2
    mask_noise_std: float = 0.01
    mask_dropout: float = 0.0
    add_masks: bool = False
    encoder_zero_mask: bool = True
    mask_channel_prob: float = 0.0         # base.py:52-53; the finetune recipe overrides them (wav2vec2.py:113-114)
    mask_channel_length: int = 64
    prenet_layerdrop: float = 0.0
    use_alibi_encoder: bool = True
    alibi_scale: float = 1.0
    learned_alibi_scale: bool = True
    learned_alibi_scale_per_head: bool = True
    local_grad_mult: float = 1.0
    decoder: D2vDecoderConfig = field(default_factory=D2vDecoderConfig)
    # reference switches that are not built (refused at any other value, as Data2VecMultiConfig.UNBUILT)
    ema_local_encoder: bool = False        # base.py:60: EMA copy of the conv feature extractor
    apply_window_to_root: bool = False     # audio.py:40: SincConv window applied to the root
    inverse_mask: bool = False
    mask_prob_min: Optional[float] = None
    keep_masked_pct: float = 0.0
    remove_masks: bool = False
    num_extra_tokens: int = 0
    init_extra_token_zero: bool = True
    start_drop_path_rate: float = 0.0
    end_drop_path_rate: float = 0.0
    model_depth_override: Optional[int] = None     # (audio.py:33 II("model.depth"): informational)

    def __post_init__(self):
        bad = []
        for k, built in (("ema_local_encoder", False), ("apply_window_to_root", False), ("inverse_mask", False), ("mask_prob_min", None),
                         ("keep_masked_pct", 0.0), ("remove_masks", False), ("num_extra_tokens", 0), ("start_drop_path_rate", 0.0),
                         ("end_drop_path_rate", 0.0), ("sinc_input", True), ("sinc_norm", "layer_norm"), ("use_pswish", True),
                         ("use_alibi_encoder", True), ("learned_alibi_scale", True), ("learned_alibi_scale_per_head", True),
                         ("add_masks", False)):
            v = getattr(self, k)
            if (v is None) != (built is None) or (built is not None and type(built)(v) != built):
                bad.append(f"{k}={v!r} (built: {built!r})")
        if bad:
            raise NotImplementedError("modalities.audio switches that are not built: " + ", ".join(bad))


@dataclass
class Data2VecMultiConfig:
    depth: int = 16
    embed_dim: int = 1024
    num_heads: int = 16
    mlp_ratio: float = 4.0
    norm_eps: float = 1e-5
    clone_batch: int = 12
    encoder_dropout: float = 0.1
    post_mlp_drop: float = 0.1
    dropout_input: float = 0.0             # data2vec2.py:273,646-647: on the encoder input of the main blocks
    attention_dropout: float = 0.1
    layerdrop: float = 0.0
    average_top_k_layers: int = 16
    instance_norm_target_layer: bool = True
    ema_decay: float = 0.9997
    ema_end_decay: float = 1.0
    ema_anneal_end_step: int = 300000
    min_target_var: float = 0.1
    min_pred_var: float = 0.01
    loss_beta: float = 0.0
    source_mixup: float = 0.5
    mixup_prob: float = 1.0
    same_mixup: bool = True
    mixing_window_length: float = 0.05
    gain_mode: str = "A_weighting"
    seed: int = 1
    audio: D2vAudioConfig = field(default_factory=D2vAudioConfig)
    # ---- reference switches that are NOT built: carried only so that a recipe that sets one is refused (see UNBUILT below)
    loss_scale: Optional[float] = None
    layer_norm_target_layer: bool = False
    batch_norm_target_layer: bool = False
    instance_norm_targets: bool = False
    layer_norm_targets: bool = False
    end_of_block_targets: bool = False
    layer_norm_first: bool = False
    norm_affine: bool = True
    activation_dropout: float = 0.0
    start_drop_path_rate: float = 0.0
    end_drop_path_rate: float = 0.0
    skip_ema: bool = False
    mae_init: bool = False
    decoder_group: bool = False
    cls_loss: float = 0.0
    recon_loss: float = 0.0
    d2v_loss: float = 1.0
    with_labels: bool = False
    ema_encoder_only: bool = False         # (the shipped recipe's value, yaml:97; the reference's default is True)

    # field -> (the only value that is built, what the reference does otherwise)
    UNBUILT: ClassVar[Dict[str, tuple]] = {
        "loss_beta": (0.0, "smooth-L1 regression loss (data2vec2.py:1009-1012)"),
        "loss_scale": (None, "a constant loss scale instead of 1/sqrt(dim) (data2vec2.py:1014-1017)"),
        "instance_norm_target_layer": (True, "teacher layers averaged without instance norm (data2vec2.py:1040-1047)"),
        "layer_norm_target_layer": (False, "layer-normed teacher layers (data2vec2.py:1049-1050)"),
        "batch_norm_target_layer": (False, "batch-normed teacher layers (data2vec2.py:1029-1038)"),
        "instance_norm_targets": (False, "instance norm of the averaged target (data2vec2.py:1061-1062)"),
        "layer_norm_targets": (False, "layer norm of the averaged target (data2vec2.py:1058-1059)"),
        "end_of_block_targets": (False, "block outputs instead of FFN outputs as targets (modules.py:333-337)"),
        "layer_norm_first": (False, "pre-LN blocks (modules.py:316-327)"),
        "norm_affine": (True, "LayerNorms without affine parameters"),
        "activation_dropout": (0.0, "dropout inside the Mlp during pretraining (finetuning has its own switch, FinetuneConfig.activation_dropout)"),
        "start_drop_path_rate": (0.0, "stochastic depth (data2vec2.py:237-262)"),
        "end_drop_path_rate": (0.0, "stochastic depth (data2vec2.py:237-262)"),
        "ema_encoder_only": (False, "a teacher of the transformer blocks only, fed by the student's context encoder (data2vec2.py:368-371)"),
        "skip_ema": (False, "student as its own teacher (data2vec2.py:345,739)"),
        "mae_init": (False, "MAE-style initialisation (data2vec2.py:283-285)"),
        "decoder_group": (False, "optimizer group of the decoder (data2vec2.py:320-322)"),
        "cls_loss": (0.0, "class-token loss (data2vec2.py:871-887)"),
        "recon_loss": (0.0, "reconstruction loss (data2vec2.py:889-902)"),
        "d2v_loss": (1.0, "a weight on the regression loss (data2vec2.py:904-909)"),
        "with_labels": (False, "the linear-eval head inside pretraining (data2vec2.py:203-210,676-712,911-954)"),
    }

    def __post_init__(self):
        """A drop-in refuses what it does not implement (VERDICT r4 missing #3): every reference switch above is accepted at the
        one value the shipped recipe uses and raises at any other, instead of being carried and silently ignored."""
        for k, (built, what) in self.UNBUILT.items():
            v = getattr(self, k)
            same = (v is None) if built is None else (v is not None and type(built)(v) == built)
            if not same:
                raise NotImplementedError(f"{k}={v!r}: {what} is not built here (only {k}={built!r}, the shipped recipe's value)")

    # ---- derived
    @property
    def conv_layers(self):
        return parse_conv_layers(self.audio.conv_feature_layers)

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def pos_kernel(self):
        return max(3, self.audio.conv_pos_width // self.audio.conv_pos_depth)    # audio.py:90-91

    @property
    def mlp_dim(self):
        return int(self.embed_dim * self.mlp_ratio)

    def frame_counts(self, n_samples: int):
        """nn/utils.py:1085-1090: Conv1d padding 'same' for stride 1 else ceil(stride/2)."""
        out, t = [], n_samples
        for (_, k, s) in self.conv_layers:
            if s != 1:
                p = int(math.ceil(s / 2))
                t = (t + 2 * p - (k - 1) - 1) // s + 1
            out.append(t)
        return out


def large_config(**kw) -> Data2VecMultiConfig:
    """configs/MeerKAT/a2v_large_pretrain_best.yaml:84-147"""
    return Data2VecMultiConfig(**kw)


def large_44k_config(**kw) -> Data2VecMultiConfig:
    """BASELINE config 5 (NIPS4Bplus-shaped, 44.1 kHz): the large model behind the front-end spec SURVEY 8d derives from
    the YAML's own rule k ~ sqrt(sr / 2): 296 sinc filters x 149 taps (the reference ships no 44.1 kHz YAML)."""
    layers = "[(296, 149, 1)] + [(512, 10, 5)] + [(512, 3, 2)] * 3 + [(512, 3, 1)] + [(512, 2, 1)] * 2"
    d = dict(audio=D2vAudioConfig(conv_feature_layers=layers, sample_rate=44100))
    d.update(kw)
    return Data2VecMultiConfig(**d)


def base_star_config(**kw) -> Data2VecMultiConfig:
    """Build-defined 'animal2vec-base' (the reference ships only the large YAML)."""
    a = D2vAudioConfig(prenet_depth=4, decoder=D2vDecoderConfig(decoder_dim=384, decoder_layers=4))
    d = dict(depth=12, embed_dim=768, num_heads=12, clone_batch=8, average_top_k_layers=12, audio=a)
    d.update(kw)
    return Data2VecMultiConfig(**d)


def tiny_config(conv_feature_layers=None, **kw) -> Data2VecMultiConfig:
    """The dims of tests/golden/tiny_pretrain.npz (oracle/gen_golden.py)."""
    layers = conv_feature_layers or "[(31, 15, 1)] + [(32, 10, 5)] + [(32, 3, 2)] * 3 + [(32, 3, 1)] + [(32, 2, 1)] * 2"
    a = D2vAudioConfig(conv_feature_layers=layers, conv_pos_groups=4, prenet_depth=1,
                       decoder=D2vDecoderConfig(decoder_dim=32, decoder_groups=4, decoder_layers=2))
    d = dict(depth=2, embed_dim=64, num_heads=4, clone_batch=2, average_top_k_layers=2, audio=a)
    d.update(kw)
    return Data2VecMultiConfig(**d)
