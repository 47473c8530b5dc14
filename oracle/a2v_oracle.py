"""CPU oracle: plain-PyTorch fp32 restatement of animal2vec's pretraining hot path.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package ``animal2vec_amd`` never does.

Parity status: PINNED for everything under /root/reference -- ``oracle/gen_golden.py`` imports the
live reference (through ``oracle/ref_shims.py``) in the build container, checks every function
here against it on identical inputs/weights and commits the resulting vectors under
``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the oracle against those vectors
everywhere.  UNPINNED for the third-party arithmetic the reference delegates to packages absent
from /root/reference (fairseq@920a548 ``compute_mask_indices`` RNG stream, ``EMAModule``,
``Fp32LayerNorm``/``SamePad``/``TransposeLast``; timm 0.6.12 ``Mlp``): restated from their
published semantics, see SURVEY.md section 8c.

Every function cites the reference lines it follows (paths relative to /root/reference).
Parameters are passed as a flat ``dict[str, Tensor]`` keyed by the reference's state-dict names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

P = "modality_encoders.AUDIO."

LARGE_LAYERS = [(127, 63, 1), (512, 10, 5)] + [(512, 3, 2)] * 3 + [(512, 3, 1)] + [(512, 2, 1)] * 2


@dataclass
class OracleConfig:
    """Dims of one Data2VecMultiModel instance (switches fixed to the shipped large YAML,
    configs/MeerKAT/a2v_large_pretrain_best.yaml:84-147)."""
    embed_dim: int = 1024
    num_heads: int = 16
    depth: int = 16
    prenet_depth: int = 8
    decoder_dim: int = 768
    decoder_layers: int = 4
    decoder_kernel: int = 7
    decoder_groups: int = 16
    clone_batch: int = 12
    conv_feature_layers: List[Tuple[int, int, int]] = field(default_factory=lambda: list(LARGE_LAYERS))
    sample_rate: int = 8000
    conv_pos_width: int = 95
    conv_pos_depth: int = 5
    conv_pos_groups: int = 16
    average_top_k_layers: int = 16
    norm_eps: float = 1e-5
    mlp_ratio: float = 4.0
    mask_noise_std: float = 0.01
    ema_decay: float = 0.9997
    ema_end_decay: float = 1.0
    ema_anneal_end_step: int = 300000
    min_low_hz: float = 50.0

    @property
    def pos_kernel(self) -> int:
        # nn/modalities/audio.py:90-91
        return max(3, self.conv_pos_width // self.conv_pos_depth)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


def frame_counts(n_samples: int, layers: Sequence[Tuple[int, int, int]]) -> List[int]:
    """Output length of every feature-extractor layer (nn/utils.py:1085-1090: Conv1d with
    padding='same' when stride==1 else ceil(stride/2); SincConv is 'same', nn/sinc.py:123-126)."""
    out, t = [], n_samples
    for i, (_, k, s) in enumerate(layers):
        if s == 1:
            pass
        else:
            p = int(math.ceil(s / 2))
            t = (t + 2 * p - (k - 1) - 1) // s + 1
        out.append(t)
    return out


# ------------------------------------------------------------------------------ SincConv (a1-a3)
def sinc_init(out_channels: int, kernel_size: int, sample_rate: int, min_low_hz: float = 50.0):
    """nn/sinc.py:225-284 mel-spaced init -> (low_hz_, band_hz_), each (C,1) fp32."""
    min_band_hz = int(np.ceil(sample_rate / kernel_size))                       # sinc.py:79
    high_hz = sample_rate / 2 - (min_low_hz + min_band_hz)                      # :229

    def to_mel(hz):
        return 2595 * np.log10(1 + hz / 700)                                    # :278-280

    mel = torch.linspace(to_mel(min_low_hz), to_mel(high_hz), out_channels + 1)  # :232-236
    hz = 700 * (10 ** (mel / 2595) - 1)                                         # :282-284
    return hz[:-1].unsqueeze(1).clone(), (hz[1:] - hz[:-1]).unsqueeze(1).clone()  # :247-248


def sinc_window_and_n(kernel_size: int, sample_rate: int):
    """nn/sinc.py:264-276: half Hamming window (k//2,) and time axis n_ (1,k//2)."""
    n_lin = torch.linspace(0, (kernel_size / 2) - 1, steps=int(kernel_size / 2))
    window = 0.53836 - 0.46164 * torch.cos(2 * math.pi * n_lin / kernel_size)
    n = (kernel_size - 1) / 2.0
    n_ = 2 * math.pi * torch.arange(-n, 0).view(1, -1) / sample_rate
    return window.float(), n_.float()


def sinc_filters(low_hz_, band_hz_, kernel_size: int, sample_rate: int, min_low_hz: float = 50.0):
    """nn/sinc.py:181-223 -> (C,1,k) fp32, differentiable in low_hz_/band_hz_."""
    min_band_hz = int(np.ceil(sample_rate / kernel_size))
    window_, n_ = sinc_window_and_n(kernel_size, sample_rate)
    low = min_low_hz + torch.abs(low_hz_)                                        # :184
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_), min_low_hz, sample_rate / 2)  # :187-191
    band = (high - low)[:, 0]                                                    # :192
    f_low = torch.matmul(low.float(), n_)                                        # :200
    f_high = torch.matmul(high.float(), n_)                                      # :201
    left = (torch.sin(f_high) - torch.sin(f_low)) / n_ * 2 * window_             # :204
    center = 2 * band.view(-1, 1)                                                # :207
    right = torch.flip(left, dims=[1])                                           # :210
    bp = torch.cat([left, center, right], dim=1)                                 # :213-215
    bp = bp / (2 * band[:, None])                                                # :218
    return bp.view(-1, 1, kernel_size)                                           # :221


def sinc_conv(x, low_hz_, band_hz_, kernel_size: int, sample_rate: int):
    """nn/sinc.py:107-159 with padding='same', reflect, stride 1: (B,L) or (B,1,L) -> (B,C,L)."""
    if x.ndim == 2:
        x = x.unsqueeze(1)
    # get_padding_elem(L_in=in_channels=1, ...) (sinc.py:305-308,330-336): L_out = 1-(k-1)-1+1
    l_out = math.floor((1 - (kernel_size - 1) - 1) / 1) + 1
    pad = math.floor((1 - l_out) / 2)
    x = F.pad(x, (pad, pad), mode="reflect")                                     # :311
    w = sinc_filters(low_hz_, band_hz_, kernel_size, sample_rate)
    return F.conv1d(x.float(), w.float(), stride=1, padding=0)                   # :144-151


def pswish(x, alpha, beta):
    """nn/utils.py:1430-1431, alpha/beta (1,C,1)."""
    return x * alpha * torch.sigmoid(beta * x)


def channel_layer_norm(x, weight, bias, eps=1e-5):
    """TransposeLast . Fp32LayerNorm(C) . TransposeLast on (B,C,T) (nn/utils.py:1111-1115)."""
    y = F.layer_norm(x.transpose(1, 2).float(), (x.shape[1],), weight, bias, eps)
    return y.transpose(1, 2)


# ------------------------------------------------------------------------------ front end (a4-a6)
def feature_extractor(p: Dict[str, torch.Tensor], source, cfg: OracleConfig, return_all=False):
    """ConvFeatureExtractionModel.forward (nn/utils.py:1155-1163), mode='layer_norm',
    sinc_input, use_pswish.  source (B,L) -> (B,512,T)."""
    pre = P + "local_encoder.conv_layers."
    outs = []
    x = source.unsqueeze(1)
    for i, (dim, k, s) in enumerate(cfg.conv_feature_layers):
        if i == 0:
            x = sinc_conv(x, p[pre + "0.0.low_hz_"], p[pre + "0.0.band_hz_"], k, cfg.sample_rate)
            x = channel_layer_norm(x, p[pre + "0.2.1.weight"], p[pre + "0.2.1.bias"])
            x = pswish(x, p[pre + "0.3.p_swish_alpha"], p[pre + "0.3.p_swish_beta"])
        else:
            w = p[pre + f"{i}.0.weight"]
            if s == 1:
                total = k - 1                       # torch padding='same': left=total//2, right=rest
                x = F.pad(x, (total // 2, total - total // 2))
                x = F.conv1d(x, w, stride=1)
            else:
                x = F.conv1d(x, w, stride=s, padding=int(math.ceil(s / 2)))     # utils.py:1089
            x = channel_layer_norm(x, p[pre + f"{i}.2.1.weight"], p[pre + f"{i}.2.1.bias"])
            x = F.gelu(x)
        outs.append(x)
    return outs if return_all else x


def project_features(p, x):
    """nn/modalities/audio.py:83-88: (B,512,T) -> (B,T,D)."""
    x = x.transpose(1, 2)
    x = F.layer_norm(x.float(), (x.shape[-1],), p[P + "project_features.1.weight"],
                     p[P + "project_features.1.bias"], 1e-5)
    return F.linear(x, p[P + "project_features.2.weight"], p[P + "project_features.2.bias"])


def local_features(p, source, cfg, local_grad_mult: float = 1.0):
    """nn/modalities/base.py:194-213: local_grad_mult == 1 (pretrain) runs the conv feature extractor with
    gradients; local_grad_mult == 0 (finetune: feature_grad_mult 0.0, finetune_mixup_100.yaml:85 ->
    wav2vec2.py:122) runs it under no_grad -- project_features keeps its gradients either way."""
    if local_grad_mult > 0:
        assert local_grad_mult == 1.0, "GradMultiply with a fractional factor is not on the benchmarked path"
        x = feature_extractor(p, source, cfg)
    else:
        with torch.no_grad():
            x = feature_extractor(p, source, cfg)
    return project_features(p, x)


# ------------------------------------------------------------------------------ masking (a8)
def make_maskinfo(mask):
    """nn/modalities/base.py:427-455.  mask (N,T) with 1 = masked, equal count per row.
    Returns ids_keep (N,K), ids_restore (N,T) (int64).  The reference calls argsort without
    ``stable``; ties are resolved here with stable=True (ascending original position), which
    gen_golden.py checks is what torch's CPU sort produced for the reference too."""
    mask = mask.to(torch.uint8)
    ids_shuffle = mask.argsort(dim=1, stable=True)
    ids_restore = ids_shuffle.argsort(dim=1, stable=True)
    len_keep = mask.shape[1] - int(mask[0].sum())
    return ids_shuffle[:, :len_keep], ids_restore


def seeded_mask(n_rows: int, T: int, keep: int, seed: int) -> torch.Tensor:
    """Deterministic precomputed_mask used by tests/bench: per row a seeded randperm
    keeps ``keep`` frames.  (Not the fairseq span sampler -- that stream is unpinned.)"""
    g = torch.Generator().manual_seed(seed)
    mask = torch.ones(n_rows, T, dtype=torch.uint8)
    for r in range(n_rows):
        mask[r, torch.randperm(T, generator=g)[:keep]] = 0
    return mask


def _mix32(u):
    u = u.astype(np.uint32)
    u ^= u >> np.uint32(16); u *= np.uint32(0x7FEB352D); u ^= u >> np.uint32(15); u *= np.uint32(0x846CA68B); u ^= u >> np.uint32(16)
    return u


def _mask_hash(seed: int, purpose: int, row: int, i):
    """the counter stream of animal2vec_amd/csrc/a2v_mask.hip::mask_hash (uint32 arithmetic)"""
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        a = _mix32(np.asarray([np.uint32(row) * np.uint32(0x9E3779B1) + lo + np.uint32(purpose) * np.uint32(0x85EBCA6B)], dtype=np.uint32))[0]
        return _mix32((a ^ hi) + np.asarray(i, dtype=np.uint32) * np.uint32(0xC2B2AE35) + np.uint32(0x27D4EB2F))


def span_mask(n_rows: int, T: int, mask_prob: float, mask_length: int, seed: int, min_masks: int = 1,
              mask_dropout: float = 0.0) -> np.ndarray:
    """Span masks with the DISTRIBUTION of fairseq@920a548 ``compute_mask_indices`` as ModalitySpecificEncoder.compute_mask
    calls it (nn/modalities/base.py:401-413: static span length, no_overlap False, min_masks 1, require_same_masks True,
    mask_dropout; the same law as animal2vec_amd.masking.compute_mask_indices) drawn from the counter-based stream of the
    device sampler (a2v_span_mask), restated here draw for draw: 'n of M without replacement' = the n smallest of the unique
    keys (hash32 << 14 | frame).  fairseq's own numpy stream is third party and unpinned; this function pins
    the kernel, the statistical tests pin the law.  Returns (n_rows, T) bool, equal count per row."""
    frames = np.arange(T, dtype=np.uint64)
    masks, counts = [], []
    for r in range(n_rows):
        u = float(int(_mask_hash(seed, 1, r, [0])[0]) >> 8) / 16777216.0
        num = max(min_masks, int(float(mask_prob) * T / float(mask_length) + u))
        min_len = mask_length
        if T - min_len <= num:
            min_len = T - num - 1
        M = max(1, min(T, T - min_len))
        num = min(num, M)
        key = (_mask_hash(seed, 2, r, np.arange(M)).astype(np.uint64) << np.uint64(14)) | frames[:M]
        starts = np.argsort(key)[:num]
        m = np.zeros(T, dtype=bool)
        for off in range(mask_length):
            idx = starts + off
            m[idx[idx < T]] = True
        masks.append(m)
        counts.append(int(m.sum()))
    mn = min(counts)
    n_final = mn - int(np.rint(mn * float(mask_dropout)))
    for r, m in enumerate(masks):
        if n_final < counts[r]:
            idx = np.nonzero(m)[0]
            key = (_mask_hash(seed, 3, r, idx).astype(np.uint64) << np.uint64(14)) | idx.astype(np.uint64)
            keep_masked = idx[np.argsort(key)[:n_final]]
            m[:] = False
            m[keep_masked] = True
    return np.stack(masks)


# ------------------------------------------------------------------------------ pos-conv (a9)
def pos_conv(p, prefix, x, cfg: OracleConfig):
    """relative_positional_encoder (nn/modalities/audio.py:90-113): (N,T,D)->(N,T,D)."""
    k = cfg.pos_kernel
    x = x.transpose(1, 2)
    for i in range(1, cfg.conv_pos_depth + 1):
        x = F.conv1d(x, p[prefix + f"relative_positional_encoder.{i}.0.weight"],
                     p[prefix + f"relative_positional_encoder.{i}.0.bias"],
                     padding=k // 2, groups=cfg.conv_pos_groups)
        if k % 2 == 0:                                  # SamePad (fairseq): trim 1 for even k
            x = x[:, :, :-1]
        x = F.layer_norm(x.transpose(1, 2).float(), (x.shape[1],), None, None, 1e-5).transpose(1, 2)
        x = F.gelu(x)
    return x.transpose(1, 2)


# ------------------------------------------------------------------------------ ALiBi (a10)
def alibi_slopes(n: int) -> List[float]:
    """nn/modalities/base.py:559-576."""
    def pow2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + alibi_slopes(2 * c)[0::2][: n - c]


def alibi_bias_from_positions(pos, num_heads: int, alibi_scale):
    """Restates get_alibi/get_alibi_bias/masked_alibi + the scale fold (base.py:292-314,
    553-698): bias[n,h,i,j] = -slope_h * clamp_min(scale_h,0) * |pos[n,i]-pos[n,j]|, fp32.
    pos (N,S) int64; alibi_scale (1,1,H,1,1)."""
    slopes = torch.tensor(alibi_slopes(num_heads), dtype=torch.float32)
    dist = (pos[:, :, None] - pos[:, None, :]).abs().to(torch.float32)           # (N,S,S)
    bias = slopes.view(1, -1, 1, 1) * (dist.unsqueeze(1) * -1)                   # base.py:586-617
    scale = alibi_scale.clamp_min(0).squeeze(0)                                  # :305-307
    return bias * scale.type_as(bias)


# ------------------------------------------------------------------------------ blocks (a11-a14)
# Test probe (not part of the restated arithmetic): a list, or None.  d loss / d alibi_scale_h = sum over layers and (n, i, j) of
# dS_nhij * bias_nhij / scale_h is dominated by cancellation; with a list here every student attention's backward appends
# sum_nij |dS| |bias| per head -- scale_h times the sum of the ABSOLUTE values of that gradient's terms, the natural yardstick of
# its rounding error (tests/test_pretrain_gpu.py::alibi_term_scale).
ALIBI_PROBE = None


def alt_attention(p, prefix, x, bias, num_heads):
    """AltAttention.forward (nn/modalities/modules.py:368-410), eval mode."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, p[prefix + "attn.qkv.weight"], p[prefix + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * hd ** -0.5                                                            # :391
    attn = q @ k.transpose(-2, -1)                                                # :392
    if bias is not None:
        attn = attn.float() + bias                                                # :394-396
        if ALIBI_PROBE is not None and attn.requires_grad:
            attn.register_hook(lambda g_, b_=bias.detach(): ALIBI_PROBE.append((g_.detach().abs() * b_.abs()).sum(dim=(0, 2, 3))))
    attn = attn.softmax(dim=-1, dtype=torch.float32)                              # :404
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)                               # :406-407
    return F.linear(x, p[prefix + "attn.proj.weight"], p[prefix + "attn.proj.bias"])


def alt_block(p, prefix, x, bias, num_heads, eps):
    """AltBlock.forward post-LN branch with ffn_targets (modules.py:328-337) -> (x, t)."""
    D = x.shape[-1]
    x = x + alt_attention(p, prefix, x, bias, num_heads)
    r = x = F.layer_norm(x, (D,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    h = F.gelu(F.linear(x, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"]))
    x = F.linear(h, p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"])
    t = x
    x = F.layer_norm(r + x, (D,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    return x, t


def prenet(p, prefix, x, bias, cfg: OracleConfig):
    """BlockEncoder.forward (modules.py:83-108), layer_norm_first=False, eval."""
    pre = prefix + "context_encoder."
    x = F.layer_norm(x, (x.shape[-1],), p[pre + "norm.weight"], p[pre + "norm.bias"], cfg.norm_eps)
    for i in range(cfg.prenet_depth):
        x, _ = alt_block(p, pre + f"blocks.{i}.", x, bias, cfg.num_heads, cfg.norm_eps)
    return x


# ------------------------------------------------------------------------------ decoder (a15)
def decoder_input(x, ids_restore, noise):
    """ModalitySpecificEncoder.decoder_input (base.py:162-192), eval: x (N,K,D), noise
    (N,T-K,D) ~ N(0, mask_noise_std) supplied by the caller -> (N,T,D)."""
    x_ = torch.cat([x, noise], dim=1)
    idx = ids_restore.unsqueeze(-1).expand(-1, -1, x.shape[-1])
    return torch.gather(x_, 1, idx)


def decoder1d(p, x, cfg: OracleConfig):
    """Decoder1d.forward (modules.py:179-192): (N,T,D) -> (N,T,D)."""
    pre = P + "decoder."
    x = x.transpose(1, 2)
    residual = x
    for i in range(cfg.decoder_layers):
        y = F.conv1d(x, p[pre + f"blocks.{i}.0.weight"], p[pre + f"blocks.{i}.0.bias"],
                     padding=cfg.decoder_kernel // 2, groups=cfg.decoder_groups)
        if cfg.decoder_kernel % 2 == 0:
            y = y[:, :, :-1]
        y = F.layer_norm(y.transpose(1, 2), (y.shape[1],), None, None, 1e-5).transpose(1, 2)
        y = F.gelu(y)
        if residual.size(1) == y.size(1):                                         # modules.py:124-134
            y = y + residual
        x = residual = y
    x = x.transpose(1, 2)
    return F.linear(x, p[pre + "proj.weight"], p[pre + "proj.bias"])


# ------------------------------------------------------------------------------ targets / loss
def make_targets(ts: List[torch.Tensor], num_layers: int):
    """Data2VecMultiModel.make_targets (data2vec2.py:1023-1066), instance_norm_target_layer."""
    ts = ts[-num_layers:]
    ts = [F.instance_norm(t.transpose(1, 2).float()).transpose(1, 2) for t in ts]
    y = ts[0].float().clone()
    for t in ts[1:]:
        y.add_(t.float())
    return y.div_(len(ts))


def d2v_loss(x, y):
    """data2vec2.py:1005-1021, loss_beta=0, loss_scale=None: unreduced MSE * D^-0.5."""
    x = x.reshape(-1, x.size(-1)).float()
    y = y.reshape(-1, x.size(-1))
    return F.mse_loss(x, y, reduction="none") * (1 / math.sqrt(x.size(-1)))


def compute_var(y):
    """data2vec2.py:1095-1110 (single process)."""
    y = y.reshape(-1, y.size(-1))
    return torch.sqrt(y.var(dim=0) + 1e-6).mean()


# ------------------------------------------------------------------------------ student / teacher
def student_forward(p, source, mask, noise, cfg: OracleConfig, features=None):
    """Steps 2-4 of Data2VecMultiModel.forward (data2vec2.py:629-674,741-748) in eval mode.
    mask (B*cb,T) 1=masked; noise (B*cb,T-K,D).  Returns dict of intermediates."""
    out = {}
    feats = local_features(p, source, cfg) if features is None else features     # (B,T,D)
    out["local_features"] = feats
    cb = cfg.clone_batch
    x = feats.repeat_interleave(cb, 0)                                            # base.py:244
    ids_keep, ids_restore = make_maskinfo(mask)
    D = x.shape[-1]
    gi = ids_keep.unsqueeze(-1).expand(-1, -1, D)
    x_unmasked = torch.gather(x, 1, gi)                                           # base.py:447
    xm = x * (1 - mask.type_as(x).unsqueeze(-1))                                  # base.py:464
    x_pos = pos_conv(p, P, xm, cfg)                                               # base.py:274
    out["x_pos"] = x_pos
    x = x_unmasked + torch.gather(x_pos, 1, gi)                                   # base.py:278-280
    out["prenet_in"] = x
    bias = alibi_bias_from_positions(ids_keep, cfg.num_heads, p[P + "alibi_scale"])
    out["alibi_bias"] = bias
    x = prenet(p, P, x, bias, cfg)
    out["prenet_out"] = x
    for i in range(cfg.depth):                                                    # data2vec2.py:650-671
        x, _ = alt_block(p, f"blocks.{i}.", x, bias, cfg.num_heads, cfg.norm_eps)
    out["encoder_out"] = x
    dx = decoder_input(x, ids_restore, noise)
    out["decoder_in"] = dx
    out["decoder_out"] = decoder1d(p, dx, cfg)
    out["ids_keep"], out["ids_restore"] = ids_keep, ids_restore
    return out


@torch.no_grad()
def teacher_forward(ema: Dict[str, torch.Tensor], feats, cfg: OracleConfig, return_layers=False):
    """Teacher section (data2vec2.py:779-846): feats (B,T,D) student local features ->
    y (B,T,D) targets.  ``ema`` uses the same key names (no decoder/local_encoder/project)."""
    B, T, D = feats.shape
    x = feats + pos_conv(ema, P, feats, cfg)                                      # base.py:273-290
    pos = torch.arange(T).unsqueeze(0).expand(B, -1)
    bias = alibi_bias_from_positions(pos, cfg.num_heads, ema[P + "alibi_scale"])
    x = prenet(ema, P, x, bias, cfg)
    ts = []
    for i in range(cfg.depth):
        x, t = alt_block(ema, f"blocks.{i}.", x, bias, cfg.num_heads, cfg.norm_eps)
        ts.append(t)
    y = make_targets(ts, cfg.average_top_k_layers)
    return (y, ts) if return_layers else y


@torch.no_grad()
def features_only_forward(p, source, cfg: OracleConfig):
    """Data2VecMultiModel.forward(features_only=True, mask=False) = extract_features
    (data2vec2.py:629-728,1112-1123): student weights, all T frames.  -> (x, layer_results)."""
    feats = local_features(p, source, cfg)
    B, T, D = feats.shape
    x = feats + pos_conv(p, P, feats, cfg)
    pos = torch.arange(T).unsqueeze(0).expand(B, -1)
    bias = alibi_bias_from_positions(pos, cfg.num_heads, p[P + "alibi_scale"])
    x = prenet(p, P, x, bias, cfg)
    lrs = []
    for i in range(cfg.depth):
        x, t = alt_block(p, f"blocks.{i}.", x, bias, cfg.num_heads, cfg.norm_eps)
        lrs.append(t)
    return x, lrs


# ------------------------------------------------------------------------------ finetune path
def sigmoid_focal_loss(inputs, targets, alpha: float = 0.25, gamma: float = 2, reduction: str = "none"):
    """nn/utils.py:971-1010 (RetinaNet focal loss on logits; criterion 'finetunecriterion' calls it with
    reduction='sum', criterions.py:243-248)."""
    inputs, targets = inputs.float(), targets.float()
    pr = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = pr * targets + (1 - pr) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def apply_mask(x, mask_t=None, replacement=None, mask_c=None):
    """ModalitySpecificEncoder.apply_mask (base.py:457-485) with the random draws made explicit: frames where
    mask_t (B,T) is set take the rows of `replacement` ((num_masked, D), the reference's normal_(0, mask_noise_std)
    draw; None = encoder_zero_mask), then the channels where mask_c (B,D) is set are zeroed on every frame."""
    if mask_t is not None:
        mt = mask_t.bool()
        if replacement is None:
            x = x * (1 - mt.unsqueeze(-1).type_as(x))
        else:
            x = x.clone()
            x[mt] = replacement.type_as(x)
    if mask_c is not None:
        x = x.masked_fill(mask_c.bool().unsqueeze(1).expand(-1, x.shape[1], -1), 0.0)
    return x


def finetune_forward(p, head_w, head_b, source, cfg: OracleConfig, top_k=None, local_grad_mult: float = 0.0,
                     mask_t=None, replacement=None, mask_c=None, active=None):
    """Wav2VecEncoderModOut.forward (wav2vec2.py:430-482): extract_features of the student, mean of the last
    `average_top_k_layers` layer results (the blocks' FFN outputs t, modules.py:328-337), final_dropout (0.0),
    Linear(D -> n_labels).  Returns logits (B, T, C).
    Eval form by default (mask=False).  Train form, dropouts excluded: mask_t / replacement / mask_c as in apply_mask
    (remove_masked is False with features_only, so the masked input feeds the pos-conv and the encoder at full
    length, base.py:268-298); active = layerdrop outcome per block, prenet blocks first (modules.py:89-94,
    data2vec2.py:651-655: a dropped block contributes no layer result)."""
    top_k = cfg.average_top_k_layers if top_k is None else top_k
    feats = local_features(p, source, cfg, local_grad_mult)
    feats = apply_mask(feats, mask_t, replacement, mask_c)
    B, T, D = feats.shape
    x = feats + pos_conv(p, P, feats, cfg)
    pos = torch.arange(T).unsqueeze(0).expand(B, -1)
    bias = alibi_bias_from_positions(pos, cfg.num_heads, p[P + "alibi_scale"])
    active = [True] * (cfg.prenet_depth + cfg.depth) if active is None else list(active)
    pre = P + "context_encoder."
    x = F.layer_norm(x, (D,), p[pre + "norm.weight"], p[pre + "norm.bias"], cfg.norm_eps)
    for i in range(cfg.prenet_depth):
        if active[i]:
            x, _ = alt_block(p, pre + f"blocks.{i}.", x, bias, cfg.num_heads, cfg.norm_eps)
    lrs = []
    for i in range(cfg.depth):
        if active[cfg.prenet_depth + i]:
            x, t = alt_block(p, f"blocks.{i}.", x, bias, cfg.num_heads, cfg.norm_eps)
            lrs.append(t)
    sel = lrs[-top_k:]
    xm = sum(sel) / len(sel)
    return F.linear(xm, head_w, head_b)


def pretrain_forward(p, ema, source, mask, noise, cfg: OracleConfig):
    """Whole eval-mode pretrain forward (data2vec2.py:516-991, mixup off): returns the net_output
    dict the reference returns plus intermediates under '_dbg'."""
    s = student_forward(p, source, mask, noise, cfg)
    y = teacher_forward(ema, s["local_features"].detach(), cfg)
    mb = mask.bool()
    y_sel = y.repeat_interleave(cfg.clone_batch, 0)[mb]                           # data2vec2.py:850-856
    x_sel = s["decoder_out"][mb]                                                  # :859-860
    loss = d2v_loss(x_sel, y_sel)                                                 # :904-909
    T = mask.shape[1]
    res = {
        "losses": {"AUDIO_regression": loss},
        "sample_size": mask.sum().long(),                                         # :864
        "masked_pct": 1 - s["ids_keep"].shape[1] / T,                             # :959-961
        "pred_var": compute_var(x_sel.float().detach()),                          # :964
        "target_var": compute_var(y_sel.float()),                                 # :970
        "_dbg": dict(s, targets=y),
    }
    return res


# ------------------------------------------------------------------------------ EMA (a20)
def get_annealed_rate(start, end, curr_step, total_steps):
    """nn/modalities/base.py:492-497."""
    if curr_step >= total_steps:
        return end
    r = end - start
    return end - r * (1 - curr_step / total_steps)


def ema_decay_at(cfg: OracleConfig, num_updates: int) -> float:
    """data2vec2.py:395-406."""
    if cfg.ema_decay == cfg.ema_end_decay:
        return cfg.ema_decay
    if num_updates >= cfg.ema_anneal_end_step:
        return cfg.ema_end_decay
    return get_annealed_rate(cfg.ema_decay, cfg.ema_end_decay, num_updates, cfg.ema_anneal_end_step)


@torch.no_grad()
def ema_step(p, ema, decay: float):
    """fairseq EMAModule.step (restated): ema <- ema*decay + (1-decay)*p for every student
    parameter that exists in the teacher; in place on ``ema``."""
    if decay >= 1:
        return ema
    for k, v in ema.items():
        if k in p:
            v.mul_(decay).add_(p[k].detach().float(), alpha=1 - decay)
    return ema


# ------------------------------------------------------------------------------ BC mixup (a7)
def a_weight_power(fs: int, n_fft: int, min_db: float = -80.0) -> torch.Tensor:
    """data2vec2.py:461-479 -> (n_fft//2+1,) float64 linear power weights."""
    freq = np.linspace(0, fs // 2, n_fft // 2 + 1)
    fsq = freq ** 2
    fsq[0] = 1.0
    w = 2.0 + 20.0 * (2 * np.log10(12194) + 2 * np.log10(fsq) - np.log10(fsq + 12194 ** 2)
                      - np.log10(fsq + 20.6 ** 2) - 0.5 * np.log10(fsq + 107.7 ** 2)
                      - 0.5 * np.log10(fsq + 737.9 ** 2))
    w = np.maximum(w, min_db)
    return torch.from_numpy(np.power(10, w / 10))


def compute_gain(sound, fs=8000, wl=0.05, min_db=-80.0):
    """compute_gain_torch, mode A_weighting (data2vec2.py:453-498): (B,L) -> (B,n_frames) dB."""
    n_fft = round(fs * wl)
    frames = sound.unfold(-1, n_fft, n_fft // 2)
    w = torch.hann_window(n_fft) * frames
    spec = torch.fft.rfft(w)
    g = (spec.abs() ** 2 * a_weight_power(fs, n_fft, min_db)).sum(-1)
    gain = torch.maximum(g, torch.tensor(10 ** (min_db / 10)))
    return 10 * torch.log10(gain)


def bc_mixup(source, perm, r: float, fs=8000, wl=0.05):
    """Mixup block (data2vec2.py:536-589) with same_mixup, mixup_prob=1, given r and perm."""
    G1 = compute_gain(source, fs, wl).max(-1)[0].to(source.dtype)
    G2 = G1[perm]
    s2 = source[perm]
    rr = torch.tensor([r], dtype=source.dtype)
    pp = (1 / (1 + 10 ** ((G1 - G2) / 20) * (1 - rr) / rr)).unsqueeze(-1)
    mixed = pp * source + (1 - pp) * s2
    return mixed / torch.sqrt(pp ** 2 + (1 - pp) ** 2)


# ------------------------------------------------------------------------------ weights
def init_params(cfg: OracleConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights with the reference's shapes/key names and init laws (a21):
    Linear N(0,0.02)/zero bias (init_bert_params), except project_features.2 and decoder.proj
    (torch default Linear init, audio.py:199-205); convs kaiming-normal (utils.py:1091); pos-conv
    and decoder convs torch default.  Used for benches and GPU-box tests (no fixtures needed)."""
    return _init_params_impl(cfg, torch.Generator().manual_seed(seed))


def _init_params_impl(cfg: OracleConfig, g) -> Dict[str, torch.Tensor]:
    D, H = cfg.embed_dim, cfg.num_heads
    p: Dict[str, torch.Tensor] = {}

    def normal(*shape, std):
        return (torch.randn(*shape, generator=g) if g is not None else torch.empty(*shape)) * std

    def uniform(*shape, bound):
        return ((torch.rand(*shape, generator=g) if g is not None else torch.empty(*shape)) * 2 - 1) * bound

    def torch_linear(name, fin, fout):
        b = 1 / math.sqrt(fin)
        p[name + ".weight"] = uniform(fout, fin, bound=b)
        p[name + ".bias"] = uniform(fout, bound=b)

    def bert_linear(name, fin, fout):
        p[name + ".weight"] = normal(fout, fin, std=0.02)
        p[name + ".bias"] = torch.zeros(fout)

    def ln(name, dim):
        p[name + ".weight"] = torch.ones(dim)
        p[name + ".bias"] = torch.zeros(dim)

    def block(prefix):
        ln(prefix + "norm1", D)
        bert_linear(prefix + "attn.qkv", D, 3 * D)
        bert_linear(prefix + "attn.proj", D, D)
        ln(prefix + "norm2", D)
        hid = int(D * cfg.mlp_ratio)
        bert_linear(prefix + "mlp.fc1", D, hid)
        bert_linear(prefix + "mlp.fc2", hid, D)

    p[P + "alibi_scale"] = torch.ones(1, 1, H, 1, 1)
    pre = P + "local_encoder.conv_layers."
    cin = 1
    for i, (dim, k, s) in enumerate(cfg.conv_feature_layers):
        if i == 0:
            lo, band = sinc_init(dim, k, cfg.sample_rate)
            p[pre + "0.0.low_hz_"], p[pre + "0.0.band_hz_"] = lo, band
            p[pre + "0.3.p_swish_alpha"] = torch.full((1, dim, 1), 2.0)
            p[pre + "0.3.p_swish_beta"] = torch.zeros(1, dim, 1)
        else:
            p[pre + f"{i}.0.weight"] = normal(dim, cin, k, std=math.sqrt(2.0 / (cin * k)))
        ln(pre + f"{i}.2.1", dim)
        cin = dim
    ln(P + "project_features.1", cin)
    torch_linear(P + "project_features.2", cin, D)
    k = cfg.pos_kernel
    cg = D // cfg.conv_pos_groups
    for i in range(1, cfg.conv_pos_depth + 1):
        b = 1 / math.sqrt(cg * k)
        p[P + f"relative_positional_encoder.{i}.0.weight"] = uniform(D, cg, k, bound=b)
        p[P + f"relative_positional_encoder.{i}.0.bias"] = uniform(D, bound=b)
    for i in range(cfg.prenet_depth):
        block(P + f"context_encoder.blocks.{i}.")
    ln(P + "context_encoder.norm", D)
    din = D
    for i in range(cfg.decoder_layers):
        cg = din // cfg.decoder_groups
        b = 1 / math.sqrt(cg * cfg.decoder_kernel)
        p[P + f"decoder.blocks.{i}.0.weight"] = uniform(cfg.decoder_dim, cg, cfg.decoder_kernel, bound=b)
        p[P + f"decoder.blocks.{i}.0.bias"] = uniform(cfg.decoder_dim, bound=b)
        din = cfg.decoder_dim
    torch_linear(P + "decoder.proj", cfg.decoder_dim, D)
    for i in range(cfg.depth):
        block(f"blocks.{i}.")
    return p


def init_param_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
    """Shapes only (meta tensors): lets tests check the surveyed parameter counts cheaply."""
    with torch.device("meta"):
        return {k: tuple(v.shape) for k, v in _init_params_impl(cfg, None).items()}


def teacher_keys(p: Dict[str, torch.Tensor]) -> List[str]:
    """Keys the EMA teacher holds (data2vec2.py:377-381 strips decoder/local_encoder/project)."""
    drop = (P + "decoder.", P + "local_encoder.", P + "project_features.")
    return [k for k in p if not k.startswith(drop)]


def make_teacher(p):
    return {k: p[k].detach().clone().float() for k in teacher_keys(p)}


BASE_STAR = dict(embed_dim=768, num_heads=12, depth=12, prenet_depth=4, decoder_dim=384,
                 decoder_layers=4, decoder_kernel=7, decoder_groups=16, clone_batch=8,
                 average_top_k_layers=12)
LARGE = dict(embed_dim=1024, num_heads=16, depth=16, prenet_depth=8, decoder_dim=768,
             decoder_layers=4, decoder_kernel=7, decoder_groups=16, clone_batch=12,
             average_top_k_layers=16)
TINY = dict(embed_dim=64, num_heads=4, depth=2, prenet_depth=1, decoder_dim=32, decoder_layers=2,
            decoder_kernel=7, decoder_groups=4, clone_batch=2, conv_pos_groups=4,
            average_top_k_layers=2)
