"""Flat parameter / gradient / optimizer-state storage sized for one MI355X (288 GB HBM3E).

All student parameters live in ONE fp32 buffer ordered by the moment their gradient becomes final in
the backward pass (decoder -> blocks[L-1..0] -> prenet -> pos-conv -> project -> conv stack -> sinc),
so that (a) gradient all-reduce buckets are contiguous slices that can be launched while the rest
of backward still runs, (b) the EMA-teacher's subset (reference data2vec2.py:377-381: everything but
decoder / local_encoder / project_features) is one contiguous range updated by a single kernel, and
(c) Adam is one launch.  Parameter names are the reference's state-dict keys.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .config import Data2VecMultiConfig

P = "modality_encoders.AUDIO."
ALIGN = 64          # floats; also the granularity of the weight-decay mask


@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    offset: int = 0
    decay: bool = True

    @property
    def numel(self) -> int:
        return int(np.prod(self.shape))


def _block_names(prefix: str, D: int, hid: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [
        (prefix + "norm2.weight", (D,)), (prefix + "norm2.bias", (D,)),
        (prefix + "mlp.fc2.weight", (D, hid)), (prefix + "mlp.fc2.bias", (D,)),
        (prefix + "mlp.fc1.weight", (hid, D)), (prefix + "mlp.fc1.bias", (hid,)),
        (prefix + "norm1.weight", (D,)), (prefix + "norm1.bias", (D,)),
        (prefix + "attn.proj.weight", (D, D)), (prefix + "attn.proj.bias", (D,)),
        (prefix + "attn.qkv.weight", (3 * D, D)), (prefix + "attn.qkv.bias", (3 * D,)),
    ]


def build_specs(cfg: Data2VecMultiConfig):
    """-> (specs in flat order, (shared_lo, shared_hi) float offsets of the EMA-shared range)."""
    D, H, hid = cfg.embed_dim, cfg.num_heads, cfg.mlp_dim
    a, dec = cfg.audio, cfg.audio.decoder
    layers = cfg.conv_layers
    names: List[Tuple[str, Tuple[int, ...]]] = []
    # 1. decoder
    names += [(P + "decoder.proj.weight", (D, dec.decoder_dim)), (P + "decoder.proj.bias", (D,))]
    for i in reversed(range(dec.decoder_layers)):
        cin = D if i == 0 else dec.decoder_dim
        names += [(P + f"decoder.blocks.{i}.0.weight", (dec.decoder_dim, cin // dec.decoder_groups, dec.decoder_kernel)),
                  (P + f"decoder.blocks.{i}.0.bias", (dec.decoder_dim,))]
    n_dec = len(names)
    # 2. main blocks, 3. prenet, 4. alibi, 5. pos-conv  (EMA-shared)
    for i in reversed(range(cfg.depth)):
        names += _block_names(f"blocks.{i}.", D, hid)
    for i in reversed(range(a.prenet_depth)):
        names += _block_names(P + f"context_encoder.blocks.{i}.", D, hid)
    names += [(P + "context_encoder.norm.weight", (D,)), (P + "context_encoder.norm.bias", (D,))]
    names += [(P + "alibi_scale", (1, 1, H, 1, 1))]
    k = cfg.pos_kernel
    for i in reversed(range(1, a.conv_pos_depth + 1)):
        names += [(P + f"relative_positional_encoder.{i}.0.weight", (D, D // a.conv_pos_groups, k)),
                  (P + f"relative_positional_encoder.{i}.0.bias", (D,))]
    n_shared_end = len(names)
    # 6. project_features, 7. conv stack, 8. sinc layer
    cf = layers[-1][0]
    names += [(P + "project_features.2.weight", (D, cf)), (P + "project_features.2.bias", (D,)),
              (P + "project_features.1.weight", (cf,)), (P + "project_features.1.bias", (cf,))]
    pre = P + "local_encoder.conv_layers."
    for i in reversed(range(1, len(layers))):
        dim, kk, _ = layers[i]
        names += [(pre + f"{i}.2.1.weight", (dim,)), (pre + f"{i}.2.1.bias", (dim,)),
                  (pre + f"{i}.0.weight", (dim, layers[i - 1][0], kk))]
    c0 = layers[0][0]
    names += [(pre + "0.3.p_swish_alpha", (1, c0, 1)), (pre + "0.3.p_swish_beta", (1, c0, 1)),
              (pre + "0.2.1.weight", (c0,)), (pre + "0.2.1.bias", (c0,)),
              (pre + "0.0.low_hz_", (c0, 1)), (pre + "0.0.band_hz_", (c0, 1))]

    specs, off = [], 0
    bounds = {}
    for idx, (n, shp) in enumerate(names):
        if idx == n_dec:
            bounds["lo"] = off
        if idx == n_shared_end:
            bounds["hi"] = off
        # reference data2vec2.py:318-320 (note low_hz_/band_hz_ are 2-D => decayed)
        nodecay = len(shp) == 1 or n.endswith(".bias") or "alibi_scale" in n or "p_swish" in n
        specs.append(ParamSpec(n, tuple(shp), off, not nodecay))
        off += (int(np.prod(shp)) + ALIGN - 1) // ALIGN * ALIGN
    return specs, (bounds["lo"], bounds["hi"]), off


def bucket_starts(cfg: Data2VecMultiConfig, specs: List[ParamSpec]) -> List[Tuple[str, int]]:
    """(engine grad-ready event, first flat offset) in backward order -- the all-reduce buckets."""
    by = {s.name: s for s in specs}
    out = [("decoder", 0)]
    for i in reversed(range(cfg.depth)):
        out.append((f"block{i}", by[f"blocks.{i}.norm2.weight"].offset))
    for i in reversed(range(cfg.audio.prenet_depth)):
        out.append((f"prenet{i}", by[P + f"context_encoder.blocks.{i}.norm2.weight"].offset))
    out.append(("posconv", by[P + "context_encoder.norm.weight"].offset))
    out.append(("project", by[P + "project_features.2.weight"].offset))
    nl = len(cfg.conv_layers)
    out.append(("frontend", by[P + f"local_encoder.conv_layers.{nl - 1}.2.1.weight"].offset))
    return out


class ParamStore:
    def __init__(self, cfg: Data2VecMultiConfig, device="cuda", with_optimizer=True):
        self.cfg = cfg
        self.device = torch.device(device)
        self.specs, (self.sh_lo, self.sh_hi), self.total = build_specs(cfg)
        self.by_name = {s.name: s for s in self.specs}
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device)
        self.flat = z(self.total)
        self.grad = z(self.total)
        self.teacher = z(self.sh_hi - self.sh_lo)
        self.adam_m = z(self.total) if with_optimizer else None
        self.adam_v = z(self.total) if with_optimizer else None
        mask = np.zeros(self.total // ALIGN, dtype=np.uint8)
        for s in self.specs:
            if s.decay:
                mask[s.offset // ALIGN:(s.offset + s.numel + ALIGN - 1) // ALIGN] = 1
        self.decay_mask = torch.from_numpy(mask).to(self.device)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        self.t: Dict[str, torch.Tensor] = {}
        for s in self.specs:
            self.p[s.name] = self.flat[s.offset:s.offset + s.numel].view(s.shape)
            self.g[s.name] = self.grad[s.offset:s.offset + s.numel].view(s.shape)
            if self.sh_lo <= s.offset < self.sh_hi:
                o = s.offset - self.sh_lo
                self.t[s.name] = self.teacher[o:o + s.numel].view(s.shape)
        # sinc buffers (reference sinc.py:264-276): half Hamming window and time axis
        c0, k0, _ = cfg.conv_layers[0]
        sr = cfg.audio.sample_rate
        n_lin = torch.linspace(0, (k0 / 2) - 1, steps=int(k0 / 2))
        self.sinc_window = (0.53836 - 0.46164 * torch.cos(2 * math.pi * n_lin / k0)).float().to(self.device)
        n = (k0 - 1) / 2.0
        self.sinc_n = (2 * math.pi * torch.arange(-n, 0).view(-1) / sr).float().to(self.device)
        self.sinc_min_band = float(np.ceil(sr / k0).astype(int))       # sinc.py:79
        self.alibi_slopes = torch.tensor(alibi_slopes(cfg.num_heads), dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ init / io
    @torch.no_grad()
    def init_reference(self, seed: int = 0):
        """Reference init laws (a21): init_bert_params on every Linear (N(0,0.02), zero bias;
        data2vec2.py:286-288) then torch-default re-init of project_features.2 and decoder.proj
        (audio.py:199-205, modules.py:119-122); feature-extractor convs kaiming-normal
        (utils.py:1091); grouped convs torch default; LN (1,0); PSwish (2,0) (utils.py:1433-1435);
        alibi_scale 1; SincConv mel init (sinc.py:225-276).  Teacher = copy of the student."""
        g = torch.Generator().manual_seed(seed)
        cfg = self.cfg
        for s in self.specs:
            n, shp = s.name, s.shape
            if n.endswith("low_hz_") or n.endswith("band_hz_"):
                continue
            if "p_swish_alpha" in n:
                v = torch.full(shp, 2.0)
            elif "p_swish_beta" in n:
                v = torch.zeros(shp)
            elif "alibi_scale" in n:
                v = torch.full(shp, cfg.audio.alibi_scale)
            elif "norm" in n or ".2.1." in n or "project_features.1." in n:
                v = torch.ones(shp) if n.endswith("weight") else torch.zeros(shp)
            elif "project_features.2." in n or "decoder.proj." in n:
                fin = self.by_name[n.rsplit(".", 1)[0] + ".weight"].shape[1]
                b = 1 / math.sqrt(fin)
                v = (torch.rand(shp, generator=g) * 2 - 1) * b
            elif "relative_positional_encoder" in n or "decoder.blocks" in n:
                w = self.by_name[n.rsplit(".", 1)[0] + ".weight"].shape
                b = 1 / math.sqrt(w[1] * w[2])
                v = (torch.rand(shp, generator=g) * 2 - 1) * b
            elif "local_encoder" in n and n.endswith("0.weight"):
                v = torch.randn(shp, generator=g) * math.sqrt(2.0 / (shp[1] * shp[2]))
            elif n.endswith(".bias"):
                v = torch.zeros(shp)
            else:
                v = torch.randn(shp, generator=g) * 0.02
            self.p[n].copy_(v)
        c0, k0, _ = cfg.conv_layers[0]
        lo, band = sinc_mel_init(c0, k0, cfg.audio.sample_rate)
        self.p[P + "local_encoder.conv_layers.0.0.low_hz_"].copy_(lo)
        self.p[P + "local_encoder.conv_layers.0.0.band_hz_"].copy_(band)
        self.sync_teacher_from_student()

    @torch.no_grad()
    def sync_teacher_from_student(self):
        self.teacher.copy_(self.flat[self.sh_lo:self.sh_hi])

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor], ema: Optional[Dict[str, torch.Tensor]] = None, strict=True):
        """sd: reference-named tensors; ema: the reference's '_ema' dict (fp32 teacher params)."""
        sd = dict(sd)
        if ema is None and "_ema" in sd:
            ema = sd.pop("_ema")
        missing = [s.name for s in self.specs if s.name not in sd]
        if strict and missing:
            raise KeyError(f"missing parameters: {missing[:5]}...")
        for s in self.specs:
            if s.name in sd:
                v = sd[s.name]
                if s.name.endswith("alibi_scale") and v.dim() == 4:     # base.py:152-157 upgrade
                    v = v.unsqueeze(0)
                self.p[s.name].copy_(v.to(self.device, torch.float32).view(s.shape))
        if ema is not None:
            for n, tv in self.t.items():
                if n in ema:
                    tv.copy_(ema[n].to(self.device, torch.float32).view(tv.shape))
                elif strict:
                    raise KeyError(f"missing teacher parameter {n}")
        else:
            self.sync_teacher_from_student()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Reference key names + '_ema' (data2vec2.py:412-418)."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)      # the teacher is updated on the trainer's side stream
        order = sorted(self.specs, key=lambda s: s.name)
        sd = {s.name: self.p[s.name].detach().clone() for s in order}
        sd["_ema"] = {n: v.detach().clone() for n, v in sorted(self.t.items())}
        return sd

    def n_params(self) -> int:
        return sum(s.numel for s in self.specs)


def alibi_slopes(n: int) -> List[float]:
    """ALiBi head slopes, reference nn/modalities/base.py:559-576."""
    def pow2(m):
        start = 2 ** (-(2 ** -(math.log2(m) - 3)))
        return [start * start ** i for i in range(m)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + alibi_slopes(2 * c)[0::2][: n - c]


def sinc_mel_init(out_channels: int, kernel_size: int, sample_rate: int, min_low_hz: float = 50.0):
    """Mel-spaced SincConv init, reference nn/sinc.py:225-284."""
    min_band_hz = int(np.ceil(sample_rate / kernel_size))
    high_hz = sample_rate / 2 - (min_low_hz + min_band_hz)
    to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
    mel = torch.linspace(to_mel(min_low_hz), to_mel(high_hz), out_channels + 1)
    hz = 700 * (10 ** (mel / 2595) - 1)
    return hz[:-1].unsqueeze(1).clone(), (hz[1:] - hz[:-1]).unsqueeze(1).clone()


# ======================================================================================== packs
def _pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def _padk(c: int, dtype) -> int:
    """Tap width in the packed operand: bf16 pads >=32-wide taps up to a multiple of 64 so that a 64-wide
    K tile never straddles two taps (the fast implicit-GEMM path); the padding reads as zero."""
    if dtype in (torch.bfloat16, torch.float16) and c >= 32 and c % 64:
        return (c + 63) // 64 * 64
    return _pad8(c)


class Packs:
    """MFMA operand images of the weights in the activation dtype: K-contiguous forward operands,
    transposed / tap-flipped operands for the input gradients.  Rebuilt by a few pack kernels after
    each optimizer step (student) / EMA step (teacher); in fp32 mode Linear forward operands alias
    the flat parameter buffer."""

    def __init__(self, store: ParamStore, dtype: torch.dtype, teacher: bool, need_grad: bool, x3: bool = False):
        # x3 (fp32 storage only): the Linear operands also get bf16 split images (ops.x3_register) -- the bf16x3 parity mode
        assert not x3 or dtype == torch.float32
        self.x3 = x3
        self.store, self.dtype, self.teacher, self.need_grad = store, dtype, teacher, need_grad
        self.src = store.t if teacher else store.p
        cfg = store.cfg
        self.w: Dict[str, torch.Tensor] = {}       # forward operands
        self.wt: Dict[str, torch.Tensor] = {}      # input-gradient operands
        self.gw: Dict[str, torch.Tensor] = {}      # fp32 packed weight-gradient scratch (convs)
        self.meta: Dict[str, dict] = {}
        dev = store.device
        D = cfg.embed_dim
        for name, src in self.src.items():
            if not name.endswith(".weight") or src.dim() < 2:
                continue
            shp = src.shape
            if src.dim() == 2:                      # Linear (N, K)
                N, K = shp
                self.meta[name] = dict(kind="linear", N=N, K=K)
                if dtype == torch.float32:
                    self.w[name] = src
                else:
                    self.w[name] = torch.empty(N, K, dtype=dtype, device=dev)
                if need_grad:
                    self.wt[name] = torch.empty(K, N, dtype=dtype, device=dev)
            else:                                   # conv (Cout, Cg, k)
                Cout, Cg, k = shp
                if "local_encoder" in name:
                    G = 1
                elif "relative_positional_encoder" in name:
                    G = cfg.audio.conv_pos_groups
                else:
                    G = cfg.audio.decoder.decoder_groups
                N = Cout // G
                # tap padding to 64 only for the grouped convs (dense feature-extractor convs are 128/512 wide); the bf16x3 mode
                # runs the bf16 kernels over split images of these operands and takes the bf16 padding
                pdt = torch.bfloat16 if x3 else dtype
                Cgp = _padk(Cg, pdt) if G > 1 else _pad8(Cg)
                Np = _padk(N, pdt) if G > 1 else N
                li = int(name.split("conv_layers.")[1].split(".")[0]) if "local_encoder" in name else None
                stride = cfg.conv_layers[li][2] if li is not None else 1
                m = dict(kind="conv", G=G, N=N, Cg=Cg, Cgp=Cgp, Np=Np, k=k, stride=stride)
                self.meta[name] = m
                self.w[name] = torch.empty(G, N, k * Cgp, dtype=dtype, device=dev)
                if need_grad:
                    self.gw[name] = torch.zeros(G, N, k * Cgp, dtype=torch.float32, device=dev)
                    if G == 1 and stride > 1:
                        self.wt[name] = torch.empty(k * Cgp, Cout, dtype=dtype, device=dev)     # mode 2
                    else:
                        assert Cg % 8 == 0, "grouped / stride-1 convs need Cg % 8 == 0"
                        self.wt[name] = torch.empty(G, Cg, k * Np, dtype=dtype, device=dev)    # mode 1

    def _descriptors(self) -> torch.Tensor:
        rows = []
        for name, m in self.meta.items():
            src = self.src[name]
            if m["kind"] == "linear":
                N, K = m["N"], m["K"]
                if self.dtype != torch.float32:
                    rows.append([src.data_ptr(), self.w[name].data_ptr(), 1, N, K, 1, K, N, 0, N * K])
                if self.need_grad:
                    rows.append([src.data_ptr(), self.wt[name].data_ptr(), 1, N, K, 1, K, N, 1, N * K])
            else:
                G, N, Cg, k, Cgp, Np = m["G"], m["N"], m["Cg"], m["k"], m["Cgp"], m["Np"]
                rows.append([src.data_ptr(), self.w[name].data_ptr(), G, N, Cg, k, Cgp, N, 0, self.w[name].numel()])
                if self.need_grad:
                    if G == 1 and m["stride"] > 1:
                        rows.append([src.data_ptr(), self.wt[name].data_ptr(), 1, N, Cg, k, Cgp, N, 2, self.wt[name].numel()])
                    else:
                        rows.append([src.data_ptr(), self.wt[name].data_ptr(), G, N, Cg, k, Cg, Np, 1, self.wt[name].numel()])
        return torch.tensor(rows, dtype=torch.int64, device=self.store.device)

    @torch.no_grad()
    def refresh(self):
        """one launch for all operand images (the descriptor table is built once: buffers never move)"""
        import os
        if os.environ.get("A2V_PACK_SINGLE") == "1":        # A/B switch: one launch per weight
            return self.refresh_one_by_one()
        if getattr(self, "_desc", None) is None:
            self._desc = self._descriptors()
        if self._desc.shape[0]:
            ops.pack_weight_batch(self._desc, self.dtype)
        if self.x3:
            self._register_x3()

    def _register_x3(self):
        """split images [hi | hi | lo] of every operand that is the right-hand side of a plain Linear product: the Linears' forward
        operands (the fp32 parameters themselves), their transposes for the input gradients, and the (k*C, Cout) images of the
        strided feature convs' input gradients"""
        for name, m in self.meta.items():
            if m["kind"] == "linear":
                ops.x3_register(self.w[name])
                if self.need_grad:
                    ops.x3_register(self.wt[name])
            else:
                ops.x3_register_conv_pack(self.w[name], m["k"], m["Cg"])
                if self.need_grad and m["G"] == 1 and m["stride"] > 1:
                    ops.x3_register(self.wt[name])
                elif self.need_grad:
                    ops.x3_register_conv_pack(self.wt[name], m["k"], m["N"])

    @torch.no_grad()
    def refresh_one_by_one(self):
        for name, m in self.meta.items():
            src = self.src[name]
            if m["kind"] == "linear":
                if self.dtype != torch.float32:
                    ops.pack_weight(src, self.w[name], G=1, N=m["N"], Cg=m["K"], k=1, mode=0)
                if self.need_grad:
                    ops.pack_weight(src, self.wt[name], G=1, N=m["N"], Cg=m["K"], k=1, mode=1)
            else:
                ops.pack_weight(src, self.w[name], G=m["G"], N=m["N"], Cg=m["Cg"], k=m["k"], Cgp=m["Cgp"], mode=0)
                if self.need_grad:
                    if m["G"] == 1 and m["stride"] > 1:
                        ops.pack_weight(src, self.wt[name], G=1, N=m["N"], Cg=m["Cg"], k=m["k"], Cgp=m["Cgp"], mode=2)
                    else:
                        ops.pack_weight(src, self.wt[name], G=m["G"], N=m["N"], Cg=m["Cg"], k=m["k"], Np=m["Np"], mode=1)
        if self.x3:
            self._register_x3()
