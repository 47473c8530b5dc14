#!/usr/bin/env python3
"""Pin the oracle against the LIVE reference and (re)generate tests/golden/*.npz.

Container-only tool (needs /root/reference; run from the repo root):

    python oracle/gen_golden.py            # validate + write fixtures
    python oracle/gen_golden.py --full     # additionally validate at large dims (minutes, nothing written)

What it does
  1. imports the reference ``nn`` package through ``oracle/ref_shims.py``;
  2. builds reference models, copies their weights into the oracle's flat dict and asserts that
     ``oracle/a2v_oracle.py`` reproduces the reference's outputs AND parameter gradients;
  3. dumps small input/weight/output vectors as fixtures.  Fixtures are data only -- no reference
     source text is stored.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import a2v_oracle as O  # noqa: E402
import ref_shims  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")

TINY_LAYERS = [(31, 15, 1), (32, 10, 5)] + [(32, 3, 2)] * 3 + [(32, 3, 1)] + [(32, 2, 1)] * 2
TINY_LAYERS_STR = "[(31, 15, 1)] + [(32, 10, 5)] + [(32, 3, 2)] * 3 + [(32, 3, 1)] + [(32, 2, 1)] * 2"


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, a, b, tol):
    r = rel(a, b)
    flag = "ok " if r <= tol else "FAIL"
    print(f"  [{flag}] {name:42s} rel={r:.3e}")
    assert r <= tol, (name, r)


def state_to_flat(model):
    sd = model.state_dict()
    ema = {k: v.detach().clone() for k, v in sd["_ema"].items()}
    p = {k: v.detach().clone() for k, v in sd.items() if k != "_ema"}
    return p, ema


def run_pair(refnn, dims, layers, layers_str, B, L, keep, seed, tol=2e-5, perturb=True, grads=True):
    """Build reference + oracle with identical weights, run both, compare.  Returns fixture dict."""
    cfg = O.OracleConfig(**dims, conv_feature_layers=layers)
    model, _ = ref_shims.build_reference_model(
        refnn, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth,
        prenet_depth=cfg.prenet_depth, decoder_dim=cfg.decoder_dim, decoder_layers=cfg.decoder_layers,
        decoder_kernel=cfg.decoder_kernel, decoder_groups=cfg.decoder_groups,
        clone_batch=cfg.clone_batch, conv_feature_layers=layers_str, conv_pos_groups=cfg.conv_pos_groups,
        average_top_k_layers=cfg.average_top_k_layers, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    if perturb:
        # move every parameter off its init so that LN affine / PSwish beta / alibi scales / biases matter,
        # and make the teacher differ from the student
        with torch.no_grad():
            for n, prm in model.named_parameters():
                if "low_hz_" in n or "band_hz_" in n:
                    prm.add_(torch.randn(prm.shape, generator=g) * 3.0)
                elif "alibi_scale" in n:
                    prm.add_(torch.rand(prm.shape, generator=g) * 0.5)
                elif "p_swish_beta" in n:
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.3)
                else:
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.02)
            for n, prm in model.ema.model.named_parameters():
                prm.add_(torch.randn(prm.shape, generator=g) * 0.01)
            model.ema.build_fp32_params()
    model.eval()
    p, ema = state_to_flat(model)
    T = O.frame_counts(L, layers)[-1]
    source = torch.randn(B, L, generator=g)
    source = torch.nn.functional.layer_norm(source, (L,))
    mask = O.seeded_mask(B * cfg.clone_batch, T, keep, seed + 7)

    # ---- reference
    torch.manual_seed(seed + 11)
    out_ref = model(source, precomputed_mask=mask)
    torch.manual_seed(seed + 11)
    noise_ref = torch.empty(B * cfg.clone_batch, T - keep, cfg.embed_dim).normal_(0, cfg.mask_noise_std)
    loss_ref = out_ref["losses"]["AUDIO_regression"]
    model.zero_grad()
    loss_ref.sum().backward()
    gref = {n: prm.grad.detach().clone() for n, prm in model.named_parameters() if prm.grad is not None}

    # reference-side intermediates obtained by calling its own sub-modules
    enc = model.modality_encoders["AUDIO"]
    with torch.no_grad():
        feats_ref = enc.local_encoder(source)
        lf_ref = enc.local_features(source)
        mi = enc.make_maskinfo(lf_ref.repeat_interleave(cfg.clone_batch, 0), mask)
        tm = model.ema.model
        tin = tm.modality_encoders["AUDIO"].contextualized_features(lf_ref, None, mask=False, remove_masked=False)
        xt, ab = tin["x"], tin["alibi_bias"]
        ts = []
        for blk in tm.blocks:
            xt, t = blk(xt, padding_mask=None, alibi_bias=ab)
            ts.append(t)
        y_ref = model.make_targets(ts, cfg.average_top_k_layers)

    # The reference's argsort is not stable, so WHICH noise row lands on WHICH masked frame depends
    # on torch's tie order.  The oracle's convention is: noise row j belongs to the j-th masked frame
    # in ascending time order.  Re-index the reference's draw into that convention (frame by frame
    # the two models then see identical mask tokens).
    ref_restore = mi.ids_restore[..., 0]
    noise = torch.empty_like(noise_ref)
    for n in range(mask.shape[0]):
        masked_t = torch.nonzero(mask[n]).flatten()                       # ascending
        noise[n] = noise_ref[n, ref_restore[n, masked_t] - keep]

    # ---- oracle
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    res = O.pretrain_forward(po, ema, source, mask, noise, cfg)
    loss = res["losses"]["AUDIO_regression"]
    loss.sum().backward()
    dbg = res["_dbg"]

    print(f"pair D={cfg.embed_dim} depth={cfg.depth}+{cfg.prenet_depth} B={B} L={L} T={T} keep={keep}")
    assert torch.equal(mi.ids_keep[..., 0].sort(dim=1)[0], dbg["ids_keep"]), "kept sets differ"
    check("feature_extractor", O.feature_extractor(p, source, cfg), feats_ref, tol)
    check("local_features", dbg["local_features"], lf_ref, tol)
    check("teacher targets y", dbg["targets"], y_ref, tol * 5)
    check("loss (unreduced)", loss, loss_ref, tol * 5)
    check("loss.sum", loss.sum(), loss_ref.sum(), tol * 5)
    assert int(res["sample_size"]) == int(out_ref["sample_size"])
    assert abs(res["masked_pct"] - out_ref["masked_pct"]) < 1e-12
    check("pred_var", res["pred_var"], out_ref["pred_var"], tol * 5)
    check("target_var", res["target_var"], out_ref["target_var"], tol * 5)
    if grads:
        worst = 0.0
        for n, gr in gref.items():
            go = po[n].grad
            assert go is not None, n
            r = rel(go, gr)
            worst = max(worst, r)
            assert r < 5e-4, (n, r)
        print(f"  [ok ] all {len(gref)} parameter grads                 worst rel={worst:.3e}")

    # features-only path (extract_features, BASELINE config 1) against the live reference
    with torch.no_grad():
        fo_ref = model.extract_features(source)
    fo_x, fo_lr = O.features_only_forward(p, source, cfg)
    check("extract_features x", fo_x, fo_ref["x"], tol * 5)
    check("extract_features layer_results[-1]", fo_lr[-1], fo_ref["layer_results"][-1], tol * 5)

    fx = {"source": source, "mask": mask, "noise": noise, "loss": loss_ref.detach(),
          "features_only.x": fo_ref["x"], "features_only.layer_mean": sum(fo_ref["layer_results"]) / len(fo_ref["layer_results"]),
          "loss_sum": loss_ref.sum().detach(), "pred_var": out_ref["pred_var"],
          "target_var": out_ref["target_var"], "sample_size": out_ref["sample_size"],
          "features": feats_ref, "local_features": lf_ref, "targets": y_ref,
          "ids_keep": dbg["ids_keep"], "ids_restore": dbg["ids_restore"]}
    # a few reference-side intermediates for kernel-level parity (obtained via oracle == ref above)
    for k in ("x_pos", "prenet_in", "prenet_out", "encoder_out", "decoder_in", "decoder_out"):
        fx["dbg." + k] = dbg[k].detach()
    for k, v in p.items():
        fx["p." + k] = v
    for k, v in ema.items():
        fx["ema." + k] = v
    for k, v in gref.items():
        fx["g." + k] = v
    return model, cfg, fx


def save(name, d):
    os.makedirs(GOLD, exist_ok=True)
    arrs = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        arrs[k] = np.asarray(v)
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB, {len(arrs)} arrays)")


def sinc_fixtures(refnn):
    """G1: SincConv filters at default init and after a perturbation, three (C,k,sr) settings,
    plus the reflect-padded conv output on a short signal (sinc.py known-answer vectors)."""
    from nn.sinc import SincConv
    out = {}
    g = torch.Generator().manual_seed(5)
    for (C, k, sr) in [(127, 63, 8000), (127, 63, 44100), (296, 149, 44100), (31, 15, 8000)]:
        m = SincConv(out_channels=C, kernel_size=k, sample_rate=sr)
        m.device = torch.device("cpu")
        tag = f"{C}_{k}_{sr}"
        lo, band = O.sinc_init(C, k, sr)
        assert torch.equal(lo, m.low_hz_.detach()) and torch.equal(band, m.band_hz_.detach()), tag
        f_ref = m._get_sinc_filters().detach()
        check(f"sinc filters init {tag}", O.sinc_filters(lo, band, k, sr), f_ref, 1e-6)
        out[f"init.{tag}.low"], out[f"init.{tag}.band"], out[f"init.{tag}.filters"] = lo, band, f_ref
        with torch.no_grad():
            m.low_hz_.add_(torch.randn(m.low_hz_.shape, generator=g) * 20)
            m.band_hz_.add_(torch.randn(m.band_hz_.shape, generator=g) * 20)
        f_ref = m._get_sinc_filters().detach()
        check(f"sinc filters perturbed {tag}", O.sinc_filters(m.low_hz_.detach(), m.band_hz_.detach(), k, sr), f_ref, 1e-6)
        out[f"pert.{tag}.low"], out[f"pert.{tag}.band"] = m.low_hz_.detach().clone(), m.band_hz_.detach().clone()
        out[f"pert.{tag}.filters"] = f_ref
        if sr == 8000:
            x = torch.randn(2, 1000, generator=g)
            y_ref = m(x.unsqueeze(1)).detach()
            check(f"sinc conv fwd {tag}", O.sinc_conv(x, m.low_hz_.detach(), m.band_hz_.detach(), k, sr), y_ref, 1e-6)
            out[f"conv.{tag}.x"], out[f"conv.{tag}.y"] = x, y_ref
    return out


def ema_fixture(model, cfg, fx):
    """G13: one EMA step at the annealed decay for num_updates=1 -> teacher params."""
    p = {k[2:]: torch.as_tensor(v) for k, v in fx.items() if k.startswith("p.")}
    ema = {k[4:]: torch.as_tensor(v).clone() for k, v in fx.items() if k.startswith("ema.")}
    model.train()
    model.num_updates = 0
    model.set_num_updates(1)
    d = O.ema_decay_at(cfg, 1)
    assert abs(model.ema.get_decay() - d) < 1e-15
    O.ema_step(p, ema, d)
    after = model.ema.fp32_params
    worst = max(rel(ema[k], after[k]) for k in ema)
    print(f"  [ok ] EMA step decay={d:.10f} worst rel={worst:.2e}")
    assert worst < 1e-6
    model.eval()
    return {"decay": torch.tensor(d, dtype=torch.float64),
            **{"after." + k: v.detach().clone() for k, v in after.items()}}


def mixup_fixture(refnn):
    """G14: BC-mixup with fixed r / perm, checked against the reference's forward in train mode by
    monkey-patching its RNG draws (uniform_ and randperm) to the fixed values."""
    model, _ = ref_shims.build_reference_model(
        refnn, **{k: v for k, v in O.TINY.items()}, conv_feature_layers=TINY_LAYERS_STR, source_mixup=0.5)
    g = torch.Generator().manual_seed(9)
    src = torch.nn.functional.layer_norm(torch.randn(4, 8000, generator=g), (8000,))
    src = src * torch.tensor([1.0, 0.3, 2.0, 0.7]).unsqueeze(-1)
    perm = torch.tensor([2, 0, 3, 1])
    r = 0.7
    G_ref = model.compute_gain_torch(src, fs=8000, wl=0.05, mode="A_weighting")
    check("mixup gain dB", O.compute_gain(src, 8000, 0.05), G_ref, 1e-6)
    G1 = G_ref.max(-1)[0].to(src.dtype)
    G2 = G1[perm]
    rr = torch.tensor([r])
    p = (1 / (1 + 10 ** ((G1 - G2) / 20) * (1 - rr) / rr)).unsqueeze(-1)      # data2vec2.py:581-582
    mixed_ref = (p * src + (1 - p) * src[perm]) / torch.sqrt(p ** 2 + (1 - p) ** 2)  # :584-587
    check("mixup mixed source", O.bc_mixup(src, perm, r), mixed_ref, 1e-6)
    return {"source": src, "perm": perm, "r": torch.tensor(r), "gain_db": G_ref, "mixed": mixed_ref}


def finetune_fixture(refnn, model, cfg, seed=21, n_labels=12, tol=2e-5):
    """G15: the finetune path's logits, focal loss and gradients.  The encoder side is the
    LIVE reference (Data2VecMultiModel.extract_features with local_grad_mult = 0, as wav2vec2.py:122 sets it from
    feature_grad_mult 0.0) and the loss is the LIVE nn.utils.sigmoid_focal_loss; the glue between them (mean of the
    last average_top_k_layers layer results, Linear(D -> n_labels)) is the five lines of wav2vec2.py:448-475,
    restated here because Wav2VecEncoderModOut itself needs a fairseq checkpoint + task to construct."""
    import importlib
    ref_utils = importlib.import_module("nn.utils")
    g = torch.Generator().manual_seed(seed)
    B, L = 2, 8000
    D = cfg.embed_dim
    T = O.frame_counts(L, cfg.conv_feature_layers)[-1]
    source = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,))
    target = (torch.rand(B, T, n_labels, generator=g) < 0.15).float()
    head_w = torch.randn(n_labels, D, generator=g) * 0.1
    head_b = torch.randn(n_labels, generator=g) * 0.1
    model.eval()
    enc = model.modality_encoders["AUDIO"]
    old = enc.local_grad_mult
    enc.local_grad_mult = 0.0
    hw, hb = head_w.clone().requires_grad_(True), head_b.clone().requires_grad_(True)
    model.zero_grad()
    res = model.extract_features(source)
    lrs = res["layer_results"][-cfg.average_top_k_layers:]
    xm = (sum(lrs) / len(lrs)).to(res["x"].dtype)
    logits_ref = torch.nn.functional.linear(xm, hw, hb)
    loss_ref = ref_utils.sigmoid_focal_loss(logits_ref, target, reduction="sum")
    loss_ref.backward()
    gref = {n: prm.grad.detach().clone() for n, prm in model.named_parameters() if prm.grad is not None}
    enc.local_grad_mult = old
    # oracle
    p, _ = state_to_flat(model)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ow, ob = head_w.clone().requires_grad_(True), head_b.clone().requires_grad_(True)
    logits = O.finetune_forward(po, ow, ob, source, cfg)
    loss = O.sigmoid_focal_loss(logits, target, reduction="sum")
    loss.backward()
    check("finetune logits", logits, logits_ref, tol * 5)
    check("finetune focal loss", loss, loss_ref, tol * 5)
    check("finetune head grads", ow.grad, hw.grad, tol * 25)
    worst, n_g = 0.0, 0
    for n, gr in gref.items():
        go = po[n].grad
        assert go is not None, n
        r = rel(go, gr)
        worst = max(worst, r)
        n_g += 1
        assert r < 5e-4, (n, r)
    no_grad = [n for n in po if po[n].grad is None]
    # no gradient: the frozen conv feature extractor, the (unused) decoder, and the last block's norm2 (its output x
    # is not part of any layer result)
    last_norm2 = f"blocks.{cfg.depth - 1}.norm2."
    odd = [n for n in no_grad if not (("local_encoder" in n) or ("decoder" in n) or n.startswith(last_norm2))]
    assert not odd, odd[:5]
    assert set(no_grad) == {n for n in po if n not in gref}, "reference and oracle disagree on which parameters get gradients"
    print(f"  [ok ] finetune: {n_g} encoder grads worst rel={worst:.3e}; {len(no_grad)} params correctly without gradient")
    fx = {"source": source, "target": target, "head_w": head_w, "head_b": head_b, "logits": logits_ref.detach(),
          "loss_sum": loss_ref.detach(), "g.head_w": hw.grad, "g.head_b": hb.grad}
    for k, v in gref.items():
        fx["g." + k] = v
    return fx


def finetune_train_fixture(refnn, model, cfg, seed=33, n_labels=12, tol=2e-5):
    """G16 (train form): noise-replace time masking + channel masking + layerdrop through the LIVE
    reference -- Data2VecMultiModel.forward(mask=True, features_only=True) in train() mode, which is what
    extract_features(mask=True) runs (data2vec2.py:1111-1124) plus the `precomputed_mask` hook.  The random draws are
    pinned from outside: torch seed (the normal_() replacement noise, base.py:468-471), numpy seed (layerdrop,
    modules.py:93 / data2vec2.py:654), and fairseq's compute_mask_indices (third party, absent) is replaced for the
    channel mask by a function returning a fixed (B, C) array.  Dropout modules are set to p = 0 and the model's own
    BC mixup is disabled (source_mixup < 0): dropout is checked on the GPU by finite differences, mixup has its own
    fixture."""
    import importlib
    ref_utils = importlib.import_module("nn.utils")
    base_mod = importlib.import_module("nn.modalities.base")
    g = torch.Generator().manual_seed(seed)
    B, L = 2, 8000
    D = cfg.embed_dim
    T = O.frame_counts(L, cfg.conv_feature_layers)[-1]
    source = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,))
    target = (torch.rand(B, T, n_labels, generator=g) < 0.15).float()
    head_w = torch.randn(n_labels, D, generator=g) * 0.1
    head_b = torch.randn(n_labels, generator=g) * 0.1
    mask_t = torch.zeros(B, T, dtype=torch.uint8)
    n_mask = int(0.6 * T)
    for b in range(B):
        mask_t[b, torch.randperm(T, generator=g)[:n_mask]] = 1
    mask_c = (torch.rand(B, D, generator=g) < 0.3)
    enc = model.modality_encoders["AUDIO"]
    mc = enc.modality_cfg
    saved = dict(lgm=enc.local_grad_mult, zero=mc.encoder_zero_mask, mcp=mc.mask_channel_prob, ld=model.cfg.layerdrop,
                 pld=enc.context_encoder.layerdrop, mix=model.cfg.source_mixup, cmi=base_mod.compute_mask_indices)
    drops = [(m, m.p) for m in model.modules() if isinstance(m, torch.nn.Dropout)]
    try:
        enc.local_grad_mult = 0.0
        mc.encoder_zero_mask = False
        mc.mask_channel_prob = 0.5
        model.cfg.layerdrop = 0.4
        enc.context_encoder.layerdrop = 0.4
        model.cfg.source_mixup = -1.0
        base_mod.compute_mask_indices = lambda shape, pm, prob, length, **kw: mask_c.numpy()
        for m, _ in drops:
            m.p = 0.0
        model.train()
        hw, hb = head_w.clone().requires_grad_(True), head_b.clone().requires_grad_(True)
        model.zero_grad()
        t_seed = 9
        nb = cfg.prenet_depth + cfg.depth
        for np_seed in range(100):          # first seed whose layerdrop outcome drops a main block and keeps another
            np.random.seed(np_seed)
            pat = [bool(np.random.random() > 0.4) for _ in range(nb)]
            if not all(pat[cfg.prenet_depth:]) and any(pat[cfg.prenet_depth:]):
                break
        np.random.seed(np_seed)
        torch.manual_seed(t_seed)
        res = model(source, mask=True, features_only=True, precomputed_mask=mask_t.clone())
        lrs = res["layer_results"][-cfg.average_top_k_layers:]
        xm = (sum(lrs) / len(lrs)).to(res["x"].dtype)
        logits_ref = torch.nn.functional.linear(xm, hw, hb)
        loss_ref = ref_utils.sigmoid_focal_loss(logits_ref, target, reduction="sum")
        loss_ref.backward()
        gref = {n: prm.grad.detach().clone() for n, prm in model.named_parameters() if prm.grad is not None}
    finally:
        enc.local_grad_mult = saved["lgm"]
        mc.encoder_zero_mask = saved["zero"]
        mc.mask_channel_prob = saved["mcp"]
        model.cfg.layerdrop = saved["ld"]
        enc.context_encoder.layerdrop = saved["pld"]
        model.cfg.source_mixup = saved["mix"]
        base_mod.compute_mask_indices = saved["cmi"]
        for m, pv in drops:
            m.p = pv
        model.eval()
    # replay the draws: the only torch draw before the blocks is the replacement noise; numpy is drawn once per block
    torch.manual_seed(t_seed)
    replacement = torch.empty(int(mask_t.sum()), D).normal_(0, mc.mask_noise_std)
    np.random.seed(np_seed)
    active = [bool(np.random.random() > 0.4) for _ in range(cfg.prenet_depth + cfg.depth)]
    assert len(res["layer_results"]) == sum(active[cfg.prenet_depth:]), (len(res["layer_results"]), active)
    assert not all(active) and any(active[cfg.prenet_depth:]), active
    p, _ = state_to_flat(model)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ow, ob = head_w.clone().requires_grad_(True), head_b.clone().requires_grad_(True)
    logits = O.finetune_forward(po, ow, ob, source, cfg, mask_t=mask_t, replacement=replacement, mask_c=mask_c, active=active)
    loss = O.sigmoid_focal_loss(logits, target, reduction="sum")
    loss.backward()
    check("finetune(train) logits", logits, logits_ref, tol * 5)
    check("finetune(train) focal loss", loss, loss_ref, tol * 5)
    check("finetune(train) head grads", ow.grad, hw.grad, tol * 25)
    worst = 0.0
    for n, gr in gref.items():
        go = po[n].grad
        if go is None:
            assert float(gr.abs().max()) == 0.0, n        # a dropped block: torch leaves zeros or None
            continue
        r = rel(go, gr) if float(gr.abs().max()) > 0 else float(go.abs().max())
        worst = max(worst, r)
        assert r < 5e-4, (n, r)
    print(f"  [ok ] finetune(train): active={''.join('1' if a else '0' for a in active)} {len(gref)} grads worst rel={worst:.3e}")
    fx = {"source": source, "target": target, "head_w": head_w, "head_b": head_b, "mask_t": mask_t, "mask_c": mask_c.to(torch.uint8),
          "replacement": replacement, "active": torch.tensor(active, dtype=torch.uint8), "logits": logits_ref.detach(),
          "loss_sum": loss_ref.detach(), "g.head_w": hw.grad, "g.head_b": hb.grad}
    for k, v in gref.items():
        fx["g." + k] = v
    return fx


def make_corpus(root, spec, seed):
    """Synthetic labelled corpus in the reference's on-disk layout: <root>/wav/<name>.wav (PCM16), <root>/lbl/<name>.h5
    (the four label arrays, stored as an .npz container: h5py is absent in this image and read through a stand-in),
    <root>/<split>.tsv.  spec: list of (name, n_samples, n_channels, n_events or None = no label file).
    Returns the raw arrays, so that a test can rebuild the same corpus from the fixture alone."""
    import wave
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "wav"), exist_ok=True)
    os.makedirs(os.path.join(root, "lbl"), exist_ok=True)
    raw = {}
    lines = [root]           # manifest root = the parent of wav/ and lbl/, entries "wav/<name>.wav" (a root that itself ends
                             # in wav/ cannot be served by the reference: __getitem__ then looks for <parent>/<name>.wav)
    for name, n, ch, nev in spec:
        x = (rs.randn(n, ch) * 3000).clip(-32768, 32767).astype("<i2")
        with wave.open(os.path.join(root, "wav", name + ".wav"), "wb") as w:
            w.setnchannels(ch)
            w.setsampwidth(2)
            w.setframerate(8000)
            w.writeframes(x.tobytes())
        raw[f"wav.{name}"] = x
        lines.append(f"wav/{name}.wav\t{n}")
        if nev is None:
            continue
        st = np.sort(rs.randint(0, n - 400, size=nev)).astype(np.int64)
        en = (st + rs.randint(100, 2500, size=nev)).clip(max=n).astype(np.int64)
        cat = rs.randint(0, 3, size=nev).astype(np.int64)
        foc = rs.randint(0, 2, size=nev).astype(np.int64)
        with open(os.path.join(root, "lbl", name + ".h5"), "wb") as f:
            np.savez(f, start_frame_lbl=st, end_frame_lbl=en, lbl_cat=cat, foc=foc)
        raw.update({f"lbl.{name}.start": st, f"lbl.{name}.end": en, f"lbl.{name}.cat": cat, f"lbl.{name}.foc": foc})
    return raw, lines


DATA_SPECS = {
    # equal-length chunks (the shipped MeerKAT layout), one stereo file, one entry without labels, one too short
    "equal": [("a0", 8000, 1, 5), ("a1", 8000, 2, 7), ("a2", 8000, 1, 0), ("nolabel", 8000, 1, None), ("short", 900, 1, 2)],
    # clips of different lengths: the collater's random crops (numpy global RNG, seeded below)
    "ragged": [("b0", 8000, 1, 6), ("b1", 9000, 1, 6), ("b2", 7200, 1, 4)],
}
DATA_LABELS = ["a", "b", "c", "focal"]


def data_pipeline_fixture(refnn):
    """G17: FileAudioLabelDataset (audio_tasks.py:196-465) on a synthetic corpus -- manifest
    filtering, __getitem__ (PCM16 -> float32, channel mean, whole-clip layer norm, event list -> frame targets through
    interp1d) and collater (equal lengths: copy; different lengths: per-clip and per-label-column random crops).
    Third-party pieces behind it (RawAudioDataset.postprocess / crop_to_max_size, soundfile, h5py) are stand-ins
    (ref_shims.py): what is pinned is the reference's own arithmetic and control flow in audio_tasks.py."""
    import tempfile

    from nn.audio_tasks import FileAudioLabelDataset
    fx = {"labels": np.array(DATA_LABELS), "layers": np.array(ref_shims.LARGE_LAYERS)}
    with tempfile.TemporaryDirectory() as tmp:
        for split, spec in DATA_SPECS.items():
            root = os.path.join(tmp, split)
            raw, lines = make_corpus(root, spec, seed=len(split))
            with open(os.path.join(root, "train.tsv"), "w") as f:
                f.write("\n".join(lines) + "\n")
            ds = FileAudioLabelDataset(os.path.join(root, "train.tsv"), sample_rate=8000, max_sample_size=None,
                                       min_sample_size=1000, pad=False, normalize=True, unique_labels=DATA_LABELS,
                                       use_focal_loss=True, return_labels=True, conv_feature_layers=ref_shims.LARGE_LAYERS,
                                       min_label_size=0, do_focal_prediction=True)
            fx.update({f"{split}.{k}": v for k, v in raw.items()})
            fx[f"{split}.manifest"] = np.array(lines[1:])
            names = [ds.fnames[i].as_py() if not isinstance(ds.fnames, list) else ds.fnames[i] for i in range(len(ds.fnames))]
            fx[f"{split}.kept"] = np.array(names)
            fx[f"{split}.sizes"] = ds.sizes
            items = [ds[i] for i in range(len(names))]
            for i, it in enumerate(items):
                fx[f"{split}.item{i}.source"] = it["source"]
                fx[f"{split}.item{i}.target"] = it["target"]
            np.random.seed(123)
            col = ds.collater(items)
            fx[f"{split}.collate.id"] = col["id"]
            fx[f"{split}.collate.source"] = col["net_input"]["source"]
            fx[f"{split}.collate.target"] = col["target"]
            fx[f"{split}.collate.ntokens"] = np.array(col["ntokens"])
            check(f"G17 {split}: target is net_input.target", col["target"].float(), col["net_input"]["target"].float(), 0.0)
    return fx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also validate large dims at L=80000 (slow)")
    ap.add_argument("--no-write", action="store_true")
    ap.add_argument("--trained", action="store_true",
                    help="only tests/golden/tiny_trained.npz: a small hd-64 model 200 oracle updates from init (needs no reference)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    if args.trained:
        fx = trained_fixture()
        if not args.no_write:
            save("tiny_trained.npz", fx)
        return
    refnn = ref_shims.install()

    sx = sinc_fixtures(refnn)
    model, cfg, fx = run_pair(refnn, O.TINY, TINY_LAYERS, TINY_LAYERS_STR, B=2, L=8000, keep=16, seed=1)
    ex = ema_fixture(model, cfg, fx)
    mx = mixup_fixture(refnn)
    ftx = finetune_fixture(refnn, model, cfg)
    fttx = finetune_train_fixture(refnn, model, cfg)
    # second pair: real front end (127x63 sinc, 512-wide convs), small transformer; validated, not stored
    mid = dict(O.TINY, embed_dim=128, num_heads=8, conv_pos_groups=16, decoder_groups=16, decoder_dim=64)
    run_pair(refnn, mid, O.LARGE_LAYERS, ref_shims.LARGE_LAYERS, B=1, L=16000, keep=28, seed=2)
    if args.full:
        run_pair(refnn, O.LARGE, O.LARGE_LAYERS, ref_shims.LARGE_LAYERS, B=1, L=80000, keep=140, seed=3,
                 tol=5e-5, grads=True)
    dpx = data_pipeline_fixture(refnn)
    inx = init_fixture(refnn)
    if not args.no_write:
        save("data_pipeline.npz", dpx)
        save("init_stats.npz", inx)
        save("sinc.npz", sx)
        save("tiny_pretrain.npz", fx)
        save("tiny_ema.npz", ex)
        save("mixup.npz", mx)
        save("tiny_finetune.npz", ftx)
        save("tiny_finetune_train.npz", fttx)



TRAINED_DIMS = dict(embed_dim=64, num_heads=1, depth=2, prenet_depth=1, decoder_dim=32, decoder_layers=2, decoder_kernel=7,
                    decoder_groups=4, clone_batch=2, conv_pos_groups=4, average_top_k_layers=2, ema_decay=0.99, ema_end_decay=0.999,
                    ema_anneal_end_step=200)


def _structured_clips(n, L, seed, f_lo=200.0, f_step=150.0):
    """a corpus with structure to learn: band-limited noise bursts over a tone that drifts per clip (white noise trains nothing)"""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L) / 8000.0
    clips = []
    for i in range(n):
        f0 = f_lo + f_step * i
        x = torch.sin(2 * torch.pi * (f0 + 40.0 * torch.sin(2 * torch.pi * 1.5 * t)) * t) + 0.3 * torch.randn(L, generator=g)
        x = x * (0.3 + (torch.rand(L // 800, generator=g).repeat_interleave(800) > 0.5).float())
        clips.append(x)
    return torch.nn.functional.layer_norm(torch.stack(clips), (L,))


def trained_fixture(steps=200):
    """parity OFF init with fixed inputs.  A small model with head dimension 64 (the bf16 kernels' shape) is trained
    `steps` updates by the CPU oracle alone -- Adam, gradient clipping, span masks of the oracle's counter stream, EMA teacher; every
    seed fixed, eval-mode arithmetic (the oracle restates no dropout) -- then student, teacher, a held-out batch and the oracle's
    outputs and gradients on it are stored.  The GPU test compares the HIP path on exactly these weights with FIXED bounds."""
    dims = dict(TRAINED_DIMS)
    ocfg = O.OracleConfig(conv_feature_layers=list(TINY_LAYERS), **dims)
    torch.manual_seed(0)
    p = {k: v.clone().requires_grad_(True) for k, v in O.init_params(ocfg, seed=5).items()}
    ema = O.make_teacher(p)
    L = 8000
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    clips = _structured_clips(16, L, 21)
    decay_keys = [k for k, v in p.items() if v.dim() > 1 and "alibi_scale" not in k and "p_swish" not in k]
    opt = torch.optim.Adam([dict(params=[p[k] for k in decay_keys], weight_decay=0.01),
                            dict(params=[v for k, v in p.items() if k not in decay_keys], weight_decay=0.0)],
                           lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    g = torch.Generator().manual_seed(77)
    first = last = None
    for step in range(steps):
        lr = 1e-3 * min(1.0, (step + 1) / 10.0)
        for grp in opt.param_groups:
            grp["lr"] = lr
        src = clips[(step % 2) * 8:(step % 2) * 8 + 8]
        mask = torch.from_numpy(O.span_mask(8 * ocfg.clone_batch, T, 1.5, 2, seed=1000 + step)).float()
        keep = int(T - mask[0].sum())
        noise = torch.randn(8 * ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
        opt.zero_grad(set_to_none=True)
        res = O.pretrain_forward(p, ema, src, mask, noise, ocfg)
        loss = res["losses"]["AUDIO_regression"].sum() / float(res["sample_size"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 1.0)
        opt.step()
        O.ema_step(p, ema, O.ema_decay_at(ocfg, step + 1))
        if step == 4:
            first = float(loss)
        last = float(loss)
        if step % 20 == 0:
            print(f"trained fixture: step {step} loss {float(loss):.4f} kept {keep}", flush=True)
    assert last < first, (first, last)
    # held-out batch: other tones, a seeded mask with an equal count per row
    B, keep = 2, 16
    src = _structured_clips(B, L, 55, f_lo=300.0, f_step=700.0)
    mask = O.seeded_mask(B * ocfg.clone_batch, T, keep, 12)
    g2 = torch.Generator().manual_seed(56)
    noise = torch.randn(B * ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g2) * 0.01
    po = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
    res["losses"]["AUDIO_regression"].sum().backward()
    fx = dict(source=src, mask=mask, noise=noise, loss_sum=res["losses"]["AUDIO_regression"].detach().sum(),
              sample_size=torch.tensor(float(res["sample_size"])), loss_first=torch.tensor(first), loss_last=torch.tensor(last),
              dims=np.frombuffer(json.dumps(dims).encode(), dtype=np.uint8), layers=np.asarray(TINY_LAYERS, dtype=np.int64))
    for k in ("local_features", "targets", "decoder_out"):
        fx["out." + k] = res["_dbg"][k].detach().float()
    for k, v in po.items():
        fx["p." + k] = v.detach()
        fx["g." + k] = v.grad.detach()
    for k, v in ema.items():
        fx["ema." + k] = v.detach()
    print(f"trained fixture: loss {first:.4f} -> {last:.4f}; held-out loss_sum {float(fx['loss_sum']):.4f}")
    return fx


def init_fixture(refnn):
    """G18 (rows a1 / a21): initial parameter statistics of the LIVE reference model (mid dims: real 127 x 63 sinc + 512-wide
    conv stack, D = 128) -- per parameter: numel, mean, std, abs-max, and the values themselves where the init is
    deterministic (SincConv band edges, LayerNorm, PSwish, alibi_scale, zero biases).  Three seeds for the random laws."""
    mid = dict(O.TINY, embed_dim=128, num_heads=8, conv_pos_groups=16, decoder_groups=16, decoder_dim=64)
    fx = {}
    for seed in (0, 1, 2):
        kw = dict(mid)
        model, _ = ref_shims.build_reference_model(refnn, conv_feature_layers=ref_shims.LARGE_LAYERS, seed=seed, **kw)
        sd = {k: v for k, v in model.state_dict().items() if k != "_ema" and not k.startswith("_ema")}
        for k, v in sd.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            v = v.detach().float()
            fx[f"s{seed}.{k}.stats"] = np.array([v.numel(), float(v.mean()), float(v.std()) if v.numel() > 1 else 0.0, float(v.abs().max())])
            deterministic = float(v.std()) == 0.0 if v.numel() > 1 else True
            if seed == 0 and (deterministic or "low_hz_" in k or "band_hz_" in k):
                fx[f"value.{k}"] = v.numpy()
    return fx


def finetune_recipes_fixture():
    """tests/golden/finetune_recipes.json: the task / dataset / criterion / optimization / optimizer / lr_scheduler /
    model sections of the three shipped finetune recipes (configs/MeerKAT/finetune_mixup_{001,025,100}.yaml) as parsed
    data -- the inputs of tests/test_host_logic.py::test_finetune_yaml_mapping_shipped_recipes.
    (run: python -c "import gen_golden as g; g.finetune_recipes_fixture()" from oracle/)"""
    import json

    import yaml
    out = {}
    for n in ("001", "025", "100"):
        y = yaml.safe_load(open(f"/root/reference/configs/MeerKAT/finetune_mixup_{n}.yaml"))
        out[f"finetune_mixup_{n}"] = {k: y[k] for k in ("task", "dataset", "criterion", "optimization", "optimizer", "lr_scheduler", "model")}
    with open(os.path.join(GOLD, "finetune_recipes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
