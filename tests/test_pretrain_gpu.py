"""GPU end-to-end parity of the pretraining step (student fwd+bwd, EMA-teacher targets, loss, every
parameter gradient) against (a) the golden vectors dumped from the LIVE reference and (b) the oracle.

Tolerance: the north-star's 1e-3 relative (||a-b||/||b||) in the fp32 parity mode; the bf16 throughput
mode is checked against the oracle at bf16-level tolerances stated per assertion."""
import pytest
import torch

import a2v_oracle as O
from conftest import parity_report, rel_err, split_golden

pytestmark = pytest.mark.gpu

TINY_LAYERS = [(31, 15, 1), (32, 10, 5)] + [(32, 3, 2)] * 3 + [(32, 3, 1)] + [(32, 2, 1)] * 2
# d alibi_scale, bf16 engine against the fp32 engine on the large model at T = 11 025 (no oracle run there to measure the term scale
# from; measured 1.3e-2): the same bound as every other gradient of that comparison
ALIBI_BF16_LONG = 6e-2
AL_KEY = "modality_encoders.AUDIO.alibi_scale"
# bounds of alibi_term_scale.check: the error of d alibi_scale against the summed magnitude A of its terms must stay below about ONE
# unit roundoff of the storage type (fp32: 2^-24 = 6e-8, bound 2e-7; bf16: 2^-9 = 2e-3).  Measured (profiles/r03_parity_report.jsonl):
# fp32 1.4e-8 .. 4.9e-8, bf16 6.9e-5 .. 6.9e-4, over condition numbers A / |g| of 20 (D = 128, 8 kHz), 452 (the large model at 8 kHz),
# 326 (base*), 99 / 673 (T = 5513 / 11 025) -- the relative errors of round 2 (5.7e-2, 2.8e-1) are exactly condition number x this.
# Round 4: the bf16 bound is 1e-3 (measured 4.0e-5 .. 6.9e-4, so at most 1.5x slack at the worst config), and
# every mode also keeps a RELATIVE ceiling of 0.5 -- a sign or scale error of the kernel (relative error >= 1) cannot hide behind A.
# What the relative error itself is when the sum does NOT cancel is pinned at full dims by
# tests/test_ops_gpu.py::test_alibi_scale_gradient_well_conditioned_full_dims (bf16: 5e-2, condition number <= 30).
ALIBI_BACK = {torch.float32: 2e-7, torch.bfloat16: 1e-3, "bf16x3": 3e-5}     # (bf16x3: products exact to 2^-16 = 1.5e-5)
ALIBI_REL_CEILING = 0.5


X3 = "bf16x3"        # parity mode on the bf16 matrix cores: fp32 storage, the Linears as three exact bf16 products (Engine(x3=True))


def build(cfg, p, ema, dtype):
    from animal2vec_amd.params import ParamStore
    from animal2vec_amd.engine import Engine
    store = ParamStore(cfg, device="cuda")
    store.load_state_dict(p, ema)
    if dtype == X3:
        return store, Engine(cfg, store, dtype=torch.float32, x3=True)
    return store, Engine(cfg, store, dtype=dtype)


class alibi_term_scale:
    """Context manager around an oracle forward + backward: collects, per head, A_h = sum over layers and (n, i, j) of
    |dS| |d bias / d alibi_scale_h| -- the sum of the absolute values of the terms whose signed sum is d loss / d alibi_scale_h.
    That gradient is cancellation-dominated (|g_h| << A_h: at random init nothing prefers near keys to far ones), so its RELATIVE
    error is the rounding error of the terms times the condition number A_h / |g_h|, whatever the kernel does: the bf16 tests
    bound the error by a multiple of 2^-9 A (bf16 inputs: each term carries a 2^-9-relative error; random signs over 10^5 .. 10^8
    terms make the measured ratio much smaller) and report A / |g| next to the relative error."""

    def __enter__(self):
        O.ALIBI_PROBE = []
        return self

    def __exit__(self, *exc):
        self.parts, O.ALIBI_PROBE = O.ALIBI_PROBE, None

    def check(self, g_hip, p_alibi_scale, g_ref, rel_bound):
        """g_hip / g_ref: (1,1,H,1,1) gradients; returns (relative error, condition number, error / A)."""
        sc = p_alibi_scale.detach().reshape(-1).clamp_min(1e-12)
        A = torch.stack(self.parts).sum(0) / sc
        err = (g_hip.detach().float().cpu().reshape(-1) - g_ref.reshape(-1)).norm()
        rel, cond, back = float(err / g_ref.norm()), float(A.norm() / g_ref.norm()), float(err / A.norm())
        if rel_bound is not None:           # (None: measure only -- the caller bounds the RELATIVE error of a well-conditioned sum)
            assert back < rel_bound, (rel, cond, back)
        assert rel < ALIBI_REL_CEILING, (rel, cond, back)
        return rel, cond, back


def run(engine, source, mask, noise, **kw):
    from animal2vec_amd.masking import make_maskinfo
    m, keep, restore = make_maskinfo(mask.cuda())
    engine.stats.zero_()
    engine.store.grad.zero_()
    dbg = engine.forward_backward(source.cuda(), m, keep, restore, noise.cuda(), keep_debug=True, **kw)
    torch.cuda.synchronize()
    return dbg


def test_tiny_fp32_vs_reference_golden(tiny_golden):
    from animal2vec_amd.config import tiny_config
    fx = tiny_golden
    p, ema, g = split_golden(fx)
    cfg = tiny_config()
    store, eng = build(cfg, p, ema, torch.float32)
    dbg = run(eng, fx["source"], fx["mask"], fx["noise"])
    TOL = 1e-3
    assert rel_err(dbg["local_features"], fx["local_features"]) < TOL
    assert rel_err(dbg["targets"], fx["targets"]) < TOL            # EMA-teacher targets
    assert rel_err(dbg["decoder_out"], fx["dbg.decoder_out"]) < TOL
    mb = fx["mask"].bool()
    assert rel_err(dbg["loss_rows"].cpu()[mb], fx["loss"]) < TOL
    D = cfg.embed_dim
    st = eng.stats.cpu()
    assert abs(float(st[4 * D]) - float(fx["loss_sum"])) / float(fx["loss_sum"]) < TOL
    assert int(st[4 * D + 1]) == int(fx["sample_size"])
    worst = ("", 0.0)
    for k, gr in g.items():
        e = rel_err(store.g[k], gr)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < TOL, worst


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 3e-2, 5e-2), (X3, 1e-3, 1e-3)])
def test_hd64_vs_oracle(dtype, tol_y, tol_g):
    """Real front end (127x63 sinc, 512-wide convs), head_dim 64, D=128: HIP path vs oracle on the box's CPU."""
    from animal2vec_amd.config import Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig
    dims = dict(embed_dim=128, num_heads=2, depth=2, prenet_depth=1, decoder_dim=64, decoder_layers=2,
                decoder_kernel=7, decoder_groups=8, clone_batch=3, conv_pos_groups=8, average_top_k_layers=2)
    ocfg = O.OracleConfig(**dims)
    p = O.init_params(ocfg, seed=3)
    g = torch.Generator().manual_seed(5)
    for k in p:                      # move off the init so every parameter matters
        if "low_hz_" in k or "band_hz_" in k:
            p[k] = p[k] + torch.randn(p[k].shape, generator=g) * 3
        elif "alibi_scale" in k:
            p[k] = p[k] + torch.rand(p[k].shape, generator=g) * 0.5
        elif "p_swish_beta" in k:
            p[k] = p[k] + torch.randn(p[k].shape, generator=g) * 0.3
        else:
            p[k] = p[k] + torch.randn(p[k].shape, generator=g) * 0.02
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.01 for k, v in O.make_teacher(p).items()}
    B, L, keep = 2, 16000, 28
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    src = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,))
    mask = O.seeded_mask(B * ocfg.clone_batch, T, keep, 11)
    noise = torch.randn(B * ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    with alibi_term_scale() as ats:
        res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
        res["losses"]["AUDIO_regression"].sum().backward()

    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1,
                       decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=2, clone_batch=3, average_top_k_layers=2, audio=a)
    store, eng = build(cfg, p, ema, dtype)
    dbg = run(eng, src, mask, noise)
    assert rel_err(dbg["local_features"].float(), res["_dbg"]["local_features"]) < tol_y
    assert rel_err(dbg["targets"], res["_dbg"]["targets"]) < tol_y
    assert rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]) < tol_y
    D = cfg.embed_dim
    ls = float(res["losses"]["AUDIO_regression"].sum())
    assert abs(float(eng.stats[4 * D]) - ls) / ls < tol_y
    bad = []
    allerr = []
    for k, v in po.items():
        e = rel_err(store.g[k], v.grad)
        allerr.append((e, k))
        if e > tol_g:
            bad.append((k, e))
    al_rel, al_cond, al_back = ats.check(store.g[AL_KEY], p[AL_KEY], po[AL_KEY].grad, ALIBI_BACK[dtype])
    parity_report("hd64_vs_oracle", dtype, allerr, targets=rel_err(dbg["targets"], res["_dbg"]["targets"]),
                  decoder_out=rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]), alibi_rel=al_rel, alibi_cond=al_cond,
                  alibi_back=al_back)
    assert not bad, bad[:8]


def test_model_surface_dropin(tiny_golden):
    """The nn.Module surface: reference state-dict keys, forward dict, autograd backward into p.grad, EMA hook."""
    import nn
    from animal2vec_amd.config import tiny_config
    fx = tiny_golden
    p, ema, g = split_golden(fx)
    model = nn.Data2VecMultiModel.build_model(tiny_config(), task=None, dtype=torch.float32)
    sd = dict(p)
    sd["_ema"] = ema
    model.load_state_dict(sd)
    got = model.state_dict()
    assert set(got) == set(p) | {"_ema"} and set(got["_ema"]) == set(ema)
    assert all(torch.equal(got[k].cpu(), p[k]) for k in p)
    model.train()
    model.dropout = model.mixup = False        # the reference vectors were produced in eval mode
    model.store.grad.zero_()
    out = model(fx["source"], precomputed_mask=fx["mask"], noise=fx["noise"].cuda())
    assert set(out) >= {"losses", "sample_size", "masked_pct", "pred_var", "target_var", "ema_decay"}
    loss = out["losses"]["AUDIO_regression"].float().sum()
    assert abs(float(loss) - float(fx["loss_sum"])) / float(fx["loss_sum"]) < 1e-3
    assert int(out["sample_size"]) == int(fx["sample_size"])
    assert abs(float(out["pred_var"]) - float(fx["pred_var"])) < 1e-4
    assert abs(float(out["target_var"]) - float(fx["target_var"])) < 1e-4
    assert out["masked_pct"] == pytest.approx(1 - 16 / 200)
    loss.backward()
    named = dict(model.named_parameters())
    assert set(named) == set(p)
    worst = max(rel_err(named[k].grad, g[k]) for k in g)
    assert worst < 1e-3, worst
    # EMA hook (set_num_updates, data2vec2.py:386-410) against the golden EMA vector
    from conftest import load_golden
    ex = load_golden("tiny_ema.npz")
    model.set_num_updates(1)
    after = model.state_dict()["_ema"]
    assert max(rel_err(after[k], ex["after." + k]) for k in after) < 1e-6
    # eval: no autograd node, unreduced loss on request
    model.eval()
    with torch.no_grad():
        out = model(fx["source"], precomputed_mask=fx["mask"], noise=fx["noise"].cuda(), unreduced_loss=True)
    assert out["losses"]["AUDIO_regression"].shape == fx["loss"].shape


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_extract_features(tiny_golden, dtype, tol):
    import nn
    from animal2vec_amd.config import tiny_config
    fx = tiny_golden
    p, ema, _ = split_golden(fx)
    if dtype == torch.bfloat16:
        pytest.skip("tiny config has head_dim 16; bf16 attention needs head_dim 32/64")
    model = nn.Data2VecMultiModel.build_model(tiny_config(), dtype=dtype)
    sd = dict(p)
    sd["_ema"] = ema
    model.load_state_dict(sd)
    model.eval()
    res = model.extract_features(fx["source"])
    assert rel_err(res["x"].float(), fx["features_only.x"]) < tol
    lm = sum(t.float() for t in res["layer_results"]) / len(res["layer_results"])
    assert rel_err(lm, fx["features_only.layer_mean"]) < tol


def test_dropout_gradients_by_finite_differences(tiny_golden):
    """Train-mode dropout (deterministic counter-based masks): the hand-written backward must be the gradient of
    the forward it was run on.  Central differences on a few scalars of the fp32 engine."""
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.masking import make_maskinfo
    fx = tiny_golden
    p, ema, _ = split_golden(fx)
    cfg = tiny_config()
    cfg.dropout_input = 0.15             # 0.0 in the shipped YAML; exercised here (data2vec2.py:646-647)
    store, eng = build(cfg, p, ema, torch.float32)
    m, keep, restore = make_maskinfo(fx["mask"].cuda())
    src, noise = fx["source"].cuda(), fx["noise"].cuda()
    D = cfg.embed_dim
    SEED = 4242

    def loss():
        eng.sp.refresh()
        eng.stats.zero_()
        eng.forward(src, m, keep, restore, noise, save=False, dropout_seed=SEED)
        torch.cuda.synchronize()
        return float(eng.stats[4 * D])

    store.grad.zero_()
    eng.stats.zero_()
    eng.forward_backward(src, m, keep, restore, noise, dropout_seed=SEED)
    l_drop = float(eng.stats[4 * D])
    assert abs(l_drop - float(fx["loss_sum"])) / float(fx["loss_sum"]) > 1e-4      # dropout really changed the forward
    probes = [("blocks.1.attn.qkv.weight", (5, 7)), ("blocks.0.mlp.fc2.bias", (3,)),
              ("modality_encoders.AUDIO.context_encoder.blocks.0.attn.proj.weight", (2, 9)),
              ("modality_encoders.AUDIO.decoder.blocks.0.0.weight", (4, 3, 2)),
              ("modality_encoders.AUDIO.context_encoder.norm.weight", (11,)),
              ("modality_encoders.AUDIO.alibi_scale", (0, 0, 1, 0, 0))]
    for name, idx in probes:
        g = float(store.g[name][idx])
        w = store.p[name]
        old = float(w[idx])
        eps = 2e-2 * max(1.0, abs(old))
        w[idx] = old + eps
        lp = loss()
        w[idx] = old - eps
        lm = loss()
        w[idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g) <= 0.08 * max(abs(g), abs(fd)) + 2e-2, (name, fd, g)
    eng.sp.refresh()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_operand_packing_matches_single_launches(tiny_golden, dtype):
    """a2v_pack_weight_batch (one launch per parameter set, tiled Linear transpose) against a2v_pack_weight per weight."""
    from animal2vec_amd.config import tiny_config
    cfg = tiny_config()
    p, ema, _ = split_golden(tiny_golden)
    _, eng = build(cfg, p, ema, dtype)
    for packs in (eng.sp, eng.tp):
        packs.refresh_one_by_one()
        ref = {("w", k): v.clone() for k, v in packs.w.items()}
        ref.update({("wt", k): v.clone() for k, v in packs.wt.items()})
        for v in list(packs.w.values()) + list(packs.wt.values()):
            if v.data_ptr() not in {s.data_ptr() for s in packs.src.values()}:      # fp32 Linear operands alias the params
                v.zero_()
        packs.refresh()
        torch.cuda.synchronize()
        for (kind, k), v in ref.items():
            got = (packs.w if kind == "w" else packs.wt)[k]
            assert torch.equal(got, v), (kind, k)


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 2e-2, 4e-2), (X3, 1e-3, 1e-3)])
def test_large_full_size_vs_oracle(dtype, tol_y, tol_g):
    """BASELINE.json's configuration at full size (large model: D 1024, 16 heads, 8 + 16 blocks, decoder 4 x 768,
    clone_batch 12, one 10 s clip @ 8 kHz, 140 of 2000 frames kept): HIP path against the oracle run on this box's
    CPU.  fp32 mode carries the north-star's 1e-3.  The bf16 bounds follow the measured per-tensor table of this very
    comparison (profiles/r02_bf16_parity_large.json, tools/bf16_parity.py): local features 1.1e-2, decoder output
    9.1e-3, loss 1.7e-4, 342 gradient tensors with median 1.1e-2 and 2.7e-2 for the worst ordinary one (the sinc
    band edges) -> 2e-2 on activations, 4e-2 on every gradient, with two named exceptions that have a written cause:
      * targets (4.4e-2 measured, bound 6e-2): the teacher runs 24 post-LN blocks over all 2000 frames in bf16, the
        average of 16 layer outputs is instance-normalised over time, which divides out the signal's scale but not
        the accumulated rounding; the reference's own bf16 autocast of the same forward is at 1.17e-1;
      * d alibi_scale: 16 numbers, each the sum over 24 layers x 96 sequences x 140^2 score gradients times |i - j| with heavy
        cancellation (norm 3e-3 of a 1e3-norm gradient vector; condition number 452 measured through the oracle's probe): its error
        is bounded against the magnitude of its TERMS (alibi_term_scale: below one bf16 unit roundoff of it; 5.7e-2 relative = 452 x
        1.3e-4), not by a relative bound a cancelling sum cannot keep.  The whole gradient vector is within 7.2e-3 (autocast oracle: 8.6e-3)."""
    from animal2vec_amd.config import large_config
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ocfg = O.OracleConfig(**O.LARGE)
    p = O.init_params(ocfg, seed=0)
    g = torch.Generator().manual_seed(9)
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.002 for k, v in O.make_teacher(p).items()}
    L, keep = 80000, 140
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,))
    mask = O.seeded_mask(ocfg.clone_batch, T, keep, 11)
    noise = torch.randn(ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    with alibi_term_scale() as ats:
        res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
        res["losses"]["AUDIO_regression"].sum().backward()

    cfg = large_config()
    store, eng = build(cfg, p, ema, dtype)
    dbg = run(eng, src, mask, noise)
    bf = dtype == torch.bfloat16
    assert rel_err(dbg["local_features"].float(), res["_dbg"]["local_features"]) < tol_y
    assert rel_err(dbg["targets"], res["_dbg"]["targets"]) < (6e-2 if bf else tol_y)
    assert rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]) < tol_y
    D = cfg.embed_dim
    ls = float(res["losses"]["AUDIO_regression"].sum())
    assert abs(float(eng.stats[4 * D]) - ls) / ls < (2e-3 if bf else tol_y)
    errs = sorted(((rel_err(store.g[k], v.grad), k) for k, v in po.items()), reverse=True)
    al_rel, al_cond, al_back = ats.check(store.g[AL_KEY], p[AL_KEY], po[AL_KEY].grad, ALIBI_BACK[dtype])
    parity_report("large_full_size", dtype, errs, targets=rel_err(dbg["targets"], res["_dbg"]["targets"]), alibi_rel=al_rel,
                  alibi_cond=al_cond, alibi_back=al_back)
    bad = [(e, k) for e, k in errs if e >= tol_g and k != AL_KEY]       # d alibi_scale: held by ats.check above (see alibi_term_scale)
    assert not bad, bad[:8]
    if bf:
        # the bf16 mode keeps GELU' in 8 bits (A2V_EPI_DACT8, |error| <= 0.0025) -- a deliberate deviation from the reference's
        # backward.  Same weights, same inputs, the exact derivative (A2V_DACT8=0): the Mlp weights' gradients -- the tensors the
        # derivative feeds -- must be within the bound BOTH ways and the 8-bit form may cost at most 15 % of their error
        from animal2vec_amd import ops as _ops
        ffn = [k for k in po if ".mlp.fc1.weight" in k or ".mlp.fc2.weight" in k]
        e8 = sorted(rel_err(store.g[k], po[k].grad) for k in ffn)
        was = _ops.DACT8
        _ops.DACT8 = False
        try:
            eng.close()
            store2, eng2 = build(cfg, p, ema, dtype)
            run(eng2, src, mask, noise)
            ex = sorted(rel_err(store2.g[k], po[k].grad) for k in ffn)
            eng2.close()
        finally:
            _ops.DACT8 = was
        parity_report("large_full_size_dact8", dtype, [(e, "ffn") for e in e8[-3:]], ffn_median_dact8=e8[len(e8) // 2], ffn_worst_dact8=e8[-1],
                      ffn_median_exact=ex[len(ex) // 2], ffn_worst_exact=ex[-1], dact8_default=bool(was))
        assert e8[-1] < tol_g and ex[-1] < tol_g, (e8[-1], ex[-1])
        assert e8[len(e8) // 2] < 1.15 * ex[len(ex) // 2] + 1e-4, (e8[len(e8) // 2], ex[len(ex) // 2])
    # the update direction as a whole (all 315.8 M gradient entries as one vector)
    ga = torch.cat([store.g[k].detach().float().cpu().reshape(-1) for k in po])
    gb = torch.cat([v.grad.reshape(-1) for v in po.values()])
    assert rel_err(ga, gb) < (1.5e-2 if bf else 1e-4)


# ------------------------------------------------------------------------------------------ finetune path (8f rank 1)
def _finetune_run(cfg, p, ema, dtype, source, target, head_w, head_b, aug=None):
    from animal2vec_amd.finetune import FinetuneHead, finetune_step
    store, eng = build(cfg, p, ema, dtype)
    head = FinetuneHead(cfg.embed_dim, head_w.shape[0], dtype=dtype)
    head.load(head_w.cuda(), head_b.cuda())
    store.grad.zero_()
    head.zero_grad()
    logits, loss = finetune_step(eng, head, source.cuda(), target.cuda(), aug=aug)
    torch.cuda.synchronize()
    return store, head, logits, float(loss)


def test_finetune_tiny_fp32_vs_reference_golden(tiny_golden):
    """logits, summed focal loss and every gradient of the finetune path against the vectors dumped from the live
    reference (extract_features + nn.utils.sigmoid_focal_loss): the north-star's 1e-3."""
    from animal2vec_amd.config import tiny_config
    from conftest import load_golden
    ft = load_golden("tiny_finetune.npz")
    p, ema, _ = split_golden(tiny_golden)
    store, head, logits, loss = _finetune_run(tiny_config(), p, ema, torch.float32, ft["source"], ft["target"],
                                              ft["head_w"], ft["head_b"])
    assert rel_err(logits.float(), ft["logits"]) < 1e-3
    assert abs(loss - float(ft["loss_sum"])) / float(ft["loss_sum"]) < 1e-3
    assert rel_err(head.grad_w, ft["g.head_w"]) < 1e-3 and rel_err(head.grad_b, ft["g.head_b"]) < 1e-3
    gk = [k[2:] for k in ft if k.startswith("g.") and not k.startswith("g.head_")]
    bad = [(k, rel_err(store.g[k], ft["g." + k])) for k in gk if rel_err(store.g[k], ft["g." + k]) > 1e-3]
    assert not bad, bad[:8]
    # parameters the reference leaves without gradient (frozen conv feature extractor, decoder, last norm2) stay zero
    for k in store.g:
        if k not in gk:
            assert float(store.g[k].abs().max()) == 0.0, k


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 3e-2, 5e-2)])
def test_finetune_hd64_vs_oracle(dtype, tol_y, tol_g):
    """Real front end, head_dim 64, S = T = 400 (the attention backward with dQ atomics): HIP finetune step vs oracle."""
    from animal2vec_amd.config import Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig
    dims = dict(embed_dim=128, num_heads=2, depth=3, prenet_depth=1, decoder_dim=64, decoder_layers=2,
                decoder_kernel=7, decoder_groups=8, clone_batch=3, conv_pos_groups=8, average_top_k_layers=2)
    ocfg = O.OracleConfig(**dims)
    p = O.init_params(ocfg, seed=4)
    g = torch.Generator().manual_seed(6)
    for k in p:
        if "alibi_scale" in k:
            p[k] = p[k] + torch.rand(p[k].shape, generator=g) * 0.5
        elif "low_hz_" not in k and "band_hz_" not in k:
            p[k] = p[k] + torch.randn(p[k].shape, generator=g) * 0.02
    ema = O.make_teacher(p)
    B, L, C = 2, 16000, 12
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    src = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,))
    target = (torch.rand(B, T, C, generator=g) < 0.2).float() * torch.rand(B, T, C, generator=g)     # soft labels
    hw, hb = torch.randn(C, 128, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ow, ob = hw.clone().requires_grad_(True), hb.clone().requires_grad_(True)
    logits_o = O.finetune_forward(po, ow, ob, src, ocfg)
    loss_o = O.sigmoid_focal_loss(logits_o, target, reduction="sum")
    loss_o.backward()
    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1,
                       decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=3, embed_dim=128, num_heads=2, clone_batch=3, average_top_k_layers=2, audio=a)
    store, head, logits, loss = _finetune_run(cfg, p, ema, dtype, src, target, hw, hb)
    assert rel_err(logits.float(), logits_o.detach()) < tol_y
    assert abs(loss - float(loss_o)) / float(loss_o) < tol_y
    assert rel_err(head.grad_w, ow.grad) < tol_g and rel_err(head.grad_b, ob.grad) < tol_g
    bad = []
    allerr = []
    for k, v in po.items():
        if v.grad is None:
            assert float(store.g[k].abs().max()) == 0.0, k
            continue
        e = rel_err(store.g[k], v.grad)
        allerr.append((e, k))
        if e > tol_g:
            bad.append((k, e))
    parity_report("finetune_hd64_vs_oracle", dtype, allerr, logits=rel_err(logits.float(), logits_o.detach()),
                  head_w=rel_err(head.grad_w, ow.grad), head_b=rel_err(head.grad_b, ob.grad))
    # (d alibi_scale needs no exception any more: 4.6e-2 in round 2, 1.7e-2 with the distance centred on the row's mean -- a2v.h: mdist)
    assert not bad, bad[:8]


def test_finetune_model_surface_dropin(tiny_golden):
    """nn.Wav2VecCcasFinetune + nn.FinetuneCrossEntropyCriterion used the fairseq way (criterion(model, sample) ->
    loss.backward()): same loss and gradients as the golden vectors of the live reference."""
    import nn
    from animal2vec_amd.config import tiny_config
    from conftest import load_golden
    ft = load_golden("tiny_finetune.npz")
    p, ema, _ = split_golden(tiny_golden)
    enc = nn.Data2VecMultiModel.build_model(tiny_config(), task=None, dtype=torch.float32)
    enc.store.load_state_dict(p, ema)
    enc.engine.refresh_student_packs()
    model = nn.Wav2VecCcasFinetune(enc, n_labels=ft["head_w"].shape[0])
    with torch.no_grad():
        model.proj.weight.copy_(ft["head_w"].cuda())
        model.proj.bias.copy_(ft["head_b"].cuda())
    model.sync_head()
    model.train()
    enc.store.grad.zero_()
    crit = nn.FinetuneCrossEntropyCriterion(report_accuracy=True)
    sample = {"net_input": {"source": ft["source"]}, "target": ft["target"].cuda(), "ntokens": ft["target"].numel()}
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ft["loss_sum"])) / float(ft["loss_sum"]) < 1e-3
    assert rel_err(model.proj.weight.grad, ft["g.head_w"]) < 1e-3 and rel_err(model.proj.bias.grad, ft["g.head_b"]) < 1e-3
    for k in ("blocks.0.mlp.fc1.weight", "modality_encoders.AUDIO.project_features.2.weight",
              "modality_encoders.AUDIO.relative_positional_encoder.1.0.weight", "modality_encoders.AUDIO.alibi_scale"):
        assert rel_err(enc.store.g[k], ft["g." + k]) < 1e-3, k
    assert log["finetune/total"] == ft["target"].numel()
    # eval mode: logits only, nothing saved
    model.eval()
    with torch.no_grad():
        out = model(ft["source"])
    assert rel_err(out["encoder_out"], ft["logits"]) < 1e-3


def test_finetune_train_form_fp32_vs_reference_golden(tiny_golden):
    """Train form of the finetune step -- noise-replace time masking, channel masking, layerdrop -- against vectors from
    the live reference in train() mode with pinned draws (tests/golden/tiny_finetune_train.npz): 1e-3."""
    from animal2vec_amd.config import tiny_config
    from conftest import load_golden
    ft = load_golden("tiny_finetune_train.npz")
    p, ema, _ = split_golden(tiny_golden)
    aug = dict(mask_t=ft["mask_t"].cuda(), mask_c=ft["mask_c"].cuda(), replacement=ft["replacement"].cuda(),
               active=[bool(v) for v in ft["active"]])
    store, head, logits, loss = _finetune_run(tiny_config(), p, ema, torch.float32, ft["source"], ft["target"],
                                              ft["head_w"], ft["head_b"], aug=aug)
    assert rel_err(logits.float(), ft["logits"]) < 1e-3
    assert abs(loss - float(ft["loss_sum"])) / float(ft["loss_sum"]) < 1e-3
    assert rel_err(head.grad_w, ft["g.head_w"]) < 1e-3 and rel_err(head.grad_b, ft["g.head_b"]) < 1e-3
    n = 0
    for k in ft:
        if not k.startswith("g.") or k.startswith("g.head_"):
            continue
        ref = ft[k]
        if float(ref.abs().max()) == 0.0:
            assert float(store.g[k[2:]].abs().max()) == 0.0, k
            continue
        assert rel_err(store.g[k[2:]], ref) < 1e-3, (k, rel_err(store.g[k[2:]], ref))
        n += 1
    assert n > 25
    dropped = [i for i, a in enumerate(aug["active"]) if not a]
    assert dropped
    for k in store.g:          # a dropped block gets no gradient at all
        for i in dropped:
            name = f"blocks.{i - tiny_config().audio.prenet_depth}." if i >= tiny_config().audio.prenet_depth else None
            if name and k.startswith(name):
                assert float(store.g[k].abs().max()) == 0.0, k


def test_finetune_train_form_dropout_by_finite_differences(tiny_golden):
    """All train-form pieces at once (device-hash replacement noise, channel mask, layerdrop, the dropouts of
    wav2vec2.py:94-128 incl. the Mlp's two activation dropouts and attention dropout): the hand-written backward is
    the gradient of the forward it ran, by central differences on the fp32 engine."""
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.finetune import FinetuneHead
    from conftest import load_golden
    ft = load_golden("tiny_finetune_train.npz")
    p, ema, _ = split_golden(tiny_golden)
    cfg = tiny_config()
    store, eng = build(cfg, p, ema, torch.float32)
    head = FinetuneHead(cfg.embed_dim, ft["head_w"].shape[0], dtype=torch.float32)
    head.load(ft["head_w"].cuda(), ft["head_b"].cuda())
    src, tgt = ft["source"].cuda(), ft["target"].cuda()
    aug = dict(mask_t=ft["mask_t"].cuda(), mask_c=ft["mask_c"].cuda(), noise_std=0.5, noise_seed=77,
               active=[True, True, True], drop_seed=991, p_drop=0.1, p_act=0.1, p_att=0.2, p_in=0.15, p_final=0.1)

    def loss():
        eng.sp.refresh()
        head.refresh()
        head.loss_sum.zero_()
        eng.finetune_forward_backward(src, tgt, head, backward=False, aug=aug)
        torch.cuda.synchronize()
        return float(head.loss_sum)

    l_eval = None
    head.loss_sum.zero_()
    eng.finetune_forward_backward(src, tgt, head, backward=False)
    l_eval = float(head.loss_sum)
    store.grad.zero_()
    head.zero_grad()
    eng.finetune_forward_backward(src, tgt, head, aug=aug)
    l_train = float(head.loss_sum)
    assert abs(l_train - l_eval) / l_eval > 1e-3                   # the train form really changed the forward
    assert loss() == pytest.approx(l_train, rel=1e-6)              # and is deterministic in its seeds
    probes = [("blocks.1.attn.qkv.weight", (5, 7)), ("blocks.0.mlp.fc2.bias", (3,)), ("blocks.0.mlp.fc1.weight", (9, 4)),
              ("blocks.1.mlp.fc1.bias", (6,)),
              ("modality_encoders.AUDIO.context_encoder.blocks.0.attn.proj.weight", (2, 9)),
              ("modality_encoders.AUDIO.context_encoder.norm.weight", (11,)),
              ("modality_encoders.AUDIO.relative_positional_encoder.1.0.weight", (3, 1, 2)),
              ("modality_encoders.AUDIO.project_features.2.weight", (4, 6)),
              ("modality_encoders.AUDIO.alibi_scale", (0, 0, 1, 0, 0))]
    for name, idx in probes:
        g = float(store.g[name][idx])
        w = store.p[name]
        old = float(w[idx])
        eps = 2e-2 * max(1.0, abs(old))
        w[idx] = old + eps
        lp = loss()
        w[idx] = old - eps
        lm = loss()
        w[idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g) <= 0.08 * max(abs(g), abs(fd)) + 2e-2, (name, fd, g)
    # head: exact (loss is linear in nothing, but the head is the last op) -- finite differences on one weight
    gw = float(head.grad_w[3, 5])
    old = float(head.w[3, 5])
    head.w[3, 5] = old + 1e-2
    lp = loss()
    head.w[3, 5] = old - 1e-2
    lm = loss()
    head.w[3, 5] = old
    assert abs((lp - lm) / 2e-2 - gw) <= 0.05 * abs(gw) + 1e-2
    eng.sp.refresh()
    head.refresh()


def test_finetune_trainer_freeze_and_layer_decay(tiny_golden):
    """FinetuneTrainer: classifier-only updates while frozen (wav2vec2.py:440-445); then the encoder trains with the
    per-block lr scales of layer_decay (first Adam step moves every weight by lr x scale); the conv feature extractor
    and the decoder never move."""
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.finetune import FinetuneConfig, FinetuneTrainer, cosine_lr, layer_lr_scales
    from animal2vec_amd.params import ParamStore, P
    from conftest import load_golden
    ft = load_golden("tiny_finetune.npz")
    p, ema, _ = split_golden(tiny_golden)
    cfg = tiny_config()
    store = ParamStore(cfg)
    store.load_state_dict(p, ema)
    f = FinetuneConfig(n_labels=12, freeze_finetune_updates=1, layer_decay=0.5, optim_overrides=True, warmup_updates=1, lr=1e-3, min_lr=1e-4,
                       max_update=10, mask_length=2, mask_channel_length=4)
    tr = FinetuneTrainer(cfg, f, store=store, dtype=torch.float32, seed=3)
    mbs = [dict(source=ft["source"].cuda(), target=ft["target"].cuda()) for _ in range(2)]
    flat0, w0 = store.flat.clone(), tr.head.w.clone()
    assert tr.frozen
    tr.train_step(mbs)
    torch.cuda.synchronize()
    assert torch.equal(store.flat, flat0)                       # frozen: encoder untouched
    assert float((tr.head.w - w0).abs().max()) > 0
    m1 = tr.metrics()
    assert m1["sample_size"] == 2 * ft["target"].shape[0] * ft["target"].shape[1] and m1["loss"] > 0
    assert not tr.frozen
    tr.train_step(mbs)
    torch.cuda.synchronize()
    lr = cosine_lr(f, 2)
    sc = layer_lr_scales(cfg.audio.prenet_depth + cfg.depth, 0.5)
    d = (store.flat - flat0).abs()

    def moved(name):
        s_ = store.by_name[name]
        return float(d[s_.offset:s_.offset + s_.numel].max())

    assert moved(P + "context_encoder.blocks.0.mlp.fc1.weight") == pytest.approx(lr * sc[0], rel=2e-2)
    assert moved("blocks.0.mlp.fc1.weight") == pytest.approx(lr * sc[1], rel=2e-2)
    assert moved("blocks.1.attn.qkv.weight") == pytest.approx(lr * sc[2], rel=2e-2)
    assert moved(P + "project_features.2.weight") == pytest.approx(lr, rel=2e-2)
    assert moved(P + "relative_positional_encoder.1.0.weight") == pytest.approx(lr, rel=2e-2)
    for name in store.by_name:
        if "local_encoder" in name or "decoder" in name:
            assert moved(name) == 0.0, name


def test_real_data_pretrain_checkpoint_finetune_end_to_end(tmp_path):
    """Rows 3 + 4 of SURVEY 8f wired through the hot loops: wav corpus + manifest + label files on disk -> one
    pretraining update on real clips -> fairseq-style checkpoint -> finetune from it on the labelled clips (frozen
    update, then an encoder update) -> finetuned checkpoint -> reload and reproduce the logits."""
    import os
    import wave
    import numpy as np
    from animal2vec_amd import checkpoint as ck, data as D
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.finetune import FinetuneConfig, FinetuneHead
    from animal2vec_amd.params import ParamStore
    from animal2vec_amd.train_routine import animal2vec_audio_main, animal2vec_finetune_main
    rng = np.random.default_rng(0)
    root = tmp_path / "corpus" / "wav"
    os.makedirs(root)
    lines = [str(root)]
    labels = ["a", "b", "focal"]
    for i in range(6):
        n = 8000
        x = (rng.standard_normal(n) * 0.1).astype(np.float32)
        with wave.open(str(root / f"c{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
            w.writeframes((x * 32768).astype("<i2").tobytes())
        s0 = np.sort(rng.integers(0, 6000, 3))
        lp = D.filename_audio2label(str(root / f"c{i}.wav"), lblext="npz")
        os.makedirs(os.path.dirname(lp), exist_ok=True)
        np.savez(lp, start_frame_lbl=s0, end_frame_lbl=s0 + 1500, lbl_cat=rng.integers(0, 2, 3), foc=rng.integers(0, 2, 3))
        lines.append(f"c{i}.wav\t{n}")
    man = tmp_path / "train_0.tsv"
    man.write_text("\n".join(lines) + "\n")
    cfg = tiny_config()
    logs = []
    pre = str(tmp_path / "checkpoint_last.pt")
    tr = animal2vec_audio_main(cfg, updates=1, update_freq=2, data=str(man), max_tokens=3 * 8000, dtype=torch.float32,
                               log=logs.append, save_path=pre)
    assert tr.num_updates == 1 and logs and logs[0]["sample_size"] > 0 and np.isfinite(logs[0]["loss"])
    f = FinetuneConfig(n_labels=3, freeze_finetune_updates=1, warmup_updates=1, lr=1e-3, min_lr=1e-4, max_update=10,
                       mask_length=2, mask_channel_length=4)
    flogs = []
    out = str(tmp_path / "finetuned.pt")
    ft = animal2vec_finetune_main(cfg, f, data=str(man), unique_labels=labels, pretrained=pre, updates=2, update_freq=2,
                                  max_tokens=3 * 8000, dtype=torch.float32, log=flogs.append, save_path=out)
    assert [l["num_updates"] for l in flogs] == [1, 2] and all(np.isfinite(l["loss"]) for l in flogs)
    assert flogs[0]["sample_size"] == 2 * 3 * 200
    # pretrained weights arrived: project_features differs from a fresh init and equals the checkpoint before training moved it
    st = ck.load_checkpoint(pre)
    assert "_ema" in st["model"]
    # reload the finetuned model into a fresh store / head: eval logits identical
    store2 = ParamStore(cfg)
    head2 = FinetuneHead(cfg.embed_dim, 3, dtype=torch.float32)
    ck.load_finetuned(store2, head2, ck.load_checkpoint(out))
    from animal2vec_amd.engine import Engine
    eng2 = Engine(cfg, store2, dtype=torch.float32)
    src = torch.nn.functional.layer_norm(torch.from_numpy(rng.standard_normal((2, 8000)).astype(np.float32)), (8000,)).cuda()
    a_ = ft.engine.finetune_forward(src, ft.head, save=False)
    b_ = eng2.finetune_forward(src, head2, save=False)
    assert torch.equal(a_, b_)


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 3e-2, 5e-2)])
def test_finetune_large_full_size_vs_oracle(dtype, tol_y, tol_g):
    """BASELINE config 4 at full size (large model, one 10 s clip, all 2000 frames, 12 labels), train form with
    explicit draws (time mask with noise replacement, channel mask, two dropped blocks): HIP path -- the two-pass
    long-sequence attention backward in bf16 -- against the oracle on this box's CPU."""
    from animal2vec_amd.config import large_config
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ocfg = O.OracleConfig(**O.LARGE)
    p = O.init_params(ocfg, seed=0)
    g = torch.Generator().manual_seed(12)
    for k in p:                                           # a trained-looking model: learned alibi scales, perturbed weights
        if "alibi_scale" in k:
            p[k] = p[k] + torch.rand(p[k].shape, generator=g) * 0.5
    ema = O.make_teacher(p)
    L, C = 80000, 12
    D = ocfg.embed_dim
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,))
    target = (torch.rand(1, T, C, generator=g) < 0.1).float()
    hw, hb = torch.randn(C, D, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1
    mask_t = (torch.rand(1, T, generator=g) < 0.6).to(torch.uint8)
    mask_c = (torch.rand(1, D, generator=g) < 0.3).to(torch.uint8)
    repl = torch.randn(int(mask_t.sum()), D, generator=g) * 0.01
    active = [True] * (ocfg.prenet_depth + ocfg.depth)
    active[3] = False
    active[ocfg.prenet_depth + 5] = False
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ow, ob = hw.clone().requires_grad_(True), hb.clone().requires_grad_(True)
    logits_o = O.finetune_forward(po, ow, ob, src, ocfg, mask_t=mask_t, replacement=repl, mask_c=mask_c, active=active)
    loss_o = O.sigmoid_focal_loss(logits_o, target, reduction="sum")
    loss_o.backward()
    aug = dict(mask_t=mask_t.cuda(), mask_c=mask_c.cuda(), replacement=repl.cuda(), active=active)
    store, head, logits, loss = _finetune_run(large_config(), p, ema, dtype, src, target, hw, hb, aug=aug)
    assert rel_err(logits.float(), logits_o.detach()) < tol_y
    assert abs(loss - float(loss_o.detach())) / float(loss_o.detach()) < tol_y
    assert rel_err(head.grad_w, ow.grad) < tol_g and rel_err(head.grad_b, ob.grad) < tol_g
    errs = []
    for k, v in po.items():
        if v.grad is None:
            assert float(store.g[k].abs().max()) == 0.0, k
            continue
        errs.append((rel_err(store.g[k], v.grad), k))
    errs.sort(reverse=True)
    parity_report("finetune_large_full_size", dtype, errs, logits=rel_err(logits.float(), logits_o.detach()),
                  head_w=rel_err(head.grad_w, ow.grad), head_b=rel_err(head.grad_b, ob.grad))
    named = {}                          # (round 2 excepted d alibi_scale at 8e-2; 2.9e-2 measured now, inside the common bound)
    bad = [(e, k) for e, k in errs if e >= named.get(k, tol_g)]
    assert len(errs) > 250 and not bad, bad[:8]


@pytest.mark.parametrize("L,T_want", [(220520, 5513), (441000, 11025)])
@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 4e-2, 5e-2)])
def test_long_sequence_pretrain_vs_oracle(dtype, tol_y, tol_g, L, T_want):
    """BASELINE config 5's sequence shape (5 s @ 44.1 kHz: T = 5513 frames, K = round(0.07 T) = 386 kept) through the
    pretraining step with head_dim 64: teacher attention over 5513 keys, student attention at S = 386 (forward with two
    query blocks, the two-pass backward in bf16), pos-conv / decoder slabs over ragged T.  (The 44.1 kHz front-end spec itself, 296 filters x 149 taps, is covered by
    test_config5_44k_frontend_vs_oracle; the waveform here goes through the 127 x 63 front end to keep the CPU oracle's run
    short.)"""
    from animal2vec_amd.config import Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig
    torch.set_num_threads(min(16, torch.get_num_threads()))
    dims = dict(embed_dim=128, num_heads=2, depth=2, prenet_depth=1, decoder_dim=64, decoder_layers=2,
                decoder_kernel=7, decoder_groups=8, clone_batch=2, conv_pos_groups=8, average_top_k_layers=2)
    ocfg = O.OracleConfig(**dims)
    p = O.init_params(ocfg, seed=5)
    g = torch.Generator().manual_seed(8)
    for k in p:
        if "alibi_scale" in k:
            p[k] = p[k] + torch.rand(p[k].shape, generator=g) * 0.5
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.002 for k, v in O.make_teacher(p).items()}
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    assert T == T_want                      # 5 s and the full 10 s of BASELINE config 5 (T = 11 025, K = 772)
    keep = round(0.07 * T)
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,))
    mask = O.seeded_mask(ocfg.clone_batch, T, keep, 3)
    noise = torch.randn(ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    with alibi_term_scale() as ats:
        res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
        res["losses"]["AUDIO_regression"].sum().backward()
    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1,
                       decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=2, clone_batch=2, average_top_k_layers=2, audio=a)
    store, eng = build(cfg, p, ema, dtype)
    dbg = run(eng, src, mask, noise)
    assert rel_err(dbg["targets"], res["_dbg"]["targets"]) < tol_y
    assert rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]) < tol_y
    ls = float(res["losses"]["AUDIO_regression"].sum())
    assert abs(float(eng.stats[4 * cfg.embed_dim]) - ls) / ls < tol_y
    errs = sorted(((rel_err(store.g[k], v.grad), k) for k, v in po.items()), reverse=True)
    al_rel, al_cond, al_back = ats.check(store.g[AL_KEY], p[AL_KEY], po[AL_KEY].grad, ALIBI_BACK[dtype])
    parity_report(f"long_sequence_pretrain_T{T}", dtype, errs, targets=rel_err(dbg["targets"], res["_dbg"]["targets"]),
                  decoder_out=rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]), alibi_rel=al_rel, alibi_cond=al_cond,
                  alibi_back=al_back)
    # d alibi_scale is cancellation-dominated (alibi_term_scale): its error is bounded against the magnitude of its terms by
    # ats.check above -- round 2's relative bounds (8e-2, and 3.5e-1 at T = 11 025) are gone.  What the kernels could fix they do:
    # the distance is centred on the row's mean attended distance (a2v.h: mdist), which removes the e_i = dO_i.(O_bf16 - O) residue
    # of every row (fp32 mode 4e-5 -> 2e-5 here, finetune bf16 4.6e-2 -> 1.7e-2); what is left at T = 11 025 (2.8e-1 relative, a
    # condition number of several thousand) is the rounding of dO / q / k / v to bf16 acting on d g / d dO_i = sum_j P_ij (|.| - c_i) V_j.
    bad = [(e, k) for e, k in errs if e >= tol_g and k != AL_KEY]
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 4e-2, 5e-2)])
def test_config5_44k_frontend_vs_oracle(dtype, tol_y, tol_g):
    """BASELINE config 5's front end: 44.1 kHz, 296 sinc filters x 149 taps (the spec SURVEY 8d derives from the YAML's
    own rule k ~ sqrt(sr / 2)), through the blocked SincConv kernels (tap chunks of 64, 320-channel pitch, filter
    gradient per (channel group, tap chunk)) and a 2960-wide first conv: the whole pretraining step against the oracle,
    including the gradients of low_hz_ / band_hz_, the channel LayerNorm and PSwish."""
    from animal2vec_amd.config import Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig
    torch.set_num_threads(min(16, torch.get_num_threads()))
    layers = [(296, 149, 1), (64, 10, 5)] + [(64, 3, 2)] * 3 + [(64, 3, 1)] + [(64, 2, 1)] * 2
    layers_str = "[(296, 149, 1), (64, 10, 5)] + [(64, 3, 2)] * 3 + [(64, 3, 1)] + [(64, 2, 1)] * 2"
    dims = dict(embed_dim=128, num_heads=2, depth=2, prenet_depth=1, decoder_dim=64, decoder_layers=2, decoder_kernel=7,
                decoder_groups=8, clone_batch=2, conv_pos_groups=8, average_top_k_layers=2, conv_feature_layers=layers,
                sample_rate=44100)
    ocfg = O.OracleConfig(**dims)
    p = O.init_params(ocfg, seed=6)
    g = torch.Generator().manual_seed(10)
    for k in p:                                # move the band edges / PSwish / LN off their init values
        if "low_hz_" in k or "band_hz_" in k:
            p[k] = p[k] * (1 + 0.05 * torch.randn(p[k].shape, generator=g))
        elif "p_swish" in k or "conv_layers.0.2.1" in k:
            p[k] = p[k] + 0.1 * torch.randn(p[k].shape, generator=g)
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.002 for k, v in O.make_teacher(p).items()}
    L = 22050                                  # 0.5 s
    T = O.frame_counts(L, layers)[-1]
    keep = max(8, round(0.07 * T))
    src = torch.nn.functional.layer_norm(torch.randn(2, L, generator=g), (L,))
    mask = O.seeded_mask(2 * ocfg.clone_batch, T, keep, 3)
    noise = torch.randn(2 * ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    with alibi_term_scale() as ats:
        res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
        res["losses"]["AUDIO_regression"].sum().backward()
    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1, conv_feature_layers=layers_str, sample_rate=44100,
                       decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=2, clone_batch=2, average_top_k_layers=2, audio=a)
    store, eng = build(cfg, p, ema, dtype)
    assert eng.sinc_cp == 320
    dbg = run(eng, src, mask, noise)
    assert rel_err(dbg["local_features"].float(), res["_dbg"]["local_features"]) < tol_y
    assert rel_err(dbg["targets"], res["_dbg"]["targets"]) < tol_y
    ls = float(res["losses"]["AUDIO_regression"].sum().detach())
    assert abs(float(eng.stats[4 * cfg.embed_dim]) - ls) / ls < tol_y
    errs = sorted(((rel_err(store.g[k], v.grad), k) for k, v in po.items()), reverse=True)
    parity_report("config5_44k_frontend", dtype, errs, targets=rel_err(dbg["targets"], res["_dbg"]["targets"]),
                  local_features=rel_err(dbg["local_features"].float(), res["_dbg"]["local_features"]))
    assert errs[0][0] < tol_g, errs[:8]
    fe = [e for e in errs if "conv_layers.0." in e[1]]
    assert len(fe) == 6, fe                    # low_hz_, band_hz_, LN weight / bias, PSwish alpha / beta


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 4e-2, 5e-2)])
def test_config5_whole_10s_44k_vs_oracle(dtype, tol_y, tol_g):
    """BASELINE config 5 as ONE step (missing #4): the 296 x 149 SincConv front end at 44.1 kHz AND the full
    10 s clip (L = 441 000 -> T = 11 025 frames, K = 772 kept) in the same forward / backward -- blocked sinc kernels over
    441 000 samples, a 2960-wide first conv, teacher attention over 11 025 keys, the S = 772 student attention with the
    two-pass backward, pos-conv / decoder slabs over ragged T -- against the oracle, with D = 128 transformer dims and 64-wide
    feature convs so that the CPU side finishes in well under a minute.  (The two halves were tested separately before:
    test_config5_44k_frontend_vs_oracle at 0.5 s, test_long_sequence_pretrain_vs_oracle behind the 127 x 63 front end.)"""
    from animal2vec_amd.config import Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig
    torch.set_num_threads(min(16, torch.get_num_threads()))
    layers = [(296, 149, 1), (64, 10, 5)] + [(64, 3, 2)] * 3 + [(64, 3, 1)] + [(64, 2, 1)] * 2
    layers_str = "[(296, 149, 1), (64, 10, 5)] + [(64, 3, 2)] * 3 + [(64, 3, 1)] + [(64, 2, 1)] * 2"
    dims = dict(embed_dim=128, num_heads=2, depth=2, prenet_depth=1, decoder_dim=64, decoder_layers=2, decoder_kernel=7,
                decoder_groups=8, clone_batch=2, conv_pos_groups=8, average_top_k_layers=2, conv_feature_layers=layers,
                sample_rate=44100)
    ocfg = O.OracleConfig(**dims)
    p = O.init_params(ocfg, seed=7)
    g = torch.Generator().manual_seed(12)
    for k in p:
        if "alibi_scale" in k:
            p[k] = p[k] + torch.rand(p[k].shape, generator=g) * 0.5
        elif "low_hz_" in k or "band_hz_" in k:
            p[k] = p[k] * (1 + 0.05 * torch.randn(p[k].shape, generator=g))
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.002 for k, v in O.make_teacher(p).items()}
    L = 441000
    T = O.frame_counts(L, layers)[-1]
    assert T == 11025
    keep = round(0.07 * T)
    assert keep == 772
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,))
    mask = O.seeded_mask(ocfg.clone_batch, T, keep, 3)
    noise = torch.randn(ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    with alibi_term_scale() as ats:
        res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
        res["losses"]["AUDIO_regression"].sum().backward()
    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1, conv_feature_layers=layers_str, sample_rate=44100,
                       decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=2, clone_batch=2, average_top_k_layers=2, audio=a)
    store, eng = build(cfg, p, ema, dtype)
    assert eng.sinc_cp == 320
    dbg = run(eng, src, mask, noise)
    assert rel_err(dbg["local_features"].float(), res["_dbg"]["local_features"]) < tol_y
    assert rel_err(dbg["targets"], res["_dbg"]["targets"]) < tol_y
    assert rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]) < tol_y
    ls = float(res["losses"]["AUDIO_regression"].sum().detach())
    assert abs(float(eng.stats[4 * cfg.embed_dim]) - ls) / ls < tol_y
    errs = sorted(((rel_err(store.g[k], v.grad), k) for k, v in po.items()), reverse=True)
    al_rel, al_cond, al_back = ats.check(store.g[AL_KEY], p[AL_KEY], po[AL_KEY].grad, ALIBI_BACK[dtype])
    parity_report("config5_whole_10s_44k", dtype, errs, targets=rel_err(dbg["targets"], res["_dbg"]["targets"]),
                  local_features=rel_err(dbg["local_features"].float(), res["_dbg"]["local_features"]), alibi_rel=al_rel,
                  alibi_cond=al_cond, alibi_back=al_back)
    bad = [(e, k) for e, k in errs if e >= tol_g and k != AL_KEY]
    assert not bad, bad[:8]


def test_config5_whole_large_model_bf16_vs_fp32_engine():
    """The thing `bench.py --model large44k` runs, as a test: the LARGE model (D 1024, 16 heads, 8 + 16 blocks, clone_batch 12,
    decoder 4 x 768) behind the 296 x 149 front end with 512-wide feature convs on one whole 10 s clip @ 44.1 kHz (T = 11 025,
    K = 772).  The oracle cannot run this size in test time (45.5 TFLOP per clip), so this is a PROPERTY check between the two
    HIP modes, stated as such: the bf16 engine against the fp32 engine (which the D = 128 form of the same step above pins to the
    oracle at 1e-3) -- finite loss, identical sample_size, targets / loss / every gradient within the bf16 bounds of the full-size
    8 kHz test."""
    from animal2vec_amd.config import large_44k_config
    from animal2vec_amd.engine import Engine
    from animal2vec_amd.masking import make_maskinfo, seeded_keep_mask
    from animal2vec_amd.params import ParamStore
    cfg = large_44k_config()
    L = 441000
    T = cfg.frame_counts(L)[-1]
    assert T == 11025
    K = 772
    g = torch.Generator().manual_seed(3)
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,)).cuda()
    m, keep, restore = make_maskinfo(seeded_keep_mask(cfg.clone_batch, T, K, 5, device="cuda"))
    noise = (torch.randn(cfg.clone_batch, T - K, cfg.embed_dim, generator=g) * 0.01).cuda()
    out = {}
    for dtype in (torch.float32, torch.bfloat16):
        store = ParamStore(cfg, device="cuda")
        store.init_reference(0)
        with torch.no_grad():                                # a teacher that is not the student, learned-looking ALiBi scales
            store.teacher.add_(torch.randn(store.teacher.shape, generator=torch.Generator(device="cuda").manual_seed(1),
                                           device="cuda") * 0.002)
        eng = Engine(cfg, store, dtype=dtype)
        eng.stats.zero_()
        store.grad.zero_()
        dbg = eng.forward_backward(src, m, keep, restore, noise, keep_debug=True)
        torch.cuda.synchronize()
        D = cfg.embed_dim
        out[dtype] = dict(loss=float(eng.stats[4 * D]), n=float(eng.stats[4 * D + 1]), targets=dbg["targets"].float().cpu(),
                          dec=dbg["decoder_out"].float().cpu(), grads={k: v.detach().float().cpu().clone() for k, v in store.g.items()})
        eng.close()
        del eng, store, dbg
        torch.cuda.empty_cache()
    f, b = out[torch.float32], out[torch.bfloat16]
    assert f["n"] == b["n"] == cfg.clone_batch * (T - K) and all(map(lambda v: v == v and abs(v) < 1e30, (f["loss"], b["loss"])))
    assert abs(f["loss"] - b["loss"]) / f["loss"] < 2e-3
    assert rel_err(b["targets"], f["targets"]) < 8e-2          # 24 post-LN blocks over 11 025 frames in bf16 (8 kHz: 4.4e-2 measured, bound 6e-2)
    assert rel_err(b["dec"], f["dec"]) < 3e-2
    errs = sorted(((rel_err(b["grads"][k], v), k) for k, v in f["grads"].items()), reverse=True)
    parity_report("config5_whole_large_bf16_vs_fp32", torch.bfloat16, errs, targets=rel_err(b["targets"], f["targets"]))
    named = {"modality_encoders.AUDIO.alibi_scale": ALIBI_BF16_LONG}
    bad = [(e, k) for e, k in errs if e >= named.get(k, 6e-2)]
    assert not bad, bad[:8]
    ga = torch.cat([b["grads"][k].reshape(-1) for k in f["grads"]])
    gb = torch.cat([v.reshape(-1) for v in f["grads"].values()])
    assert rel_err(ga, gb) < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_loop_learns(dtype):
    """Forty optimizer steps of the native trainer on a fixed set of synthetic clips (two-block model with head_dim 64
    behind the real 127 x 63 front end, dropout and mixup on,
    EMA teacher, clip + Adam): the masked-prediction loss stays finite and falls, pred_var does not collapse, the
    teacher follows the student."""
    from animal2vec_amd.config import Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig
    from animal2vec_amd.masking import make_maskinfo, seeded_keep_mask
    from animal2vec_amd.trainer import OptimConfig, PretrainTrainer
    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1,
                       decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=2, clone_batch=2, average_top_k_layers=2, audio=a)
    tr = PretrainTrainer(cfg, OptimConfig(lr=2e-3, warmup_updates=5, max_update=100, clip_norm=1.0), dtype=dtype, seed=1)
    L = 8000
    T = cfg.frame_counts(L)[-1]
    g = torch.Generator().manual_seed(0)
    mbs = []
    for j in range(2):
        src = torch.nn.functional.layer_norm(torch.randn(4, L, generator=g), (L,)).cuda()
        m, keep, restore = make_maskinfo(seeded_keep_mask(4 * cfg.clone_batch, T, 40, 100 + j, device="cuda"))
        mbs.append(dict(source=src, mask_u8=m, ids_keep=keep, ids_restore=restore, noise=None))
    teacher0 = tr.store.teacher.clone()
    losses, pvar = [], []
    for _ in range(40):
        tr.train_step(mbs)
        met = tr.metrics()
        losses.append(met["loss"])
        pvar.append(met["pred_var"])
    assert all(map(lambda v: v == v and abs(v) < 1e6, losses)), losses
    first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    assert last < 0.8 * first, (first, last)
    assert pvar[-1] > 0.01
    assert float((tr.store.teacher - teacher0).abs().max()) > 0
    assert tr.num_updates == 40


_TRAINED_HD64 = {}


def _trained_hd64(steps=200):
    """The hd64 configuration of test_hd64_vs_oracle after `steps` optimizer updates of the native bf16 trainer (fixed seeds, dropout
    and mixup on, masks sampled in the step, EMA teacher): student + teacher state as CPU fp32 dicts, made once per test session."""
    if "sd" in _TRAINED_HD64:
        return _TRAINED_HD64["sd"]
    from animal2vec_amd.config import Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig
    from animal2vec_amd.trainer import OptimConfig, PretrainTrainer
    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1, decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=2, clone_batch=3, average_top_k_layers=2, ema_decay=0.99,
                              ema_end_decay=0.999, ema_anneal_end_step=200, audio=a)
    tr = PretrainTrainer(cfg, OptimConfig(lr=1e-3, warmup_updates=10, max_update=400, clip_norm=1.0), dtype=torch.bfloat16, seed=3)
    L = 16000
    g = torch.Generator().manual_seed(21)
    # a corpus with structure to learn: band-limited noise bursts over a tone that drifts per clip (white noise alone trains nothing)
    t = torch.arange(L) / 8000.0
    clips = []
    for i in range(16):
        f0 = 200.0 + 150.0 * i
        x = torch.sin(2 * torch.pi * (f0 + 40.0 * torch.sin(2 * torch.pi * 1.5 * t)) * t) + 0.3 * torch.randn(L, generator=g)
        x = x * (0.3 + (torch.rand(L // 800, generator=g).repeat_interleave(800) > 0.5).float())
        clips.append(x)
    src = torch.nn.functional.layer_norm(torch.stack(clips), (L,)).cuda()
    mbs = [dict(source=src[:8], noise=None), dict(source=src[8:], noise=None)]      # masks: the step draws them (trainer._step_masks)
    first = last = None
    for i in range(steps):
        tr.train_step(mbs)
        if i == 4:
            first = tr.metrics()["loss"]
    last = tr.metrics()["loss"]
    assert last == last and last < first, (first, last)
    sd = tr.store.state_dict()
    ema = {k: v.float().cpu() for k, v in sd.pop("_ema").items()}
    p = {k: v.float().cpu() for k, v in sd.items()}
    tr.engine.close()
    _TRAINED_HD64["sd"] = (cfg, p, ema, float(first), float(last))
    return _TRAINED_HD64["sd"]


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 3e-2, 5e-2), (X3, 1e-3, 1e-3)])
def test_hd64_vs_oracle_after_200_updates(dtype, tol_y, tol_g):
    """every other parity test runs at (perturbed) random init, where LayerNorm statistics are generic,
    attention is flat and d alibi_scale is cancellation-dominated.  Here the HIP path is compared with the oracle on weights 200
    optimizer updates away from init (student and EMA teacher exported from the native trainer): local features, teacher targets,
    decoder output, loss, every gradient -- at the tolerances of the at-init test -- and the conditioning of d alibi_scale is
    recorded (`alibi_cond`): once the heads have learnt to prefer near frames the sum no longer cancels, and the bf16 mode's RELATIVE
    error of that gradient is asserted as well."""
    cfg, p, ema, first, last = _trained_hd64()
    dims = dict(embed_dim=128, num_heads=2, depth=2, prenet_depth=1, decoder_dim=64, decoder_layers=2,
                decoder_kernel=7, decoder_groups=8, clone_batch=3, conv_pos_groups=8, average_top_k_layers=2)
    ocfg = O.OracleConfig(**dims)
    g = torch.Generator().manual_seed(55)
    B, L, keep = 2, 16000, 28
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    t = torch.arange(L) / 8000.0
    src = torch.stack([torch.sin(2 * torch.pi * (300.0 + 700.0 * i) * t) + 0.3 * torch.randn(L, generator=g) for i in range(B)])
    src = torch.nn.functional.layer_norm(src, (L,))
    mask = O.seeded_mask(B * ocfg.clone_batch, T, keep, 12)
    noise = torch.randn(B * ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    with alibi_term_scale() as ats:
        res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
        res["losses"]["AUDIO_regression"].sum().backward()
    # bf16: the yardstick is what STOCK bf16 autocast of the oracle loses on the same weights and inputs.  Near a minimum the loss
    # gradient is a small difference of bf16-rounded predictions and fp32 targets: every bf16 path loses more relative accuracy there
    # than at init (measured: median 2.4e-2 against 0.7e-2 at init; autocast of the oracle: recorded in the parity report)
    ac, acy = {}, {}
    ls = float(res["losses"]["AUDIO_regression"].detach().sum())
    if dtype == torch.bfloat16:
        pa = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ra = O.pretrain_forward(pa, ema, src, mask, noise, ocfg)
            ra["losses"]["AUDIO_regression"].float().sum().backward()
        ac = {k: rel_err(pa[k].grad.float(), po[k].grad) for k in po}
        acy = {k: rel_err(ra["_dbg"][k].float(), res["_dbg"][k]) for k in ("local_features", "targets", "decoder_out")}
        acy["loss"] = abs(float(ra["losses"]["AUDIO_regression"].float().sum()) - ls) / ls
    store, eng = build(cfg, p, ema, dtype)
    dbg = run(eng, src, mask, noise)
    D = cfg.embed_dim
    ey = {k: rel_err(dbg[k].float(), res["_dbg"][k]) for k in ("local_features", "targets", "decoder_out")}
    ey["loss"] = abs(float(eng.stats[4 * D]) - ls) / ls
    # forward quantities: fp32 / bf16x3 at 1e-3; bf16 at 3e-2 or twice what stock autocast of the oracle loses on the same weights,
    # whichever is larger (the trainer is not bit-reproducible: the state 200 updates from init differs from run to run, loss 0.9 - 1.3,
    # and with it the error of the teacher targets -- measured 1.5e-2 ... 3.4e-2)
    ybound = {k: max(tol_y, 2.0 * acy.get(k, 0.0)) for k in ey}
    bad, allerr = [], []
    for k, v in po.items():
        e = rel_err(store.g[k], v.grad)
        allerr.append((e, k))
        # fp32 / bf16x3: 1e-3 on everything.  bf16 (measured on two runs of the non-bit-reproducible bf16 trainer: median 2.4e-2 /
        # 2.7e-2, worst 6.7e-2 / 5.7e-2; stock autocast of the oracle on the same weights: median 1.8e-2, worst 9.2e-2): every tensor
        # within 1e-1, the median within twice the autocast's
        # (d alibi_scale in bf16 has its own, condition-scaled bound below: at cond 11.7 it read 3.0e-1 = 2.6e-2 x cond)
        if e > (0.1 if dtype == torch.bfloat16 else tol_g) and not (k == AL_KEY and dtype == torch.bfloat16):
            bad.append((k, e, ac.get(k)))
    al_rel, al_cond, al_back = ats.check(store.g[AL_KEY], p[AL_KEY], po[AL_KEY].grad, None)
    extra = {}
    if ac:
        es = sorted(ac.values())
        extra = dict(autocast_median=es[len(es) // 2], autocast_worst=es[-1], autocast_alibi_rel=ac[AL_KEY])
    parity_report("hd64_after_200_updates", dtype, allerr, targets=ey["targets"], decoder_out=ey["decoder_out"],
                  local_features=ey["local_features"], loss=ey["loss"], alibi_rel=al_rel, alibi_cond=al_cond,
                  alibi_back=al_back, loss_first=first, loss_last=last, **extra, **{"autocast_" + k: v for k, v in acy.items()})
    for k in ey:
        assert ey[k] < ybound[k], (k, ey[k], ybound[k], acy.get(k))
    assert not bad, bad[:8]
    if ac:
        med = sorted(e for e, _ in allerr)[len(allerr) // 2]
        assert med < max(4e-2, 2.0 * extra["autocast_median"]), (med, extra)
    # d alibi_scale: where the sum is well conditioned ("expectation: << 452 => 5e-2 in bf16"; measured: 7.2) it is held to
    # the RELATIVE tolerance of every other gradient; should training ever leave it ill conditioned, to the backward-error bound
    # (bf16: the terms of the sum carry the relative error of every other bf16 gradient -- held to 5e-2 here -- and the sum amplifies it by
    # its condition number: five runs of this test measured cond 1.5 ... 12.7 and relative errors 2e-3 ... 1.8e-1 (= 3e-2 x cond at worst);
    # stock autocast of the oracle on the same weights: 1.5e-3 ... 1.6e-1)
    if al_cond < 60:
        assert al_rel < (5e-2 * max(2.0, al_cond) if dtype == torch.bfloat16 else tol_g), (al_rel, al_cond, ac.get(AL_KEY))
    else:
        assert al_back < ALIBI_BACK[dtype], (al_rel, al_cond, al_back)


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 3e-2, 5e-2)])
def test_finetune_hd64_vs_oracle_after_200_updates(dtype, tol_y, tol_g):
    """One finetune step (eval form: logits, focal loss, every gradient) from the same trained weights, HIP path vs oracle."""
    cfg, p, ema, _, _ = _trained_hd64()
    dims = dict(embed_dim=128, num_heads=2, depth=2, prenet_depth=1, decoder_dim=64, decoder_layers=2,
                decoder_kernel=7, decoder_groups=8, clone_batch=3, conv_pos_groups=8, average_top_k_layers=2)
    ocfg = O.OracleConfig(**dims)
    g = torch.Generator().manual_seed(56)
    B, L, C = 2, 16000, 12
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    t = torch.arange(L) / 8000.0
    src = torch.stack([torch.sin(2 * torch.pi * (500.0 + 400.0 * i) * t) + 0.3 * torch.randn(L, generator=g) for i in range(B)])
    src = torch.nn.functional.layer_norm(src, (L,))
    target = (torch.rand(B, T, C, generator=g) < 0.2).float() * torch.rand(B, T, C, generator=g)
    hw, hb = torch.randn(C, 128, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ow, ob = hw.clone().requires_grad_(True), hb.clone().requires_grad_(True)
    logits_o = O.finetune_forward(po, ow, ob, src, ocfg)
    loss_o = O.sigmoid_focal_loss(logits_o, target, reduction="sum")
    loss_o.backward()
    store, head, logits, loss = _finetune_run(cfg, p, ema, dtype, src, target, hw, hb)
    assert rel_err(logits.float(), logits_o.detach()) < tol_y
    assert abs(loss - float(loss_o)) / float(loss_o) < tol_y
    assert rel_err(head.grad_w, ow.grad) < tol_g and rel_err(head.grad_b, ob.grad) < tol_g
    bad, allerr = [], []
    for k, v in po.items():
        if v.grad is None:
            assert float(store.g[k].abs().max()) == 0.0, k
            continue
        e = rel_err(store.g[k], v.grad)
        allerr.append((e, k))
        if e > tol_g:
            bad.append((k, e))
    parity_report("finetune_hd64_after_200_updates", dtype, allerr, logits=rel_err(logits.float(), logits_o.detach()),
                  head_w=rel_err(head.grad_w, ow.grad), head_b=rel_err(head.grad_b, ob.grad))
    assert not bad, bad[:8]


# bf16 bounds of the deterministic off-init fixture: FIXED numbers, set from the measured values of the fixture
# (profiles/r06_parity_report.jsonl, rows "tiny_trained_golden") with about 1.5x slack -- not scaled by anything the run measures
TRAINED_BOUNDS = {torch.float32: dict(y=1e-3, med=1e-3, worst=1e-3, ffn=1e-3),
                  X3: dict(y=1e-3, med=1e-3, worst=1e-3, ffn=1e-3),
                  torch.bfloat16: dict(y=2.5e-2, med=3.5e-2, worst=8e-2, ffn=5e-2)}


def _trained_fixture():
    import json
    from conftest import load_golden
    fx = load_golden("tiny_trained.npz")
    dims = json.loads(bytes(fx["dims"].numpy().astype("uint8")).decode())
    p = {k[2:]: v for k, v in fx.items() if k.startswith("p.")}
    ema = {k[4:]: v for k, v in fx.items() if k.startswith("ema.")}
    g = {k[2:]: v for k, v in fx.items() if k.startswith("g.")}
    return fx, dims, p, ema, g


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3])
def test_tiny_trained_golden(dtype):
    """Parity OFF init on a committed state (tests/golden/tiny_trained.npz, oracle/gen_golden.py --trained: a head-dimension-64 model
    200 oracle updates from init, loss 5.4 -> 0.75; student, EMA teacher, a held-out batch, the oracle's outputs and every
    gradient).  The HIP path runs on exactly those weights: local features, teacher targets, decoder output, loss and all gradients
    against FIXED bounds; the Mlp weights' gradients (whose backward operand is the 8-bit GELU' in the bf16 mode) are bounded and
    reported separately."""
    from animal2vec_amd.config import tiny_config
    fx, dims, p, ema, gref = _trained_fixture()
    cfg = tiny_config(embed_dim=dims["embed_dim"], num_heads=dims["num_heads"], ema_decay=dims["ema_decay"],
                      ema_end_decay=dims["ema_end_decay"], ema_anneal_end_step=dims["ema_anneal_end_step"])
    b = TRAINED_BOUNDS[dtype]
    store, eng = build(cfg, p, ema, dtype)
    dbg = run(eng, fx["source"], fx["mask"], fx["noise"])
    D = cfg.embed_dim
    ey = {k: rel_err(dbg[k].float(), fx["out." + k]) for k in ("local_features", "targets", "decoder_out")}
    ls = float(fx["loss_sum"])
    ey["loss"] = abs(float(eng.stats[4 * D]) - ls) / ls
    allerr = sorted((rel_err(store.g[k], v), k) for k, v in gref.items())
    med, worst = allerr[len(allerr) // 2][0], allerr[-1]
    ffn = [(e, k) for e, k in allerr if ".mlp.fc1.weight" in k or ".mlp.fc2.weight" in k]
    parity_report("tiny_trained_golden", dtype, allerr, targets=ey["targets"], decoder_out=ey["decoder_out"],
                  local_features=ey["local_features"], loss=ey["loss"], ffn_weight_worst=max(ffn)[0],
                  ffn_weight_median=sorted(ffn)[len(ffn) // 2][0], loss_first=float(fx["loss_first"]), loss_last=float(fx["loss_last"]))
    eng.close()
    for k, v in ey.items():
        assert v < b["y"], (k, v)
    assert med < b["med"], med
    al = [e for e, k in allerr if k == AL_KEY][0]
    rest = [(e, k) for e, k in allerr if k != AL_KEY]
    assert rest[-1][0] < b["worst"], rest[-3:]
    assert max(ffn)[0] < b["ffn"], sorted(ffn)[-3:]
    # d alibi_scale off init is well conditioned (the heads have learnt to prefer near frames): the same ceiling as every tensor in
    # fp32 / bf16x3, twice it in bf16 (one number per head)
    assert al < (2 * b["worst"] if dtype == torch.bfloat16 else b["worst"]), al


def test_fp16_teacher_option():
    """the EMA teacher in IEEE half (Engine(teacher_half=True) / A2V_TEACHER_F16=1; liba2v_hip_f16.so, the same
    kernels with half as the 16-bit type).  On the committed off-init state the teacher targets must come out at least twice as close
    to the oracle's as with the bf16 teacher and within 1e-2; the rest of the step (bf16 student) stays inside the bf16 bounds, with
    the gradient ceiling at 1e-1: prediction and target no longer share their rounding, so their difference -- the loss gradient
    near a minimum -- loses the common-mode cancellation (measured 7.1e-2 -> 7.9e-2 here, 3.8e-2 -> 9.3e-2 at D 128).  Opt-in."""
    from animal2vec_amd import ops as _ops
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.params import ParamStore
    from animal2vec_amd.engine import Engine
    if not _ops.have_half_library():
        pytest.fail("liba2v_hip_f16.so is missing: build() makes it")
    fx, dims, p, ema, gref = _trained_fixture()
    cfg = tiny_config(embed_dim=dims["embed_dim"], num_heads=dims["num_heads"], ema_decay=dims["ema_decay"],
                      ema_end_decay=dims["ema_end_decay"], ema_anneal_end_step=dims["ema_anneal_end_step"])
    errs = {}
    for half in (False, True):
        store = ParamStore(cfg, device="cuda")
        store.load_state_dict(p, ema)
        eng = Engine(cfg, store, dtype=torch.bfloat16, teacher_half=half)
        assert eng.teacher_half == half
        dbg = run(eng, fx["source"], fx["mask"], fx["noise"])
        g = sorted(rel_err(store.g[k], v) for k, v in gref.items() if k != AL_KEY)
        errs[half] = dict(targets=rel_err(dbg["targets"].float(), fx["out.targets"]), med=g[len(g) // 2], worst=g[-1],
                          dec=rel_err(dbg["decoder_out"].float(), fx["out.decoder_out"]))
        eng.close()
    parity_report("fp16_teacher_option", torch.bfloat16, [(errs[True]["worst"], "worst, fp16 teacher"), (errs[False]["worst"], "worst, bf16 teacher")],
                  targets=errs[True]["targets"], targets_bf16_teacher=errs[False]["targets"], grad_median=errs[True]["med"],
                  grad_median_bf16_teacher=errs[False]["med"])
    assert errs[True]["targets"] < 1e-2 and errs[True]["targets"] < 0.5 * errs[False]["targets"], errs
    assert errs[True]["med"] < 3.5e-2 and errs[True]["worst"] < 1e-1 and errs[True]["dec"] < 2.5e-2, errs


def test_exact_resume_from_checkpoint(tmp_path):
    """Stop after two updates, write the fairseq-style checkpoint with the optimizer state, resume in a fresh trainer and
    take two more: same parameters / teacher as four updates in one go (dropout seeds derive from num_updates, the mixup
    generators are restored), up to the atomics' summation order."""
    from animal2vec_amd import checkpoint as ck
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.masking import make_maskinfo, seeded_keep_mask
    from animal2vec_amd.trainer import OptimConfig, PretrainTrainer
    cfg = tiny_config()
    oc = OptimConfig(lr=1e-3, warmup_updates=2, max_update=20)
    L = 3200
    T = cfg.frame_counts(L)[-1]
    g = torch.Generator().manual_seed(2)
    src = torch.randn(4, L, generator=g).cuda()
    m, keep, restore = make_maskinfo(seeded_keep_mask(4 * cfg.clone_batch, T, T // 4, 5, device="cuda"))
    noise = torch.randn(4 * cfg.clone_batch, T - T // 4, cfg.embed_dim, generator=g).cuda() * 0.01
    mbs = [dict(source=src, mask_u8=m, ids_keep=keep, ids_restore=restore, noise=noise)]
    a = PretrainTrainer(cfg, oc, dtype=torch.float32, seed=4)
    for _ in range(4):
        a.train_step(mbs)
    b = PretrainTrainer(cfg, oc, dtype=torch.float32, seed=4)
    for _ in range(2):
        b.train_step(mbs)
    path = tmp_path / "checkpoint_2.pt"
    ck.save_checkpoint(path, b.store, num_updates=b.num_updates, optimizer_state=b.optimizer_state())
    c = PretrainTrainer(cfg, oc, dtype=torch.float32, seed=99)          # different init: everything must come from the file
    state = ck.load_checkpoint(path)
    ck.load_pretrained(c.store, state)
    c.engine.refresh_student_packs()
    c.engine.refresh_teacher_packs()
    c.load_optimizer_state(state["last_optimizer_state"])
    assert c.num_updates == 2
    for _ in range(2):
        c.train_step(mbs)
    torch.cuda.synchronize()
    assert float((a.store.flat - c.store.flat).abs().max()) < 5e-6
    assert float((a.store.teacher - c.store.teacher).abs().max()) < 5e-6
    assert float((a.store.adam_v - c.store.adam_v).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------------ BASELINE configs 1 and 2 (base*)
def _base_star_setup(seed=0):
    ocfg = O.OracleConfig(**O.BASE_STAR)
    p = O.init_params(ocfg, seed=seed)
    g = torch.Generator().manual_seed(21)
    for k in p:                      # off the init: biases / alibi / p-swish matter too
        if "alibi_scale" in k:
            p[k] = p[k] + torch.rand(p[k].shape, generator=g) * 0.3
        elif k.endswith(".bias"):
            p[k] = p[k] + torch.randn(p[k].shape, generator=g) * 0.01
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.002 for k, v in O.make_teacher(p).items()}
    return ocfg, p, ema, g


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 3e-2, 5e-2)])
def test_base_star_pretrain_vs_oracle(dtype, tol_y, tol_g):
    """BASELINE config 2's model (base*: D 768, 12 heads, 4 + 12 blocks, decoder 4 x 384, clone_batch 8) on one 10 s clip
    @ 8 kHz with 140 of 2000 frames kept: targets, decoder output, loss and every gradient against the oracle on this
    box's CPU.  bf16 gradient bound: profiles/r02_bf16_parity_base.json (per-tensor table of the same comparison)."""
    from animal2vec_amd.config import base_star_config
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ocfg, p, ema, g = _base_star_setup()
    L, keep = 80000, 140
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,))
    mask = O.seeded_mask(ocfg.clone_batch, T, keep, 13)
    noise = torch.randn(ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    with alibi_term_scale() as ats:
        res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
        res["losses"]["AUDIO_regression"].sum().backward()
    cfg = base_star_config()
    store, eng = build(cfg, p, ema, dtype)
    dbg = run(eng, src, mask, noise)
    assert rel_err(dbg["local_features"].float(), res["_dbg"]["local_features"]) < tol_y
    assert rel_err(dbg["targets"], res["_dbg"]["targets"]) < tol_y
    assert rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]) < tol_y
    D = cfg.embed_dim
    ls = float(res["losses"]["AUDIO_regression"].sum())
    assert abs(float(eng.stats[4 * D]) - ls) / ls < tol_y
    assert int(eng.stats[4 * D + 1]) == int(res["sample_size"])
    errs = sorted(((rel_err(store.g[k], v.grad), k) for k, v in po.items()), reverse=True)
    al_rel, al_cond, al_back = ats.check(store.g[AL_KEY], p[AL_KEY], po[AL_KEY].grad, ALIBI_BACK[dtype])
    parity_report("base_star_pretrain", dtype, errs, targets=rel_err(dbg["targets"], res["_dbg"]["targets"]),
                  decoder_out=rel_err(dbg["decoder_out"].float(), res["_dbg"]["decoder_out"]), alibi_rel=al_rel, alibi_cond=al_cond,
                  alibi_back=al_back)
    bad = [(e, k) for e, k in errs if e >= tol_g and k != AL_KEY]       # d alibi_scale: ats.check above
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
def test_base_star_extract_features_4x10s(dtype, tol):
    """BASELINE config 1's workload (base*, forward only, 4 x 10 s @ 8 kHz) through the drop-in surface
    (nn.Data2VecMultiModel.extract_features) against the oracle; head_dim 64, so the bf16 leg runs the MFMA attention
    forward over all 2000 frames."""
    import nn
    from animal2vec_amd.config import base_star_config
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ocfg, p, ema, g = _base_star_setup(seed=1)
    B, L = 4, 80000
    src = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,))
    with torch.no_grad():
        x_ref, lrs_ref = O.features_only_forward(p, src, ocfg)
    model = nn.Data2VecMultiModel.build_model(base_star_config(), dtype=dtype)
    sd = dict(p)
    sd["_ema"] = ema
    model.load_state_dict(sd)
    model.eval()
    res = model.extract_features(src)
    assert res["x"].shape == x_ref.shape and len(res["layer_results"]) == len(lrs_ref)
    assert rel_err(res["x"].float(), x_ref) < tol
    worst = max(rel_err(a.float(), b) for a, b in zip(res["layer_results"], lrs_ref))
    assert worst < tol, worst


def test_training_routine_resumes_exactly_with_drawn_noise_and_masks(tmp_path):
    """animal2vec_audio_main stopped after two updates and resumed for two more equals four updates in one go, with the
    mask-token noise (noise=None: drawn from the trainer's generator), the span masks and the synthetic-batch cursor all
    coming out of the checkpoint (the default CUDA generator / mask rng were not saved)."""
    import nn
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.trainer import OptimConfig
    cfg = tiny_config()
    oc = OptimConfig(lr=1e-3, warmup_updates=2, max_update=20)
    kw = dict(clip_samples=3200, clips_per_micro_batch=2, update_freq=2, dtype=torch.float32, log=lambda *_: None)
    a = nn.animal2vec_audio_main(cfg, oc, updates=4, **kw)
    path = str(tmp_path / "ck.pt")
    nn.animal2vec_audio_main(cfg, oc, updates=2, save_path=path, **kw)
    torch.manual_seed(12345)                       # whatever happened to the global generators in between must not matter
    c = nn.animal2vec_audio_main(cfg, oc, updates=2, restore_file=path, **kw)
    torch.cuda.synchronize()
    assert c.num_updates == 4 == a.num_updates
    assert float((a.store.flat - c.store.flat).abs().max()) < 5e-6
    assert float((a.store.teacher - c.store.teacher).abs().max()) < 5e-6
