"""CPU: host-side logic of the drop-in surface (no HIP compute): config grammar, flat parameter layout,
decay rules, all-reduce buckets, masking, registries, criterion, YAML mapping."""
import math
import os

import numpy as np
import pytest
import torch

import a2v_oracle as O
from animal2vec_amd import config as C
from animal2vec_amd.params import ALIGN, P, build_specs, bucket_starts
from animal2vec_amd.ddp import make_buckets, grad_scale_factor
from animal2vec_amd.masking import compute_mask_indices, make_maskinfo, seeded_keep_mask


def test_conv_layer_grammar_is_parsed_without_eval():
    assert C.parse_conv_layers(C.LARGE_LAYERS_STR) == O.LARGE_LAYERS
    with pytest.raises(ValueError):
        C.parse_conv_layers("__import__('os').system('true')")


def test_frame_counts_match_oracle():
    cfg = C.large_config()
    assert cfg.frame_counts(80000) == O.frame_counts(80000, O.LARGE_LAYERS)
    assert cfg.pos_kernel == 19 and cfg.head_dim == 64


@pytest.mark.parametrize("mk,dims,n", [(C.large_config, O.LARGE, 315_830_026), (C.base_star_config, O.BASE_STAR, 122_780_678)])
def test_flat_layout(mk, dims, n):
    cfg = mk()
    specs, (lo, hi), total = build_specs(cfg)
    assert sum(s.numel for s in specs) == n                       # SURVEY section 4 known answers
    shapes = O.init_param_shapes(O.OracleConfig(**dims))
    assert {s.name: s.shape for s in specs} == shapes              # same keys / shapes as the reference
    # aligned, non-overlapping, ordered
    end = 0
    for s in specs:
        assert s.offset % ALIGN == 0 and s.offset >= end
        end = s.offset + s.numel
    assert end <= total and total % ALIGN == 0
    # EMA-shared range == reference teacher keys (data2vec2.py:377-381)
    shared = {s.name for s in specs if lo <= s.offset < hi}
    assert shared == set(O.teacher_keys(shapes))
    # weight-decay rule (data2vec2.py:318-320): 1-D, biases, alibi_scale, p_swish are NOT decayed; low_hz_/band_hz_ are
    for s in specs:
        nodecay = len(s.shape) == 1 or s.name.endswith(".bias") or "alibi_scale" in s.name or "p_swish" in s.name
        assert s.decay == (not nodecay), s.name
    assert next(s for s in specs if s.name.endswith("low_hz_")).decay


def test_buckets_cover_gradient_exactly_once():
    cfg = C.large_config()
    specs, _, total = build_specs(cfg)
    b = make_buckets(bucket_starts(cfg, specs), total)
    assert len(b) == 1 + cfg.depth + cfg.audio.prenet_depth + 3
    cover = np.zeros(total // ALIGN, dtype=np.int32)
    for lo, hi in b.values():
        assert lo % ALIGN == 0
        cover[lo // ALIGN:hi // ALIGN] += 1
    assert (cover == 1).all()
    # a block's bucket holds exactly that block's parameters
    lo, hi = b["block3"]
    names = [s.name for s in specs if lo <= s.offset < hi]
    assert names and all(n.startswith("blocks.3.") for n in names)
    # alibi_scale (accumulated by every attention backward) is released only after the prenet
    a = next(s for s in specs if s.name.endswith("alibi_scale"))
    assert b["posconv"][0] <= a.offset < b["posconv"][1]


def test_maskinfo_matches_oracle_and_reference_counts():
    m = O.seeded_mask(6, 200, 16, 3)
    mu, keep, restore = make_maskinfo(m)
    ok, orr = O.make_maskinfo(m)
    assert torch.equal(keep.long(), ok) and torch.equal(restore.long(), orr)
    assert torch.equal(seeded_keep_mask(6, 200, 16, 3), m)


def test_span_mask_distribution():
    # yaml:127-130 "mask_prob 1.5, mask_length 2 => ~93 % masked"; equal count per row (base.py:437)
    rng = np.random.default_rng(0)
    m = compute_mask_indices((24, 2000), 1.5, 2, rng)
    per_row = m.sum(1)
    assert (per_row == per_row[0]).all()
    frac = per_row[0] / 2000
    assert 0.90 < frac < 0.96


def test_grad_scale_factor_matches_fairseq_semantics():
    sumsq = torch.tensor(4.0e6, dtype=torch.float64)
    scale, gnorm = grad_scale_factor(sumsq, torch.tensor(1000.0, dtype=torch.float64), 1.0)
    assert float(gnorm) == pytest.approx(2.0)
    assert float(scale) == pytest.approx(1e-3 * 0.5, rel=1e-5)
    scale, gnorm = grad_scale_factor(torch.tensor(1.0, dtype=torch.float64), torch.tensor(10.0, dtype=torch.float64), 1.0)
    assert float(scale) == pytest.approx(0.1)


def test_registries_and_criterion():
    import nn
    assert "data2vec_multi" in nn.MODEL_REGISTRY and "audio_ccas" in nn.TASK_REGISTRY
    assert "expanded_model" in nn.CRITERION_REGISTRY

    class Fake(torch.nn.Module):
        def forward(self, source):
            return {"losses": {"AUDIO_regression": torch.tensor([3.0, 4.0])}, "sample_size": torch.tensor(7),
                    "pred_var": torch.tensor(0.5), "ema_decay": 999.7}

    crit = nn.ExpandedModelCriterion(None, None, ["pred_var", "ema_decay", "missing"])
    f = Fake()
    loss, ss, log = crit(f, {"net_input": {"source": torch.zeros(2, 8)}})
    assert float(loss) == 7.0 and int(ss) == 7 and log["pred_var"] == 0.5 and "missing" not in log
    assert crit.logging_outputs_can_be_summed()
    f.eval()
    crit(f, {"net_input": {"source": torch.zeros(2, 8)}})
    assert not crit.logging_outputs_can_be_summed()               # criterions.py:405-411


def test_yaml_mapping(tmp_path):
    from animal2vec_amd.train_routine import config_from_yaml
    y = tmp_path / "c.yaml"
    y.write_text("""
task: {conv_feature_layers: '[(127, 63, 1)] + [(512, 10, 5)]', sample_rate: 8000}
dataset: {max_tokens: 408000}
optimization: {update_freq: [5], max_update: 100, clip_norm: 1}
optimizer: {groups: {default: {lr_float: 0.0002, optimizer: {adam_betas: [0.9, 0.98], adam_eps: 1e-06, weight_decay: 0.01}, lr_scheduler: {warmup_updates: 10}}}}
model: {depth: 4, embed_dim: 256, num_heads: 4, clone_batch: 3, ema_decay: 0.999, unknown_key: 1,
        modalities: {audio: {prenet_depth: 2, conv_pos_groups: 4, decoder: {decoder_dim: 64, decoder_layers: 2}}}}
""")
    cfg, oc, run = config_from_yaml(str(y))
    assert (cfg.depth, cfg.embed_dim, cfg.audio.prenet_depth, cfg.audio.decoder.decoder_dim) == (4, 256, 2, 64)
    assert cfg.conv_layers == [(127, 63, 1), (512, 10, 5)]
    assert oc.lr == 0.0002 and oc.max_update == 100 and run["update_freq"] == 5


def test_finetune_schedule_and_layer_decay():
    """cosine schedule of finetune_mixup_100.yaml:69-74 and the layer_decay scales of wav2vec2.py:214-224"""
    from animal2vec_amd.finetune import FinetuneConfig, cosine_lr, layer_lr_scales
    f = FinetuneConfig()
    assert cosine_lr(f, 0) == pytest.approx(1e-10)
    assert cosine_lr(f, 1000) == pytest.approx(1e-10 + 1000 * (3e-5 - 1e-10) / 2000)
    assert cosine_lr(f, 2000) == pytest.approx(3e-5)
    assert cosine_lr(f, 16000) == pytest.approx(5e-6 + 0.5 * (3e-5 - 5e-6))
    assert cosine_lr(f, 29999) == pytest.approx(5e-6, rel=1e-3)
    lrs = [cosine_lr(f, n) for n in range(2000, 30000, 500)]
    assert all(a > b for a, b in zip(lrs, lrs[1:]))
    sc = layer_lr_scales(24, 0.75)
    assert len(sc) == 24 and sc[-1] == pytest.approx(0.75) and sc[0] == pytest.approx(0.75 ** 24)
    assert all(a < b for a, b in zip(sc, sc[1:]))


def test_finetune_aug_draws():
    """make_aug on the host: layerdrop frequencies, dropout settings (the span masks come from the device sampler:
    tests/test_masks_gpu.py::test_finetune_make_aug_draws_on_the_device_and_repeats_per_seed)"""
    import numpy as np
    from animal2vec_amd.finetune import FinetuneConfig, make_aug
    f = FinetuneConfig(apply_mask=False)
    rng = np.random.default_rng(0)
    B, T, D = 8, 2000, 1024
    drops = np.zeros(24)
    for _ in range(40):
        a = make_aug(f, B, T, D, 8, 24, rng, 0.01, device="cpu")
        assert "mask_t" not in a and "mask_c" not in a
        assert a["p_att"] == 0.2 and a["p_drop"] == 0.1 and a["p_act"] == 0.1
        assert any(a["active"][8:])
        drops += 1 - np.asarray(a["active"], dtype=float)
    assert 0.03 < drops.mean() / 40 < 0.2                                # layerdrop 0.1
    f0 = FinetuneConfig(apply_mask=False, layerdrop=0.0, dropout=0.0, activation_dropout=0.0, attention_dropout=0.0)
    a = make_aug(f0, B, T, D, 8, 24, rng, 0.01, device="cpu")
    assert "mask_t" not in a and "mask_c" not in a and all(a["active"]) and "drop_seed" not in a
    # with masks asked for and no GPU library call possible on CPU tensors, the product path fails loudly (no host fallback)
    with pytest.raises(Exception):
        make_aug(FinetuneConfig(), B, T, D, 8, 24, rng, 0.01, device="cpu")


def test_checkpoint_wire_format_roundtrip(tmp_path):
    """fairseq-style .pt: reference parameter names + "_ema" under "model"; pretrain resume, finetune load (with and
    without load_ema), finetuned-model layout (w2v_encoder.w2v_model.* / w2v_encoder.proj.*), alibi_scale upgrade."""
    import torch
    from animal2vec_amd import checkpoint as ck
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.params import ParamStore, P
    cfg = tiny_config()
    a = ParamStore(cfg, device="cpu", with_optimizer=False)
    a.init_reference(3)
    a.teacher.mul_(0.5)                                         # teacher != student
    path = tmp_path / "checkpoint_last.pt"
    st = ck.save_checkpoint(path, a, cfg={"model": {"_name": "data2vec_multi", "embed_dim": cfg.embed_dim}}, num_updates=7)
    raw = torch.load(path, weights_only=False)
    assert set(raw) >= {"cfg", "model", "optimizer_history", "extra_state", "last_optimizer_state"}
    assert raw["optimizer_history"][-1]["num_updates"] == 7
    assert "_ema" in raw["model"] and "blocks.0.attn.qkv.weight" in raw["model"]
    assert P + "decoder.proj.weight" in raw["model"] and raw["model"][P + "alibi_scale"].dim() == 5
    assert all(v.device.type == "cpu" for v in raw["model"].values() if torch.is_tensor(v))
    # resume
    b = ParamStore(cfg, device="cpu", with_optimizer=False)
    ck.load_pretrained(b, ck.load_checkpoint(path))
    assert torch.equal(a.flat, b.flat) and torch.equal(a.teacher, b.teacher)
    # old checkpoints carry a 4-d alibi_scale (base.py:152-157)
    raw["model"][P + "alibi_scale"] = raw["model"][P + "alibi_scale"].squeeze(0)
    torch.save(raw, path)
    c = ParamStore(cfg, device="cpu", with_optimizer=False)
    ck.load_pretrained(c, ck.load_checkpoint(path))
    assert torch.equal(a.flat, c.flat)
    # finetune load: decoder keys optional, _ema dropped; load_ema puts the teacher's weights into the encoder
    for k in [k for k in raw["model"] if ".decoder." in k]:
        del raw["model"][k]
    torch.save(raw, path)
    d = ParamStore(cfg, device="cpu", with_optimizer=False)
    ck.load_for_finetune(d, ck.load_checkpoint(path))
    assert torch.equal(d.p["blocks.0.attn.qkv.weight"], a.p["blocks.0.attn.qkv.weight"])
    e = ParamStore(cfg, device="cpu", with_optimizer=False)
    ck.load_for_finetune(e, ck.load_checkpoint(path), load_ema=True)
    assert torch.equal(e.p["blocks.0.attn.qkv.weight"], a.t["blocks.0.attn.qkv.weight"])
    assert not torch.equal(e.p["blocks.0.attn.qkv.weight"], a.p["blocks.0.attn.qkv.weight"])
    with pytest.raises(KeyError):
        del raw["model"]["blocks.0.attn.qkv.weight"]
        torch.save(raw, path)
        ck.load_for_finetune(ParamStore(cfg, device="cpu", with_optimizer=False), ck.load_checkpoint(path))

    class Head:                                                    # FinetuneHead's parameter surface
        def __init__(self):
            self.w, self.b = torch.randn(12, cfg.embed_dim), torch.randn(12)

        def load(self, w, b):
            self.w.copy_(w); self.b.copy_(b)

    h = Head()
    ck.save_checkpoint(path, a, head=h, num_updates=3)
    raw = torch.load(path, weights_only=False)
    assert "w2v_encoder.proj.weight" in raw["model"] and "w2v_encoder.w2v_model.blocks.1.mlp.fc2.bias" in raw["model"]
    assert not any(".decoder." in k for k in raw["model"]) and "_ema" not in raw["model"]
    f, h2 = ParamStore(cfg, device="cpu", with_optimizer=False), Head()
    ck.load_finetuned(f, h2, ck.load_checkpoint(path))
    assert torch.equal(h2.w, h.w) and torch.equal(f.p["blocks.1.mlp.fc2.bias"], a.p["blocks.1.mlp.fc2.bias"])


def test_checkpoint_with_foreign_config_objects(tmp_path):
    """Published checkpoints pickle omegaconf / argparse objects next to the tensors; omegaconf is not installed
    here, so the loader must still return the tensors and a readable config."""
    import pickle
    import sys
    import types
    import torch
    from animal2vec_amd import checkpoint as ck
    mod = types.ModuleType("fakeconf_pkg")

    class DictConfig:
        def __init__(self, content):
            self._content = content
            self._metadata = "meta"

    DictConfig.__module__ = "fakeconf_pkg"
    DictConfig.__qualname__ = "DictConfig"
    mod.DictConfig = DictConfig
    sys.modules["fakeconf_pkg"] = mod
    try:
        state = {"cfg": DictConfig({"model": DictConfig({"_name": "data2vec_multi", "depth": 16})}),
                 "model": {"blocks.0.norm1.weight": torch.arange(4.0), "_ema": {}}}
        path = tmp_path / "foreign.pt"
        torch.save(state, path)
    finally:
        del sys.modules["fakeconf_pkg"]
    got = ck.load_checkpoint(path)
    assert torch.equal(got["model"]["blocks.0.norm1.weight"], torch.arange(4.0))
    assert ck.config_to_dict(got["cfg"]) == {"model": {"_name": "data2vec_multi", "depth": 16}}


def test_finetune_criterion_label_smoothed_branch():
    """criterions.py:245-247 with use_focal_loss false: fairseq's label-smoothed NLL (eps_i = eps / (C - 1)), accuracy
    and confusion counts, on a stand-in model (the criterion only needs get_logits / get_targets)."""
    import torch
    from animal2vec_amd.finetune import FinetuneCrossEntropyCriterion, label_smoothed_nll_loss

    class M:
        def __call__(self, source):
            return {"encoder_out": source, "target": None}

        def get_logits(self, out, reshape=True):
            return out["encoder_out"].reshape(-1, out["encoder_out"].size(-1))

        def get_targets(self, sample, out, reshape=True):
            return sample["target"].reshape(-1)

    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, 7, 5, generator=g, requires_grad=True)
    target = torch.randint(0, 5, (2, 7), generator=g)
    crit = FinetuneCrossEntropyCriterion(use_focal_loss=False, label_smoothing=0.1, report_accuracy=True)
    loss, n, log = crit(M(), {"net_input": {"source": logits}, "target": target, "ntokens": 14})
    lp = torch.log_softmax(logits.detach().reshape(-1, 5), -1)
    nll = -lp[torch.arange(14), target.reshape(-1)].sum()
    smooth = -lp.sum()
    assert float(loss) == pytest.approx(float((1 - 0.1 - 0.1 / 4) * nll + 0.1 / 4 * smooth), rel=1e-6)
    assert float(log["nll_loss"]) == pytest.approx(float(nll), rel=1e-6) and n == 14
    loss.backward()
    assert logits.grad is not None and float(logits.grad.abs().sum()) > 0
    assert log["finetune/total"] == 14 and log["finetune/tp"] + log["finetune/fn"] == 14
    l0, _ = label_smoothed_nll_loss(lp, target.reshape(-1), 0.0)
    assert float(l0) == pytest.approx(float(nll), rel=1e-6)


def test_finetune_yaml_mapping_shipped_recipes(tmp_path):
    """configs/MeerKAT/finetune_mixup_{001,025,100}.yaml -> FinetuneConfig: the three recipes differ in lr, max_update,
    mask_prob and mask_length (and in control-plane cadence); a trainer built from the 025 recipe must not run the 100
    recipe's hyper-parameters.  Inputs: the recipes' parsed sections (tests/golden/finetune_recipes.json, dumped by
    oracle/gen_golden.py::finetune_recipes_fixture)."""
    import json
    import yaml
    from animal2vec_amd.finetune import cosine_lr
    from animal2vec_amd.train_routine import finetune_config_from_yaml
    from conftest import GOLD
    recipes = json.load(open(os.path.join(GOLD, "finetune_recipes.json")))
    want = {"finetune_mixup_100": dict(lr=3e-5, max_update=30000, mask_prob=0.825, mask_length=4, train_subset="train_0"),
            "finetune_mixup_025": dict(lr=7.5e-5, max_update=18000, mask_prob=1.0, mask_length=3, train_subset="train_0_few_2"),
            "finetune_mixup_001": dict(lr=1e-4, max_update=13000, mask_prob=1.1, mask_length=2, train_subset="train_0_few_0")}
    for name, w in want.items():
        p = tmp_path / (name + ".yaml")
        p.write_text(yaml.safe_dump(recipes[name]))
        f, run = finetune_config_from_yaml(str(p))
        assert (f.lr, f.max_update, f.mask_prob, f.mask_length) == (w["lr"], w["max_update"], w["mask_prob"], w["mask_length"]), name
        assert run["train_subset"] == w["train_subset"]
        # common to the three (model / optimizer / scheduler blocks)
        assert f.n_labels == 12 and run["unique_labels"][-1] == "focal" and len(run["unique_labels"]) == 12
        assert (f.apply_mask, f.mask_channel_prob, f.mask_channel_length) == (True, 0.5, 64)
        assert (f.dropout, f.activation_dropout, f.attention_dropout, f.dropout_input, f.final_dropout, f.layerdrop) == \
            (0.1, 0.1, 0.2, 0.0, 0.0, 0.1)
        assert f.freeze_finetune_updates == 10000 and f.target_mixup and f.source_mixup == 0.5 and f.mixup_prob == 1.0
        assert f.adam_betas == (0.9, 0.98) and f.adam_eps == 1e-8 and f.weight_decay == 0.0 and f.clip_norm == 0.0
        assert (f.warmup_updates, f.warmup_init_lr, f.min_lr) == (2000, 1e-10, 5e-6)
        assert (run["max_tokens"], run["update_freq"], run["average_top_k_layers"]) == (426667, 9, 16)
        assert run["min_label_size"] == 3032 and run["metric_threshold"] == 0.175 and run["use_focal_loss"] is True
        assert abs(cosine_lr(f, f.warmup_updates) - f.lr) < 1e-12 and f.min_lr <= cosine_lr(f, f.max_update - 1) < f.min_lr * 1.001
    ref_dir = "/root/reference/configs/MeerKAT"
    if os.path.isdir(ref_dir):                       # in the build container: the files themselves
        f, run = finetune_config_from_yaml(os.path.join(ref_dir, "finetune_mixup_025.yaml"))
        assert (f.lr, f.max_update, f.mask_prob, f.mask_length) == (7.5e-5, 18000, 1.0, 3)


def test_reduce_metrics_of_both_criterions_and_the_task():
    """ExpandedModelCriterion.reduce_metrics (criterions.py:413-502), FinetuneCrossEntropyCriterion.reduce_metrics
    (:279-380) and AudioTaskCCAS.reduce_metrics (audio_tasks.py:154-188) on two workers' logging outputs."""
    import math
    import nn
    from animal2vec_amd import metrics
    metrics.reset()
    logs = [{"loss": torch.tensor(30.0), "ntokens": 10, "nsentences": 2, "sample_size": 10, "_world_size": 1, "pred_var": 0.5,
             "loss_AUDIO_regression": torch.tensor(30.0), "ema_decay": 999.7},
            {"loss": torch.tensor(10.0), "ntokens": 30, "nsentences": 6, "sample_size": 30, "_world_size": 1, "pred_var": 0.3,
             "loss_AUDIO_regression": torch.tensor(10.0), "ema_decay": 999.7}]
    crit = nn.ExpandedModelCriterion(None, None, ["pred_var", "ema_decay"])
    task = nn.AudioTaskCCAS.setup_task(nn.AudioConfigCCAS())
    task.reduce_metrics(logs, crit)
    m = metrics.get_smoothed_values()
    assert m["loss"] == 1.0 and m["loss_AUDIO_regression"] == 1.0 and m["sample_size"] == 40 and m["ntokens"] == 40
    assert m["pred_var"] == 0.4 and m["ema_decay"] == 999.7            # mean over workers (/ sum of _world_size)
    assert m["misc/wpb"] == 40 and m["misc/bsz"] == 8 and "misc/wps" in m
    # a second call aggregates as fairseq's meters do: loss weighted by sample_size
    task.reduce_metrics([{"loss": 120.0, "ntokens": 40, "nsentences": 8, "sample_size": 40, "_world_size": 1, "pred_var": 0.2,
                          "loss_AUDIO_regression": 120.0, "ema_decay": 999.8}], crit)
    assert metrics.get_smoothed_values()["loss"] == 2.0
    # finetune criterion: bits, counts and the derived scores
    metrics.reset()
    flogs = [{"loss": 8.0 * math.log(2), "nll_loss": 0.0, "ntokens": 4, "nsentences": 1, "sample_size": 4, "finetune/n_correct": 30.0,
              "finetune/total": 48.0, "finetune/tp": 6, "finetune/fp": 2, "finetune/tn": 36, "finetune/fn": 4},
             {"loss": 4.0 * math.log(2), "nll_loss": 0.0, "ntokens": 4, "nsentences": 1, "sample_size": 4, "finetune/n_correct": 40.0,
              "finetune/total": 48.0, "finetune/tp": 2, "finetune/fp": 2, "finetune/tn": 42, "finetune/fn": 2}]
    nn.FinetuneCrossEntropyCriterion.reduce_metrics(flogs)
    m = metrics.get_smoothed_values()
    assert m["loss"] == 1.5 and m["misc/ntokens"] == 8 and m["misc/sample_size"] == 8
    # (the reference logs the non-builtin keys as means over the logging outputs; the derived scores use meter SUMS)
    assert m["finetune/tp"] == 4.0
    # "finetune/total" is logged twice per call by the reference (the sum of the absent "count" key = 0, then the mean of
    # the logged totals): its meter averages to 24 while its SUM -- what the derived scores read -- is 48
    assert m["finetune/total"] == 24.0
    assert m["metrics/finetune/accuracy"] == round(35.0 * 100 / 48.0, 3)
    assert m["metrics/finetune/precision"] == round(4 * 100.0 / 6, 3) and m["metrics/finetune/recall"] == round(4 * 100.0 / 7, 3)


def test_init_laws_match_the_reference_model(tmp_path):
    """Rows a1 / a21: ParamStore.init_reference against the LIVE reference model's initial state (mid dims: 127 x 63 sinc,
    512-wide conv stack, D = 128; tests/golden/init_stats.npz from oracle/gen_golden.py::init_fixture).  Deterministic
    inits (mel-spaced SincConv band edges, LayerNorm 1 / 0, PSwish 2 / 0, alibi_scale 1, zero biases) are equal value by
    value; every random tensor follows the same law: same support for the uniform ones (abs-max), same standard
    deviation and zero mean within sampling error -- for each of three seeds on both sides."""
    from conftest import load_golden, rel_err
    from animal2vec_amd.config import D2vAudioConfig, D2vDecoderConfig, Data2VecMultiConfig
    from animal2vec_amd.params import ParamStore, sinc_mel_init
    fx = load_golden("init_stats.npz")
    a = D2vAudioConfig(conv_pos_groups=16, prenet_depth=1, decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=16, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=8, clone_batch=2, average_top_k_layers=2, audio=a)
    names = sorted(k[3:-6] for k in fx if k.startswith("s0.") and k.endswith(".stats"))
    st = ParamStore(cfg, device="cpu", with_optimizer=False)
    assert sorted(s.name for s in st.specs) == names
    for seed in (0, 1, 2):
        st.init_reference(seed)
        for n in names:
            v = st.p[n].float()
            numel, mean, std, amax = (float(x) for x in fx[f"s{seed}.{n}.stats"])
            assert v.numel() == int(numel), n
            if f"value.{n}" in fx:                                   # deterministic
                assert torch.allclose(v.reshape(-1), fx[f"value.{n}"].reshape(-1), rtol=1e-6, atol=1e-6), n
                continue
            assert std > 0, n
            se = 1.0 / math.sqrt(2 * numel)                          # relative standard error of a sample std
            assert abs(float(v.std()) / std - 1) < 6 * math.sqrt(2) * se + 1e-3, (n, float(v.std()), std)
            assert abs(float(v.mean()) - mean) < 6 * math.sqrt(2) * std / math.sqrt(numel), n
            uniform = amax < 2.0 * std                               # a uniform law: abs-max = sqrt(3) std; a normal one: > 3 std
            if uniform:
                assert abs(float(v.abs().max()) / amax - 1) < 0.03, n
            else:
                assert float(v.abs().max()) > 2.5 * std, n
    # a1 on the other front-end specs of tests/golden/sinc.npz (8 kHz tiny, 44.1 kHz, the blocked 296 x 149 one)
    sx = load_golden("sinc.npz")
    for spec in ("127_63_8000", "127_63_44100", "296_149_44100", "31_15_8000"):
        c, k, sr = (int(x) for x in spec.split("_"))
        lo, band = sinc_mel_init(c, k, sr)
        assert torch.allclose(lo, sx[f"init.{spec}.low"], rtol=1e-6, atol=1e-4) and torch.allclose(band, sx[f"init.{spec}.band"], rtol=1e-6, atol=1e-4)
    # window / time axis buffers (sinc.py:264-276): the init filters of the golden follow from them (sinc.py:181-223)
    for spec, layers in (("127_63_8000", None), ("31_15_8000", "[(31, 15, 1)] + [(32, 10, 5)] + [(32, 3, 2)] * 3 + [(32, 3, 1)] + [(32, 2, 1)] * 2")):
        c, k, sr = (int(x) for x in spec.split("_"))
        cfg2 = cfg if layers is None else Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=8, audio=D2vAudioConfig(
            conv_feature_layers=layers, conv_pos_groups=16, prenet_depth=1, decoder=D2vDecoderConfig(decoder_dim=64, decoder_layers=2)))
        s2 = ParamStore(cfg2, device="cpu", with_optimizer=False)
        low = 50.0 + sx[f"init.{spec}.low"].abs()
        high = torch.clamp(low + s2.sinc_min_band + sx[f"init.{spec}.band"].abs(), 50.0, sr / 2)
        bandw = (high - low)[:, 0]
        left = ((torch.sin(high * s2.sinc_n) - torch.sin(low * s2.sinc_n)) / (s2.sinc_n / 2)) * s2.sinc_window
        filt = torch.cat([left, 2 * bandw.view(-1, 1), left.flip(1)], dim=1) / (2 * bandw[:, None])
        assert rel_err(filt, sx[f"init.{spec}.filters"].view(c, k)) < 1e-5, spec


def test_oracle_span_mask_follows_the_law_of_compute_mask_indices():
    """oracle/a2v_oracle.py::span_mask restates the DEVICE sampler's counter stream (the GPU tests hold the kernel to it bit
    for bit); here its law is compared with the numpy statement of fairseq's compute_mask_indices: equal counts per row,
    per-frame masking frequency, kept count."""
    import a2v_oracle as O
    N, T, prob, length = 1024, 200, 0.65, 5
    dm = O.span_mask(N, T, prob, length, seed=11)
    hm = compute_mask_indices((N, T), prob, length, np.random.default_rng(5))
    assert len(set(dm.sum(1).tolist())) == 1 and len(set(hm.sum(1).tolist())) == 1
    fd, fh = dm.mean(0), hm.mean(0)
    sigma = (np.maximum(fh * (1 - fh), 0.02) * 2 / N) ** 0.5
    assert np.all(np.abs(fd - fh) < 5 * sigma)
    assert abs(int(dm[0].sum()) - int(hm[0].sum())) <= 4
    big = O.span_mask(96, 2000, 1.5, 2, seed=3)                     # the large recipe: ~93 % masked, K in SURVEY 8c's band
    assert 125 <= 2000 - int(big[0].sum()) <= 156
    assert np.array_equal(big, O.span_mask(96, 2000, 1.5, 2, seed=3))


def test_mask_seed_is_a_distinct_function_of_every_counter():
    from animal2vec_amd.masking import mask_seed
    seen = {mask_seed(s, u, j, r) for s in (0, 1) for u in range(50) for j in range(8) for r in range(8)}
    assert len(seen) == 2 * 50 * 8 * 8
    assert all(0 <= v < 2 ** 64 for v in seen)


def test_device_sampler_refuses_laws_it_does_not_implement():
    """add_masks / mask_length == 1 select other sampling laws in the reference (base.py:395-396, 409); with no host
    path left in the step they must be rejected, not sampled with the span law silently."""
    from animal2vec_amd.masking import check_span_law
    check_span_law(mask_length=2, add_masks=False)
    for kw in (dict(mask_length=1), dict(mask_length=2, add_masks=True), dict(mask_length=4, inverse_mask=True),
               dict(mask_length=4, keep_masked_pct=0.1)):
        with pytest.raises(NotImplementedError):
            check_span_law(**kw)


def test_mask_seed_is_a_pure_function_of_the_counters():
    from animal2vec_amd.masking import mask_seed
    a = mask_seed(1, 10, 0, 0)
    assert a == mask_seed(1, 10, 0, 0)
    assert len({a, mask_seed(1, 11, 0, 0), mask_seed(1, 10, 1, 0), mask_seed(1, 10, 0, 1), mask_seed(2, 10, 0, 0)}) == 5


def test_unbuilt_reference_switches_are_refused(tmp_path):
    """VERDICT r4 missing #3: a drop-in refuses what it does not implement.  Every reference switch the hot path does not build
    raises at construction (dataclass, YAML recipe) instead of being carried and ignored; the shipped recipe itself loads."""
    import copy
    import pytest
    import yaml
    from animal2vec_amd.config import D2vAudioConfig, Data2VecMultiConfig, large_config
    from animal2vec_amd.train_routine import config_from_yaml
    for kw in (dict(loss_beta=0.5), dict(loss_scale=1.0), dict(instance_norm_target_layer=False), dict(layer_norm_target_layer=True),
               dict(layer_norm_targets=True), dict(instance_norm_targets=True), dict(batch_norm_target_layer=True),
               dict(with_labels=True), dict(ema_encoder_only=True), dict(end_of_block_targets=True), dict(activation_dropout=0.1)):
        with pytest.raises(NotImplementedError):
            large_config(**kw)
    for kw in (dict(inverse_mask=True), dict(mask_prob_min=0.1), dict(ema_local_encoder=True), dict(add_masks=True), dict(use_pswish=False)):
        with pytest.raises(NotImplementedError):
            D2vAudioConfig(**kw)
    assert Data2VecMultiConfig().loss_beta == 0.0 and large_config(loss_beta=0).loss_scale is None
    ref = "/root/reference/configs/MeerKAT/a2v_large_pretrain_best.yaml"
    if os.path.exists(ref):                    # (build container only: the GPU box has no reference tree)
        cfg, _, _ = config_from_yaml(ref)
        assert cfg.depth == 16 and cfg.embed_dim == 1024 and cfg.audio.mask_prob == 1.5
        y = yaml.safe_load(open(ref))
        for sect, k, v in (("model", "loss_beta", 0.1), ("model", "layer_norm_targets", True), ("model", "loss_scale", 2.0),
                           ("task", "with_labels", True)):
            y2 = copy.deepcopy(y)
            y2[sect][k] = v
            p = tmp_path / f"{k}.yaml"
            p.write_text(yaml.safe_dump(y2))
            with pytest.raises(NotImplementedError):
                config_from_yaml(str(p))


def test_pretrain_trainer_refuses_layerdrop():
    import pytest
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.trainer import PretrainTrainer
    with pytest.raises(NotImplementedError):
        PretrainTrainer(tiny_config(layerdrop=0.1), device="cpu")


def test_finetune_span_lengths_are_checked_and_clamped():
    """both mask laws are validated before the device sampler is asked (span length 1 = another law in the reference),
    and spans that do not fit the axis are clamped as fairseq's compute_mask_indices clamps them, not failed in the kernel."""
    import pytest
    from animal2vec_amd.finetune import FinetuneConfig, span_lengths
    f = FinetuneConfig()
    assert span_lengths(f, 2000, 1024) == (f.mask_length, f.mask_channel_length)
    assert span_lengths(f, 4, 32) == (min(f.mask_length, 3), 31)
    with pytest.raises(NotImplementedError):
        span_lengths(FinetuneConfig(mask_length=1), 2000, 1024)
    with pytest.raises(NotImplementedError):
        span_lengths(FinetuneConfig(mask_channel_length=1), 2000, 1024)
