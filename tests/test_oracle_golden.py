"""CPU: the oracle (oracle/a2v_oracle.py) against the golden vectors dumped from the LIVE reference
by oracle/gen_golden.py.  These are the tests that pin the oracle."""
import math
import os

import pytest
import torch

import a2v_oracle as O
from conftest import load_golden, rel_err, split_golden

TINY_LAYERS = [(31, 15, 1), (32, 10, 5)] + [(32, 3, 2)] * 3 + [(32, 3, 1)] + [(32, 2, 1)] * 2


def tiny_cfg():
    return O.OracleConfig(**O.TINY, conv_feature_layers=TINY_LAYERS)


def test_frame_count_chain():
    # SURVEY section 4 known answer: 80000 -> 80000 -> 16000 -> 8000 -> 4000 -> 2000 x4
    assert O.frame_counts(80000, O.LARGE_LAYERS) == [80000, 16000, 8000, 4000, 2000, 2000, 2000, 2000]
    assert O.frame_counts(8000, TINY_LAYERS)[-1] == 200


def test_sinc_known_answers():
    # nn/sinc.py:225-276 is RNG-free: centre tap exactly 1, row 0 starts with the surveyed values
    lo, band = O.sinc_init(127, 63, 8000)
    f = O.sinc_filters(lo, band, 63, 8000)
    assert f.shape == (127, 1, 63)
    assert torch.all(f[:, 0, 31] == 1.0)
    assert torch.allclose(f[0, 0, :5], torch.tensor([-0.0258, -0.0326, -0.0419, -0.0543, -0.0704]), atol=1e-4)
    assert int(math.ceil(8000 / 63)) == 127 and int(math.ceil(44100 / 63)) == 700


@pytest.mark.parametrize("tag", ["127_63_8000", "127_63_44100", "296_149_44100", "31_15_8000"])
def test_sinc_filters_golden(tag):
    g = load_golden("sinc.npz")
    C, k, sr = (int(v) for v in tag.split("_"))
    lo, band = O.sinc_init(C, k, sr)
    assert torch.equal(lo, g[f"init.{tag}.low"]) and torch.equal(band, g[f"init.{tag}.band"])
    assert rel_err(O.sinc_filters(lo, band, k, sr), g[f"init.{tag}.filters"]) < 1e-6
    assert rel_err(O.sinc_filters(g[f"pert.{tag}.low"], g[f"pert.{tag}.band"], k, sr), g[f"pert.{tag}.filters"]) < 1e-6


@pytest.mark.parametrize("tag", ["127_63_8000", "31_15_8000"])
def test_sinc_conv_golden(tag):
    g = load_golden("sinc.npz")
    C, k, sr = (int(v) for v in tag.split("_"))
    y = O.sinc_conv(g[f"conv.{tag}.x"], g[f"pert.{tag}.low"], g[f"pert.{tag}.band"], k, sr)
    assert rel_err(y, g[f"conv.{tag}.y"]) < 1e-6


def test_alibi_slopes():
    # base.py:559-576: H=16 -> 2^(-0.5 i), i=1..16 ; non power of two uses the interleave rule
    s = O.alibi_slopes(16)
    assert all(abs(s[i] - 2 ** (-0.5 * (i + 1))) < 1e-12 for i in range(16))
    assert len(O.alibi_slopes(12)) == 12 and abs(O.alibi_slopes(12)[8] - 2 ** -0.5) < 1e-12


def test_annealed_rate():
    assert O.get_annealed_rate(0.9997, 1.0, 0, 300000) == pytest.approx(0.9997)
    assert O.get_annealed_rate(0.9997, 1.0, 150000, 300000) == pytest.approx(0.99985)
    assert O.get_annealed_rate(0.9997, 1.0, 300000, 300000) == 1.0


def test_param_counts():
    # 315 830 026 (large student), 122 780 678 (base*)
    for dims, n in ((O.LARGE, 315_830_026), (O.BASE_STAR, 122_780_678)):
        cfg = O.OracleConfig(**dims)
        shapes = O.init_param_shapes(cfg) if hasattr(O, "init_param_shapes") else None
        if shapes is None:
            pytest.skip("shape-only counter not available")
        assert sum(math.prod(s) for s in shapes.values()) == n


def test_pretrain_forward_backward_golden(tiny_golden):
    fx = tiny_golden
    p, ema, g = split_golden(fx)
    cfg = tiny_cfg()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    res = O.pretrain_forward(po, ema, fx["source"], fx["mask"], fx["noise"], cfg)
    loss = res["losses"]["AUDIO_regression"]
    assert rel_err(loss, fx["loss"]) < 1e-5
    assert rel_err(res["_dbg"]["targets"], fx["targets"]) < 1e-5
    assert rel_err(res["_dbg"]["local_features"], fx["local_features"]) < 1e-5
    assert int(res["sample_size"]) == int(fx["sample_size"])
    assert rel_err(res["pred_var"], fx["pred_var"]) < 1e-5
    assert rel_err(res["target_var"], fx["target_var"]) < 1e-5
    loss.sum().backward()
    for k, gr in g.items():
        assert rel_err(po[k].grad, gr) < 5e-4, k


def test_ema_step_golden(tiny_golden):
    p, ema, _ = split_golden(tiny_golden)
    ex = load_golden("tiny_ema.npz")
    cfg = tiny_cfg()
    d = O.ema_decay_at(cfg, 1)
    assert d == pytest.approx(float(ex["decay"]), abs=1e-15)
    ema = {k: v.clone() for k, v in ema.items()}
    O.ema_step(p, ema, d)
    for k, v in ema.items():
        assert rel_err(v, ex["after." + k]) < 1e-6, k


def test_mixup_golden():
    mx = load_golden("mixup.npz")
    assert rel_err(O.compute_gain(mx["source"], 8000, 0.05), mx["gain_db"]) < 1e-6
    assert rel_err(O.bc_mixup(mx["source"], mx["perm"], float(mx["r"])), mx["mixed"]) < 1e-6


def test_init_params_shapes_match_reference_keys(tiny_golden):
    p, ema, _ = split_golden(tiny_golden)
    cfg = tiny_cfg()
    q = O.init_params(cfg, seed=0)
    assert set(q) == set(p)
    for k in p:
        assert q[k].shape == p[k].shape, k
    assert set(O.teacher_keys(q)) == set(ema)


def test_features_only_golden(tiny_golden):
    p, _, _ = split_golden(tiny_golden)
    x, lrs = O.features_only_forward(p, tiny_golden["source"], tiny_cfg())
    assert rel_err(x, tiny_golden["features_only.x"]) < 1e-5
    assert rel_err(sum(lrs) / len(lrs), tiny_golden["features_only.layer_mean"]) < 1e-5


def test_finetune_path_golden(tiny_golden):
    """G15: logits, focal loss and every gradient of the finetune path (wav2vec2.py:430-482 in eval mode with the
    conv feature extractor frozen, criterions.py:243-248) against the live reference's extract_features +
    sigmoid_focal_loss."""
    ft = load_golden("tiny_finetune.npz")
    p, _, _ = split_golden(tiny_golden)
    cfg = tiny_cfg()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    hw, hb = ft["head_w"].clone().requires_grad_(True), ft["head_b"].clone().requires_grad_(True)
    logits = O.finetune_forward(po, hw, hb, ft["source"], cfg)
    loss = O.sigmoid_focal_loss(logits, ft["target"], reduction="sum")
    loss.backward()
    assert rel_err(logits, ft["logits"]) < 1e-5
    assert abs(float(loss) - float(ft["loss_sum"])) / float(ft["loss_sum"]) < 1e-5
    assert rel_err(hw.grad, ft["g.head_w"]) < 1e-4 and rel_err(hb.grad, ft["g.head_b"]) < 1e-4
    gk = [k[2:] for k in ft if k.startswith("g.") and not k.startswith("g.head_")]
    assert len(gk) > 40
    for k in gk:
        assert rel_err(po[k].grad, ft["g." + k]) < 5e-4, k
    # the frozen front end and the last block's unused norm2 carry no gradient, as in the reference
    assert all(po[k].grad is None for k in po if "local_encoder" in k)
    assert set(k for k in po if po[k].grad is not None) == set(gk)


def test_finetune_train_form_golden(tiny_golden):
    """G16: the train form of the finetune forward -- noise-replace time masking, channel masking (base.py:457-485)
    and layerdrop (modules.py:89-94, data2vec2.py:651-655) -- against the live reference run in train() mode with the
    random draws pinned (oracle/gen_golden.py::finetune_train_fixture)."""
    ft = load_golden("tiny_finetune_train.npz")
    p, _, _ = split_golden(tiny_golden)
    cfg = tiny_cfg()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    hw, hb = ft["head_w"].clone().requires_grad_(True), ft["head_b"].clone().requires_grad_(True)
    active = [bool(v) for v in ft["active"]]
    assert not all(active)
    logits = O.finetune_forward(po, hw, hb, ft["source"], cfg, mask_t=ft["mask_t"], replacement=ft["replacement"],
                                mask_c=ft["mask_c"], active=active)
    loss = O.sigmoid_focal_loss(logits, ft["target"], reduction="sum")
    loss.backward()
    assert rel_err(logits, ft["logits"]) < 1e-5
    assert abs(float(loss) - float(ft["loss_sum"])) / float(ft["loss_sum"]) < 1e-5
    assert rel_err(hw.grad, ft["g.head_w"]) < 1e-4 and rel_err(hb.grad, ft["g.head_b"]) < 1e-4
    n = 0
    for k in ft:
        if not k.startswith("g.") or k.startswith("g.head_"):
            continue
        ref = ft[k]
        got = po[k[2:]].grad
        if got is None:
            assert float(ref.abs().max()) == 0.0, k          # a dropped block
            continue
        assert rel_err(got, ref) < 5e-4, k
        n += 1
    assert n > 25
    # the masks really act: masked frames / channels of the encoder input carry no gradient into project_features' input
    x = O.apply_mask(torch.ones(2, 5, 4), torch.tensor([[1, 0, 0, 0, 1], [0, 0, 0, 0, 0]]), torch.full((2, 4), 7.0),
                     torch.tensor([[0, 1, 0, 0], [0, 0, 0, 1]]))
    assert x[0, 0].tolist() == [7.0, 0.0, 7.0, 7.0] and x[0, 1].tolist() == [1.0, 0.0, 1.0, 1.0] and x[1, 2].tolist() == [1.0, 1.0, 1.0, 0.0]


def test_focal_loss_known_values():
    # nn/utils.py:971-1010 at logit 0: p = 0.5, ce = ln 2, (1 - p_t)^2 = 0.25, alpha_t = 0.25 (positive) / 0.75
    l = O.sigmoid_focal_loss(torch.zeros(2), torch.tensor([1.0, 0.0]))
    assert l[0].item() == pytest.approx(0.25 * 0.25 * math.log(2), rel=1e-6)
    assert l[1].item() == pytest.approx(0.75 * 0.25 * math.log(2), rel=1e-6)


def test_trained_fixture_is_the_oracles(tmp_path):
    """tests/golden/tiny_trained.npz (oracle/gen_golden.py --trained): the stored outputs and gradients are what the oracle computes
    on the stored weights and batch -- the off-init GPU test compares the HIP path with exactly these numbers."""
    import json
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_trained.npz"))
    dims = json.loads(bytes(z["dims"]).decode())
    ocfg = O.OracleConfig(conv_feature_layers=[tuple(int(x) for x in r) for r in z["layers"]], **dims)
    p = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("p.")}
    ema = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("ema.")}
    res = O.pretrain_forward(p, ema, torch.from_numpy(z["source"]), torch.from_numpy(z["mask"]), torch.from_numpy(z["noise"]), ocfg)
    res["losses"]["AUDIO_regression"].sum().backward()
    assert abs(float(res["losses"]["AUDIO_regression"].sum()) - float(z["loss_sum"])) < 1e-3 * float(z["loss_sum"])
    for k in ("local_features", "targets", "decoder_out"):
        a, b = res["_dbg"][k].detach().double(), torch.from_numpy(z["out." + k]).double()
        assert float((a - b).norm() / b.norm()) < 1e-5, k
    worst = max(float((p[k].grad.double() - torch.from_numpy(z["g." + k]).double()).norm() / (torch.from_numpy(z["g." + k]).double().norm() + 1e-30))
                for k in p)
    assert worst < 1e-4, worst
    assert float(z["loss_last"]) < 0.25 * float(z["loss_first"])          # the state is far from init
