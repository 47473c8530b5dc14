"""The checker behind __graft_entry__.smoke(): one small end-to-end invocation of the hot path on cuda:0 (student
fwd+bwd, EMA-teacher targets, loss) through the HIP library, compared with the CPU oracle.  Test infrastructure: it
lives under tests/ (not in the package) because it imports the oracle; the product package never does
(tests/test_abi.py::test_product_never_imports_oracle walks the package's syntax trees)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def run():
    import a2v_oracle as O                                   # the checker
    from animal2vec_amd.config import D2vAudioConfig, D2vDecoderConfig, Data2VecMultiConfig
    from animal2vec_amd.engine import Engine
    from animal2vec_amd.masking import make_maskinfo
    from animal2vec_amd.params import ParamStore

    dims = dict(embed_dim=128, num_heads=2, depth=2, prenet_depth=1, decoder_dim=64, decoder_layers=2,
                decoder_kernel=7, decoder_groups=8, clone_batch=2, conv_pos_groups=8, average_top_k_layers=2)
    ocfg = O.OracleConfig(**dims)
    p = O.init_params(ocfg, seed=1)
    ema = O.make_teacher(p)
    g = torch.Generator().manual_seed(2)
    B, L, keep = 2, 8000, 16
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    src = torch.nn.functional.layer_norm(torch.randn(B, L, generator=g), (L,))
    mask = O.seeded_mask(B * ocfg.clone_batch, T, keep, 3)
    noise = torch.randn(B * ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
    ref["losses"]["AUDIO_regression"].sum().backward()

    a = D2vAudioConfig(conv_pos_groups=8, prenet_depth=1, decoder=D2vDecoderConfig(decoder_dim=64, decoder_groups=8, decoder_layers=2))
    cfg = Data2VecMultiConfig(depth=2, embed_dim=128, num_heads=2, clone_batch=2, average_top_k_layers=2, audio=a)
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, 5e-2)):
        store = ParamStore(cfg, device="cuda", with_optimizer=False)
        store.load_state_dict(p, ema)
        eng = Engine(cfg, store, dtype=dtype)
        m, k_, r_ = make_maskinfo(mask.cuda())
        eng.stats.zero_()
        eng.forward_backward(src.cuda(), m, k_, r_, noise.cuda())
        torch.cuda.synchronize()
        D = cfg.embed_dim
        ls, lr = float(eng.stats[4 * D]), float(ref["losses"]["AUDIO_regression"].sum().detach())
        gq = "blocks.1.attn.qkv.weight"
        ge = float((store.g[gq].cpu() - po[gq].grad).norm() / po[gq].grad.norm())
        assert abs(ls - lr) / lr < tol, (dtype, ls, lr)
        assert ge < (1e-3 if dtype == torch.float32 else 0.15), (dtype, ge)
        print(f"smoke ok [{dtype}]: loss {ls:.4f} vs oracle {lr:.4f}; grad rel err {ge:.2e}")
    # the step's span masks come from the device sampler: bit-exact against the oracle's restatement of its stream
    from animal2vec_amd import ops
    mk, keep_flat, restore, kdev, _ = ops.span_mask(24, 500, mask_prob=1.5, mask_length=2, seed=9)
    torch.cuda.synchronize()
    ref_mask = torch.from_numpy(O.span_mask(24, 500, 1.5, 2, 9))
    K = int(kdev)
    ik, ir = O.make_maskinfo(ref_mask)
    assert torch.equal(mk.cpu().bool(), ref_mask) and K == ik.shape[1]
    assert torch.equal(keep_flat[:24 * K].view(24, K).cpu().long(), ik) and torch.equal(restore.cpu().long(), ir)
    print(f"smoke ok [span masks]: 24 x 500 frames, K = {K}, bit-exact vs the oracle")
