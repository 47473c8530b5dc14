#!/usr/bin/env python3
"""Per-tensor error table of the bf16 throughput mode at BASELINE.json's full size (test infrastructure, GPU box).

For one 10 s clip (large: D 1024, 8 + 16 blocks, clone_batch 12, 140 / 2000 frames kept) it runs
  * the oracle on the host CPU in fp32                                 (the pinned restatement of the reference),
  * optionally the oracle under torch.autocast("cpu", bfloat16)        (what stock bf16 autocast of the reference loses),
  * the HIP engine in fp32 mode and in bf16 mode,
and writes, for local features, teacher targets, decoder output, loss and EVERY parameter gradient:
  ||bf16 - oracle|| / ||oracle||, ||fp32 - oracle|| / ||oracle||, ||bf16 - fp32 engine|| / ||fp32 engine||, the tensor's
  norm and size, and (with --autocast) the autocast oracle's error.

  python tools/bf16_parity.py --model large --out gpurun_out/r02_bf16_parity.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch


def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large", choices=["large", "base"])
    ap.add_argument("--out", default="gpurun_out/r02_bf16_parity.json")
    ap.add_argument("--autocast", action="store_true")
    ap.add_argument("--keep", type=int, default=140)
    ap.add_argument("--samples", type=int, default=80000)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import a2v_oracle as O
    from animal2vec_amd.config import base_star_config, large_config
    from animal2vec_amd.engine import Engine
    from animal2vec_amd.masking import make_maskinfo
    from animal2vec_amd.params import ParamStore

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    dims = O.LARGE if args.model == "large" else O.BASE_STAR
    ocfg = O.OracleConfig(**dims)
    cfg = large_config() if args.model == "large" else base_star_config()
    p = O.init_params(ocfg, seed=args.seed)
    g = torch.Generator().manual_seed(9)
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.002 for k, v in O.make_teacher(p).items()}
    L, keep = args.samples, args.keep
    T = O.frame_counts(L, ocfg.conv_feature_layers)[-1]
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,))
    mask = O.seeded_mask(ocfg.clone_batch, T, keep, 11)
    noise = torch.randn(ocfg.clone_batch, T - keep, ocfg.embed_dim, generator=g) * 0.01

    def oracle(autocast):
        po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        t0 = time.perf_counter()
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
                res["losses"]["AUDIO_regression"].float().sum().backward()
        else:
            res = O.pretrain_forward(po, ema, src, mask, noise, ocfg)
            res["losses"]["AUDIO_regression"].sum().backward()
        out = {"local_features": res["_dbg"]["local_features"].detach().float(), "targets": res["_dbg"]["targets"].detach().float(),
               "decoder_out": res["_dbg"]["decoder_out"].detach().float(),
               "loss_sum": res["losses"]["AUDIO_regression"].float().sum().detach().reshape(1)}
        out.update({"g." + k: v.grad.detach().float() for k, v in po.items()})
        return out, time.perf_counter() - t0

    def engine(dtype):
        store = ParamStore(cfg, device="cuda")
        store.load_state_dict(p, ema)
        eng = Engine(cfg, store, dtype=dtype)
        m, k_, r_ = make_maskinfo(mask.cuda())
        eng.stats.zero_()
        store.grad.zero_()
        dbg = eng.forward_backward(src.cuda(), m, k_, r_, noise.cuda(), keep_debug=True)
        torch.cuda.synchronize()
        D = cfg.embed_dim
        out = {"local_features": dbg["local_features"].float().cpu(), "targets": dbg["targets"].float().cpu(),
               "decoder_out": dbg["decoder_out"].float().cpu(), "loss_sum": eng.stats[4 * D].float().cpu().reshape(1)}
        out.update({"g." + k: store.g[k].detach().float().cpu().clone() for k in p})
        del eng, store
        torch.cuda.empty_cache()
        return out

    ref, t_ref = oracle(False)
    f32 = engine(torch.float32)
    b16 = engine(torch.bfloat16)
    ac = None
    if args.autocast:
        ac, t_ac = oracle(True)
    rows = []
    for k in ref:
        row = dict(tensor=k, numel=ref[k].numel(), ref_norm=float(ref[k].double().norm()),
                   bf16_vs_oracle=rel(b16[k], ref[k]), fp32_vs_oracle=rel(f32[k], ref[k]), bf16_vs_fp32_engine=rel(b16[k], f32[k]))
        if ac is not None:
            row["autocast_oracle_vs_oracle"] = rel(ac[k], ref[k])
        rows.append(row)
    grads = [r for r in rows if r["tensor"].startswith("g.")]
    gs = sorted(grads, key=lambda r: -r["bf16_vs_oracle"])
    # whole-gradient error: all parameter gradients as one vector (what the optimizer's update direction sees)
    gall = lambda d: torch.cat([d[k].double().reshape(-1) for k in ref if k.startswith("g.")])
    summary = dict(model=args.model, samples=L, keep=keep, frames=T, clone_batch=ocfg.clone_batch, oracle_cpu_s=round(t_ref, 1),
                   n_grad_tensors=len(grads),
                   activations={r["tensor"]: {k: r[k] for k in r if k.endswith("oracle") or k.endswith("engine")} for r in rows
                                if not r["tensor"].startswith("g.")},
                   grad_whole_vector=dict(bf16_vs_oracle=float((gall(b16) - gall(ref)).norm() / gall(ref).norm()),
                                          fp32_vs_oracle=float((gall(f32) - gall(ref)).norm() / gall(ref).norm())),
                   grad_bf16_vs_oracle=dict(max=gs[0]["bf16_vs_oracle"], worst=[(r["tensor"][2:], round(r["bf16_vs_oracle"], 5)) for r in gs[:12]],
                                            median=sorted(r["bf16_vs_oracle"] for r in grads)[len(grads) // 2],
                                            n_above_5e_2=sum(r["bf16_vs_oracle"] > 5e-2 for r in grads),
                                            n_above_2e_2=sum(r["bf16_vs_oracle"] > 2e-2 for r in grads)),
                   grad_fp32_vs_oracle_max=max(r["fp32_vs_oracle"] for r in grads))
    if ac is not None:
        summary["autocast_oracle"] = dict(cpu_s=round(t_ac, 1), grad_max=max(r["autocast_oracle_vs_oracle"] for r in grads),
                                          grad_median=sorted(r["autocast_oracle_vs_oracle"] for r in grads)[len(grads) // 2],
                                          grad_whole_vector=float((gall(ac) - gall(ref)).norm() / gall(ref).norm()),
                                          targets=rel(ac["targets"], ref["targets"]), decoder_out=rel(ac["decoder_out"], ref["decoder_out"]))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(dict(summary=summary, rows=rows), f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
