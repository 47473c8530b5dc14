#!/usr/bin/env python3
"""Would an fp16 teacher pay?  (decide with the oracle, on the CPU.)

The reference's own recipe trains in fp16 (configs/MeerKAT/a2v_large_pretrain_best.yaml:4); the build's headline dtype is bf16.
The teacher is a no-grad forward (data2vec2.py:755-844) whose output -- the instance-normalised mean of the top-k FFN outputs --
is the regression target, so its rounding goes straight into the quantity north_star bounds.  fp16 has 11 significand bits
against bf16's 8 at the same MFMA rate, but a 65504 ceiling.  This probe runs the ORACLE's teacher (oracle/a2v_oracle.py:
teacher_forward) on one full-size clip under

    fp32 | torch.autocast("cpu", bfloat16) | torch.autocast("cpu", float16)

and records, per mode, the relative error of the targets against fp32 and, per teacher layer, the largest |value| any Linear /
matmul / LayerNorm of that layer produced (the fp16 overflow margin).  Two weight states: the reference initialisation with an
EMA perturbation (what the full-size parity tests use) and -- when tests/golden/tiny_trained.npz exists -- the tiny
configuration 200 updates from init (off init, the state the bf16 error is worst in).

    python tools/fp16_teacher_probe.py --out profiles/r06_fp16_teacher.json [--model large] [--samples 80000]

Test infrastructure (imports oracle/); never part of the product path.
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
import torch.nn.functional as F


def rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


class MaxTracker:
    """max |output| of every F.linear / F.layer_norm / torch.matmul call made while active, keyed by a running call index"""

    def __init__(self):
        self.vals = []

    def __enter__(self):
        self._lin, self._ln, self._gelu = F.linear, F.layer_norm, F.gelu
        tr = self

        def lin(x, w, b=None):
            y = tr._lin(x, w, b)
            tr.vals.append(("linear%dx%d" % (w.shape[1], w.shape[0]), float(y.detach().abs().max()), str(y.dtype)))
            return y

        def ln(x, shape, w=None, b=None, eps=1e-5):
            y = tr._ln(x, shape, w, b, eps)
            tr.vals.append(("ln_in", float(x.detach().abs().max()), str(x.dtype)))
            return y

        F.linear, F.layer_norm = lin, ln
        return self

    def __exit__(self, *a):
        F.linear, F.layer_norm = self._lin, self._ln


def run_modes(O, ocfg, ema, feats, label):
    out = {"label": label, "T": int(feats.shape[1]), "D": int(feats.shape[2])}
    ref = None
    for mode in ("fp32", "bf16", "fp16"):
        t0 = time.perf_counter()
        with MaxTracker() as tr:
            if mode == "fp32":
                y, ts = O.teacher_forward(ema, feats, ocfg, return_layers=True)
            else:
                with torch.autocast("cpu", dtype=torch.bfloat16 if mode == "bf16" else torch.float16):
                    y, ts = O.teacher_forward(ema, feats, ocfg, return_layers=True)
        y = y.float()
        dt = time.perf_counter() - t0
        rec = {"seconds": round(dt, 1), "finite": bool(torch.isfinite(y).all()),
               "max_abs_linear_out": max(v for k, v, _ in tr.vals if k.startswith("linear")),
               "max_abs_ln_in": max(v for k, v, _ in tr.vals if k == "ln_in"),
               "max_abs_ffn_out_per_layer": [round(float(t.float().abs().max()), 2) for t in ts]}
        # the largest Linear outputs by shape (qkv D->3D, proj D->D, fc1 D->4D, fc2 4D->D)
        by = {}
        for k, v, _ in tr.vals:
            by[k] = max(by.get(k, 0.0), v)
        rec["max_abs_by_op"] = {k: round(v, 2) for k, v in sorted(by.items())}
        if mode == "fp32":
            ref = y
            ref_ts = [t.float() for t in ts]
        else:
            rec["targets_rel_err"] = rel(y, ref)
            rec["ffn_out_rel_err_per_layer"] = [float("%.3g" % rel(t.float(), r)) for t, r in zip(ts, ref_ts)]
        out[mode] = rec
        print(label, mode, json.dumps({k: v for k, v in rec.items() if not isinstance(v, (list, dict))}), flush=True)
    out["fp16_over_bf16"] = out["fp16"]["targets_rel_err"] / out["bf16"]["targets_rel_err"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large", choices=["large", "base", "tiny"])
    ap.add_argument("--samples", type=int, default=80000)
    ap.add_argument("--out", default="profiles/r06_fp16_teacher.json")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    import a2v_oracle as O

    torch.set_num_threads(args.threads)
    res = {"tool": "tools/fp16_teacher_probe.py", "note": "oracle teacher under CPU autocast; errors are ||y - y_fp32|| / ||y_fp32|| of the targets"}
    dims = {"large": O.LARGE, "base": O.BASE_STAR, "tiny": O.TINY}[args.model]
    ocfg = O.OracleConfig(**dims)
    p = O.init_params(ocfg, seed=0)
    g = torch.Generator().manual_seed(9)
    ema = {k: v + torch.randn(v.shape, generator=g) * 0.002 for k, v in O.make_teacher(p).items()}
    L = args.samples
    src = F.layer_norm(torch.randn(1, L, generator=g), (L,))
    with torch.no_grad():
        feats = O.local_features(p, src, ocfg)
    res["at_init_" + args.model] = run_modes(O, ocfg, ema, feats, "at_init_" + args.model)

    fx = os.path.join(ROOT, "tests", "golden", "tiny_trained.npz")
    if os.path.exists(fx):
        import numpy as np
        import json as _json
        z = np.load(fx)
        tcfg = O.OracleConfig(conv_feature_layers=[tuple(int(x) for x in r) for r in z["layers"]], **_json.loads(bytes(z["dims"]).decode()))
        tp = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
        tema = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("ema.")}
        tsrc = torch.from_numpy(z["source"])
        with torch.no_grad():
            tfeats = O.local_features(tp, tsrc, tcfg)
        res["off_init_tiny_trained"] = run_modes(O, tcfg, tema, tfeats, "off_init_tiny_trained")
    os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
    with open(os.path.join(ROOT, args.out), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
