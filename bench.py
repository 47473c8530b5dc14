#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/s of animal2vec pretraining (10 s @ 8 kHz clips) on N MI355X.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one optimizer step on a FIXED global batch (strong scaling): micro-batches of fwd+bwd through
the hand-written HIP path, bucketed RCCL gradient all-reduce overlapped with backward, clip + Adam, EMA
teacher update, operand re-pack.  Rank 0 prints ONE JSON line (contract in the task brief) carrying the
roofline of the dominant kernel and the CPU baseline (the oracle timed on this box's host cores).
Synthetic data: randn waveforms, per-clip normalised.  Span masks are sampled INSIDE the timed step, every
step afresh, by the device sampler (a2v_span_mask; the reference samples them on the host in every forward,
base.py:401-413); K = the kept frames per row is whatever the draw gives (T - the batch's smallest masked count).
`--masks precomputed` is the A/B form with seeded keep-masks of `--keep` frames made once outside the loop.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
SUSTAINED_BF16_TFLOPS = 1925.0   # measured: profiles/r05_mfma_power.txt (16 x 16 x 32 bf16 MFMA stream under the power cap)
PEAK_HBM_GBS = 8000.0


def flops_model(cfg, L):
    import math
    D, cb = cfg.embed_dim, cfg.clone_batch
    a, dec = cfg.audio, cfg.audio.decoder
    nb = a.prenet_depth + cfg.depth
    Ts = cfg.frame_counts(L)
    T = Ts[-1]
    fe, cin = 0.0, 1
    for (dim, k, s), t in zip(cfg.conv_layers, Ts):
        fe += t * dim * cin * k
        cin = dim
    proj = T * cin * D

    def tr(S, rows):     # MACs of nb blocks on `rows` sequences of S tokens
        return rows * nb * (S * 12 * D * D + 2 * S * S * D)

    pos = a.conv_pos_depth * D * (D // a.conv_pos_groups) * cfg.pos_kernel
    dd = dec.decoder_dim
    decm = dd * (D // dec.decoder_groups) * dec.decoder_kernel + (dec.decoder_layers - 1) * dd * (dd // dec.decoder_groups) * dec.decoder_kernel + dd * D
    return dict(T=T, front=fe + proj, teacher=tr(T, 1) + pos * T, pos_student=pos * cb * T, decoder=decm * cb * T,
                student=lambda K: tr(K, cb))


def gflop_per_clip(cfg, L, K):
    m = flops_model(cfg, L)
    fwd = m["front"] + m["teacher"] + m["pos_student"] + m["decoder"] + m["student"](K)
    fb = 3 * (m["front"] + m["pos_student"] + m["decoder"] + m["student"](K)) + m["teacher"]
    return 2 * fwd / 1e9, 2 * fb / 1e9


class GemmRecorder:
    """Brackets every conv_gemm / wgrad launch with HIP events on the launch stream (torch's current
    stream) and accumulates algorithmic FLOPs = 2*M*N*K*G per launch."""

    def __init__(self):
        self.recs = []
        self.bytes = {}

    def wrap(self, ops):
        self._cg, self._wg = ops.conv_gemm, ops.conv_wgrad
        rec = self

        def cg(x, w, out, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = rec._cg(x, w, out, **kw)
            e1.record()
            G = kw.get("G", 1)
            epi = "+".join(n for n, on in (("bias", kw.get("bias") is not None), ("gelu", kw.get("gelu")),
                                           ("res", kw.get("res") is not None and not kw.get("gelu_bwd")),
                                           ("gelu_bwd", kw.get("gelu_bwd")), ("pre", kw.get("pre_out") is not None),
                                           ("dact", kw.get("dact")), ("colsum", kw.get("colsum") is not None),
                                           ("drop", kw.get("drop") is not None and kw["drop"][0] > 0)) if on) or "plain"
            rec.recs.append(("nt", 2.0 * kw["M"] * kw["N"] * kw["K"] * G, e0, e1, (kw["M"], kw["N"], kw["K"], G, epi), None))
            # algorithmic bytes: every operand element once (the im2col view re-uses each input element `taps` times)
            rec.bytes["nt"] = rec.bytes.get("nt", 0.0) + x.element_size() * G * (
                kw["M"] * kw["K"] / kw.get("taps", 1) + kw["N"] * kw["K"] + kw["M"] * kw["N"])
            return r

        def wg(x, dy, gw, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = rec._wg(x, dy, gw, **kw)
            e1.record()
            # (a row-masked input -- the first pos-conv layer -- is multiplied over its non-zero rows only since round 5: its FLOPs are
            #  counted over those rows, the fraction read from the mask when the call is recorded; a row list already passes M = listed rows)
            rm = kw.get("rowmask")
            rec.recs.append(("tn", 2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("G", 1) * rec._kept(rm), e0, e1,
                             (kw["M"], kw["N"], kw["K"], kw.get("G", 1), "wgrad" if rm is None else "wgrad, masked input"), None))
            return r

        def sp(x, w_packed, bias, **kw):
            # the row-sparse-input products of the pos-conv's first forward / last input gradient (a2v_conv_sparse_in): FLOPs over
            # the KEPT rows (what the kernel multiplies), the kept fraction read when the call is recorded
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = rec._sp(x, w_packed, bias, **kw)
            e1.record()
            G, T, n_seq, k_ = kw["groups"], kw["T"], kw["n_seq"], kw["k"]
            rows = kw["rowidx"].numel() if kw.get("rowidx") is not None else n_seq * T * rec._kept(kw.get("rowmask"))
            cg_, ng_ = x.shape[-1] // G, kw["n_out"] // G
            rec.recs.append(("nt", 2.0 * rows * ng_ * (k_ * cg_) * G, e0, e1, (int(rows), ng_, k_ * cg_, G, "row-sparse input"), None))
            rec.bytes["nt"] = rec.bytes.get("nt", 0.0) + x.element_size() * (rows * cg_ * G + ng_ * k_ * cg_ * G + n_seq * T * kw["n_out"])
            return r

        def wgp(xa, dya, gwa, xb, dyb, gwb):
            # the paired weight-gradient launch of the Linear layers (fc2 + fc1, proj + qkv: one launch, the library called directly --
            # until round 6 these, the bulk of the TN work, were not recorded at all and "wgrad_kernel" described the convolutions only)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = rec._wgp(xa, dya, gwa, xb, dyb, gwb)
            e1.record()
            M = xa.shape[0]
            fl = 2.0 * M * (dya.shape[1] * xa.shape[1] + dyb.shape[1] * xb.shape[1])
            rec.recs.append(("tn", fl, e0, e1, (M, dya.shape[1] + dyb.shape[1], xa.shape[1], 1,
                                                "wgrad pair %dx%d + %dx%d" % (dya.shape[1], xa.shape[1], dyb.shape[1], xb.shape[1])), None))
            return r

        self._sp, self._wgp = ops.conv1d_cl_sparse_in, ops.linear_wgrad_pair
        ops.conv_gemm, ops.conv_wgrad, ops.conv1d_cl_sparse_in, ops.linear_wgrad_pair = cg, wg, sp, wgp
        # HBM-bound families: algorithmic bytes = every tensor element the call must read or write, once
        self._hbm_orig = {}
        self.hbm = []

        def timed(name, nbytes_of):
            orig = getattr(ops, name)
            rec._hbm_orig[name] = orig

            def f(*a, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig(*a, **kw)
                e1.record()
                rec.hbm.append((name, float(nbytes_of(a, kw, r)), e0, e1))
                return r
            setattr(ops, name, f)

        nb = lambda t: 0 if t is None else t.numel() * t.element_size()
        timed("layernorm_fwd", lambda a, kw, r: nb(a[0]) * (2 + (kw.get("res") is not None) + (kw.get("sum_out") is not None)
                                                            + (kw.get("post") is not None)))
        timed("layernorm_bwd", lambda a, kw, r: nb(a[0]) * (3 + (2 if kw.get("drop_mode", 0) == 2 and kw.get("drop") else 0)))
        timed("sinc_fused_fwd", lambda a, kw, r: nb(a[0]) + nb(r[0]))
        timed("sinc_fused_bwd", lambda a, kw, r: nb(a[0]) + nb(a[6]))

    def unwrap(self, ops):
        ops.conv_gemm, ops.conv_wgrad, ops.conv1d_cl_sparse_in, ops.linear_wgrad_pair = self._cg, self._wg, self._sp, self._wgp
        for name, orig in self._hbm_orig.items():
            setattr(ops, name, orig)

    def hbm_summary(self):
        torch.cuda.synchronize()
        out = {}
        for name in sorted({h[0] for h in self.hbm}):
            rs = [(b, e0.elapsed_time(e1)) for n_, b, e0, e1 in self.hbm if n_ == name]
            by, ms = sum(b for b, _ in rs), sum(t for _, t in rs)
            out[name] = dict(launches=len(rs), avg_us=round(1e3 * ms / len(rs), 2), achieved_GBps=round(by / ms / 1e6, 1),
                             frac_of_8TBps=round(by / ms / 1e6 / 8000.0, 3))
        return out

    @staticmethod
    def _kept(rowmask):
        """fraction of non-zero input rows of a row-masked product (1 without a mask)"""
        return 1.0 if rowmask is None else 1.0 - float(rowmask.float().mean())

    def table(self):
        torch.cuda.synchronize()
        agg = {}
        for k, f, e0, e1, shape, rm in self.recs:
            a = agg.setdefault((k,) + shape, [0, 0.0, 0.0])
            a[0] += 1; a[1] += f * self._kept(rm); a[2] += e0.elapsed_time(e1)
        rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
        return [dict(kind=k[0], M=k[1], N=k[2], K=k[3], G=k[4], epilogue=k[5], launches=v[0], ms=round(v[2], 3),
                     tflops=round(v[1] / v[2] / 1e9, 1)) for k, v in rows]

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind in ("nt", "tn"):
            rs = [(f * self._kept(rm), e0.elapsed_time(e1)) for k, f, e0, e1, _, rm in self.recs if k == kind]
            if rs:
                fl, ms = sum(f for f, _ in rs), sum(t for _, t in rs)
                out[kind] = dict(launches=len(rs), flops=fl, ms=ms, avg_us=1e3 * ms / len(rs), tflops=fl / ms / 1e9,
                                 alg_bytes=self.bytes.get(kind, 0.0) / len(rs))
        return out


PMC_TRAFFIC = None          # set by select_pmc_files(): the newest committed profiles/rNN_pmc_traffic[_<model>].json
PMC_MFMA = None             # ... / rNN_mfma_util[_<model>].json


def select_pmc_files(model):
    """The committed rocprofv3 --pmc summaries of THIS model's workload: the newest round that has them (profiles/rNN_pmc_traffic
    [_<model>].json / rNN_mfma_util[_<model>].json; the large model's files carry no suffix)."""
    global PMC_TRAFFIC, PMC_MFMA
    suf = "" if model == "large" else "_" + model
    for kind in ("pmc_traffic", "mfma_util"):
        best = None
        for rnd in range(9, 0, -1):
            rel = os.path.join("profiles", "r%02d_%s%s.json" % (rnd, kind, suf))
            if os.path.exists(os.path.join(ROOT, rel)):
                best = rel
                break
        if kind == "pmc_traffic":
            PMC_TRAFFIC = best or os.path.join("profiles", "none")
        else:
            PMC_MFMA = best or os.path.join("profiles", "none")


def _pmc_doc(rel):
    if rel is None:
        return None
    try:
        with open(os.path.join(ROOT, rel)) as f:
            return json.load(f)
    except OSError:
        return None


def pmc_traffic(kernel_substr):
    """Average fabric bytes per launch, and the GB/s they amount to, of the kernels whose name contains `kernel_substr`,
    from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be collected from inside this process;
    tools/pmc_traffic.py makes the file from two separate counter runs of this bench's workload at the default
    micro-batch, corrected as MI355X_MICROARCH.md prescribes)."""
    doc = _pmc_doc(PMC_TRAFFIC)
    if doc is None:
        return None, None, None
    n = b = us = 0
    subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
    for name, v in doc["kernels"].items():
        if any(s_ in name for s_ in subs):
            n += v["launches"]
            b += v["launches"] * v["hbm_bytes_per_launch"]
            us += v["launches"] * v["avg_us_under_pmc"]
    if not n:
        return None, None, PMC_TRAFFIC
    return round(b / n), round(b / us / 1e3, 1), PMC_TRAFFIC


def pmc_mfma(kernel_substr):
    """MFMA-busy share (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), tools/pmc_mfma.py) of matching kernels."""
    doc = _pmc_doc(PMC_MFMA)
    if doc is None:
        return None
    n = acc = 0.0
    for name, v in doc["kernels"].items():
        if kernel_substr in name and v.get("mfma_busy_share") is not None:
            w = v["launches"] * v["avg_us_under_pmc"]
            n += w
            acc += w * v["mfma_busy_share"]
    return round(acc / n, 4) if n else None


def make_microbatch(cfg, B, L, K, seed, device, masks="sampled"):
    from animal2vec_amd.masking import make_maskinfo, seeded_keep_mask
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, L, generator=g)
    src = torch.nn.functional.layer_norm(src, (L,)).to(device)
    if masks == "sampled":
        return dict(source=src, noise=None)          # the step draws its own span masks (trainer._step_masks)
    T = cfg.frame_counts(L)[-1]
    mask = seeded_keep_mask(B * cfg.clone_batch, T, K, seed + 1, device=device)
    m, keep, restore = make_maskinfo(mask)
    return dict(source=src, mask_u8=m, ids_keep=keep, ids_restore=restore, noise=None)


def _cpu_leg(model_name, L, K, threads, budget_s):
    """Child-process body of one cpu_baseline leg: the oracle (CPU restatement pinned to the reference), one clip,
    forward + backward, `threads` torch threads: 1 untimed warm-up, then up to 3 timed runs (stops early once
    `budget_s` of timed work is spent).  Prints one JSON object."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import a2v_oracle as O
    dims = O.LARGE if model_name.startswith("large") else O.BASE_STAR
    cfg = O.OracleConfig(**dims)
    torch.set_num_threads(threads)
    p = O.init_params(cfg, seed=0)
    ema = O.make_teacher(p)
    for v in p.values():
        v.requires_grad_(True)
    T = O.frame_counts(L, cfg.conv_feature_layers)[-1]
    g = torch.Generator().manual_seed(1234)
    src = torch.nn.functional.layer_norm(torch.randn(1, L, generator=g), (L,))
    mask = O.seeded_mask(cfg.clone_batch, T, K, 7)
    noise = torch.randn(cfg.clone_batch, T - K, cfg.embed_dim, generator=g) * 0.01

    def one():
        for v in p.values():
            v.grad = None
        t0 = time.perf_counter()
        res = O.pretrain_forward(p, ema, src, mask, noise, cfg)
        res["losses"]["AUDIO_regression"].sum().backward()
        return time.perf_counter() - t0

    warm = one()
    times, spent = [], 0.0
    while len(times) < 3 and (not times or spent + times[-1] <= budget_s):
        times.append(one())
        spent += times[-1]
    print(json.dumps(dict(threads=threads, warmup_s=round(warm, 3), runs_s=[round(t, 3) for t in times])), flush=True)


def _host_memory_available():
    """bytes this process may still allocate: min(MemAvailable, cgroup limit - cgroup usage)"""
    avail = float("inf")
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = float(ln.split()[1]) * 1024
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            l_ = open(lim).read().strip()
            if l_ != "max":
                avail = min(avail, float(l_) - float(open(cur).read().strip()))
        except (OSError, ValueError):
            pass
    return avail if avail != float("inf") else 0.0


def cpu_baseline(model_name, L, K, sample_rate=8000, thread_legs=(16, 32, 64), procs=4, leg_timeout=90):
    """The oracle timed on this box's host cores (SURVEY 8d / BASELINE.md 4: 1 warm-up + median of 3, thread counts stated).
    Legs, each a child process so that torch's thread pool is sized before anything runs and the leg is bounded:
      * one clip on 16 / 32 / 64 torch threads (this small-op-dominated graph does not scale: 608 s on 256 threads against 52 s on 8 in
        round 1 -- the all-core leg of rounds 1-4 never finished a clip inside its 100 s and is gone);
      * `procs` processes x 16 threads at the same time, one clip each (what a CPU user with a many-core host would run): the
        aggregate rate is the sum of the processes' rates while they all run.
    `value` is the best of these, `cores` the threads it used."""
    import subprocess
    ncpu = os.cpu_count() or 1
    sec = L / sample_rate

    def cmd(threads, budget=30.0):
        # (timed work per leg is bounded: the 16-thread leg gets 3 runs, the wider ones stop earlier if they are slower)
        return [sys.executable, os.path.abspath(__file__), "--cpu-leg", str(threads), "--model", model_name, "--samples", str(L),
                "--keep", str(K), "--cpu-budget", str(budget)]

    def env(threads):
        return dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")

    def parse(stdout, threads):
        rec = json.loads(stdout.strip().splitlines()[-1])
        ts = sorted(rec["runs_s"])
        med = ts[len(ts) // 2]
        return dict(threads=threads, audio_s_per_s=round(sec / med, 4), median_s=med, runs_s=rec["runs_s"], warmup_s=rec["warmup_s"])

    legs, seen = [], set()
    for threads in thread_legs:
        threads = min(threads, ncpu)
        if threads in seen:
            continue
        seen.add(threads)
        try:
            out = subprocess.run(cmd(threads, 30.0 if threads <= 16 else 20.0), capture_output=True, text=True, timeout=leg_timeout,
                                 env=env(threads))
            legs.append(parse(out.stdout, threads))
        except subprocess.TimeoutExpired:
            legs.append(dict(threads=threads, audio_s_per_s=None, note=f"no timed run finished within {leg_timeout} s"))
        except Exception as e:                                   # a failed leg must not lose the GPU measurement
            legs.append(dict(threads=threads, audio_s_per_s=None, note=f"leg failed: {type(e).__name__}: {e}"))
    # ---- k processes x 16 threads, one clip each, at the same time (13 GB resident each at the large model: k is also bounded by
    # the memory this process may use -- MemAvailable and the cgroup's limit -- at 20 GB per process, so that a wider leg asked for
    # with --cpu-procs can never take the bench itself down)
    k = max(1, min(procs, ncpu // 16, int(_host_memory_available() // (20 << 30)) or 1))
    if k > 1:
        try:
            ps = [subprocess.Popen(cmd(16, 25.0), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env(16)) for _ in range(k)]
            outs = []
            t_end = time.time() + leg_timeout + 30
            for p_ in ps:
                try:
                    outs.append(p_.communicate(timeout=max(1.0, t_end - time.time()))[0])
                except subprocess.TimeoutExpired:
                    p_.kill()                                    # (the exact child this process started)
                    p_.communicate()
                    outs.append("")
            recs = [parse(o, 16) for o in outs if o.strip()]
            if len(recs) == k:
                legs.append(dict(threads=16 * k, processes=k, threads_per_process=16,
                                 audio_s_per_s=round(sum(r["audio_s_per_s"] for r in recs), 4),
                                 median_s=[r["median_s"] for r in recs], note="aggregate of concurrent one-clip processes"))
            else:
                legs.append(dict(threads=16 * k, processes=k, audio_s_per_s=None, note=f"{k - len(recs)} of {k} processes did not finish"))
        except Exception as e:
            legs.append(dict(threads=16 * k, processes=k, audio_s_per_s=None, note=f"leg failed: {type(e).__name__}: {e}"))
    ok = [l for l in legs if l.get("audio_s_per_s")]
    if not ok:
        return dict(value=None, unit="audio-s/s", cores=legs[0]["threads"], kind="port", sample="cpu legs failed", legs=legs)
    best = max(ok, key=lambda l: l["audio_s_per_s"])
    med = best["median_s"] if not isinstance(best["median_s"], list) else sorted(best["median_s"])[len(best["median_s"]) // 2]
    return dict(value=best["audio_s_per_s"], unit="audio-s/s", cores=best["threads"], kind="port", host_cores=ncpu,
                sample=f"1 clip x {sec:.1f}s per process, {model_name} model, fwd+bwd (teacher targets + student grads), fp32 oracle on "
                       f"torch CPU; per leg 1 warm-up + median of up to 3 timed runs ({med:.1f}s for the reported leg: "
                       f"{best.get('processes', 1)} process(es) x {best['threads'] // best.get('processes', 1)} threads); legs = one clip on "
                       f"{'/'.join(str(t) for t in sorted(seen))} threads and {k} concurrent processes x 16 threads, on a {ncpu}-core host",
                legs=legs)


def parity_mode_leg(cfg, L, dev, args, micro=16, warmup=2, steps=5, f32_clips=16, f32_steps=2):
    """The parity mode of the SAME optimizer step at the headline's protocol: `--global-batch` clips per step as
    micro-batches of `micro` clips (fp32 storage: 64 clips do not fit), `warmup` untimed + `steps` timed steps, masks sampled in the
    step: fp32 storage with every product as three exact bf16 products over split operands on the bf16 matrix cores (bf16x3,
    include/a2v.h: A2V_BF16X3) -- the mode `-m gpu` holds to 1e-3 against the oracle at BASELINE's full size.  Beside it, on a
    small sample, the older all-v_mfma_f32_16x16x4_f32 form of the same mode."""
    from animal2vec_amd.trainer import PretrainTrainer

    def leg(x3, clips_step, micro_, warm, nsteps):
        tr = PretrainTrainer(cfg, dtype=torch.float32, device=dev, seed=0, x3=x3)
        mbs = [make_microbatch(cfg, micro_, L, 0, 4242 + i, dev, masks="sampled") for i in range(clips_step // micro_)]
        tr.prime_memory(micro_, L)                # untimed, as the headline: pools sized above the kept-frame counts the sampler draws
        for _ in range(warm):
            tr.train_step(mbs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            tr.train_step(mbs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tr.engine.close()
        del tr
        torch.cuda.empty_cache()
        return clips_step * nsteps * L / cfg.audio.sample_rate / dt, 1e3 * dt / nsteps

    try:
        gb = int(args.global_batch)
        micro = min(micro, gb)
        while gb % micro:
            micro -= 1
        v3, ms3 = leg(True, gb, micro, warmup, steps)
        v1, ms1 = leg(False, f32_clips, min(f32_clips, micro), 1, f32_steps)
        return dict(dtype="bf16x3 (fp32 storage; every Linear, convolution and attention product, forward and backward, = 3 exact "
                          "v_mfma_f32_16x16x32_bf16 products over split operands, fp32 accumulation; fp32 LayerNorm / softmax / GELU)",
                    value=round(v3, 2), unit="audio-s/s", ms_per_step=round(ms3, 2), steps=steps, warmup=warmup,
                    config=dict(global_batch=gb, micro_batch=micro, micro_batches_per_step=gb // micro,
                                workload="the headline's optimizer step (same clips per step, sampled masks, dropout, mixup, clip + Adam, EMA)"),
                    tolerance="1e-3 relative L2 against the CPU oracle on targets, decoder output, loss and every parameter gradient "
                              "(tests/test_pretrain_gpu.py::test_large_full_size_vs_oracle[bf16x3])",
                    f32_mfma_mode=dict(value=round(v1, 2), ms_per_step=round(ms1, 2),
                                       note=f"the same mode with every product on v_mfma_f32_16x16x4_f32 (round 2's parity mode); "
                                            f"{f32_steps} steps of {f32_clips} clips after 1 warm-up"),
                    sample=f"{steps} timed optimizer steps of {gb} clips ({gb // micro} micro-batches of {micro}) after {warmup} warm-ups: "
                           f"the headline's protocol")
    except Exception as e:                                   # the leg must not lose the headline measurement
        return dict(value=None, note=f"parity leg failed: {type(e).__name__}: {e}")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as children (torch.distributed.run, one process per
    GPU, rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU -- the parent only waits and returns
    the children's exit code; rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    """The bench's control flow on CPU tensors (no HIP kernel runs, nothing is measured): per step every rank fills
    a tiny model's flat gradient buffer, GradSync all-reduces it bucket by bucket in backward order plus the packed
    statistics; timing, barrier and the MAX over ranks are the real code path's."""
    from animal2vec_amd import config as C
    from animal2vec_amd.ddp import GradSync, make_buckets
    from animal2vec_amd.params import bucket_starts, build_specs
    use_dist = world > 1
    if use_dist:
        dist.init_process_group(args.backend if args.backend == "gloo" else "gloo")
    cfg = C.tiny_config()
    specs, _, total = build_specs(cfg)
    buckets = make_buckets(bucket_starts(cfg, specs), total)
    grad = torch.zeros(total)
    stats = torch.zeros(6, dtype=torch.float64)
    sync = GradSync(grad, buckets)
    assert args.global_batch % (world * args.micro_batch) == 0, "global batch must split into micro-batches per rank"
    n_mb = args.global_batch // (world * args.micro_batch)

    expect = float(n_mb * world * (world + 1) // 2)     # every rank adds (rank + 1) per micro-batch; SUM over ranks
    okbox = [True]

    def step():
        grad.zero_()
        for j in range(n_mb):
            sync.arm(j == n_mb - 1)
            grad.add_(float(rank + 1))
            for ev in buckets:
                sync.on_ready(ev)
        sync.arm(False)
        stats.fill_(1.0)
        sync.finish(stats)
        okbox[0] = okbox[0] and bool(torch.all(grad == expect)) and float(stats[0]) == float(world)

    def sync_all():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    okt = torch.tensor([1.0 if okbox[0] else 0.0])
    if use_dist:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    ok = bool(okt[0] == 1.0)
    if rank == 0:
        dt = float(tmax)
        audio_s = args.global_batch * args.samples / 8000.0
        print(json.dumps({"metric": "audio-seconds/s pretrain (10s@8kHz, large model)", "value": round(audio_s * args.steps / dt, 2),
                          "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic", "dry": True,
                          "observed_world_size": dist.get_world_size() if use_dist else 1, "grad_sum_ok": ok,
                          "config": {"workload": "DRY RUN: host-only stand-in, not a measurement", "global_batch": args.global_batch,
                                     "micro_batch": args.micro_batch, "micro_batches_per_rank": n_mb, "parallelism": f"dp{world}"},
                          "roofline": None, "cpu_baseline": None}), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return 0


def default_micro_batch(global_batch, world, samples):
    """The largest micro-batch that divides the rank's share of the global batch: 64 clips of 10 s @ 8 kHz fit one GPU with
    room to spare (fewer, longer launches: +2 % over two micro-batches of 32); longer clips keep to 8."""
    per_rank = max(1, global_batch // world)
    cap = 64 if samples <= 80000 else 8
    return next(m for m in (64, 32, 16, 8, 4, 2, 1) if m <= cap and per_rank % m == 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large", choices=["large", "base", "large44k"],
                    help="large44k: BASELINE config 5 (296x149 front end @ 44.1 kHz; use --samples 220500 --keep 386)")
    ap.add_argument("--global-batch", type=int, default=128,
                    help="clips per optimizer step over all GPUs (fixed as N grows: strong scaling).  128 = the reference recipe's "
                         "update (4 GPUs x 408 000 samples ~ 5 clips x update_freq 5 ~ 102 clips of 10 s, a2v_large_pretrain_best.yaml:"
                         "34-36,45-47,62) rounded up to a power of two so that 1/2/4/8 ranks divide it")
    ap.add_argument("--micro-batch", type=int, default=0,
                    help="clips per fwd/bwd launch sequence per GPU; 0 = the largest of 64/32/16/... that divides the rank's share "
                         "(default_micro_batch: 64 clips of 10 s @ 8 kHz fit one GPU; longer clips keep to 8)")
    ap.add_argument("--samples", type=int, default=80000)
    ap.add_argument("--keep", type=int, default=140, help="kept frames per row of the seeded masks (--masks precomputed only)")
    ap.add_argument("--masks", default="sampled", choices=["sampled", "precomputed"],
                    help="sampled: span masks drawn on the device inside every timed step (default); precomputed: seeded "
                         "keep-masks made once outside the timed loop (A/B)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "bf16"],
                    help="dtype of the gradient buckets on the links (ddp.GradSync): bf16 halves the all-reduce bytes, as the reference's "
                         "fp16 recipe does, for two conversion passes over the flat gradient; fp32 (default) reduces the buffer in place")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "skip"])
    ap.add_argument("--cpu-procs", type=int, default=4, help="concurrent one-clip oracle processes x 16 threads of the cpu_baseline leg "
                    "(bounded by cores / 16 and by the host memory this process may use; 16 = the whole 256-core host, about a minute longer)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--parity-leg", default="auto", choices=["auto", "skip"],
                    help="auto: also time the fp32 parity mode on a bounded sample (rank 0 at N = 1) -> `parity_mode` in the line")
    ap.add_argument("--debug-set", action="append", default=[], help="key=value for a2v_debug_set (A/B experiments)")
    ap.add_argument("--gemm-table", default=None, help="dump per-shape GEMM timings (json) from the roofline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (GPU runs); gloo only with --dry")
    ap.add_argument("--dry", action="store_true",
                    help="host-only stand-in step (tiny flat gradient buffer through the trainer's GradSync on CPU tensors): "
                         "exercises spawn / rendezvous / barrier / max-over-ranks / the JSON line without a GPU")
    ap.add_argument("--cpu-leg", type=int, default=0, help=argparse.SUPPRESS)      # internal: one cpu_baseline leg
    ap.add_argument("--cpu-budget", type=float, default=30.0, help=argparse.SUPPRESS)   # internal: seconds of timed work of that leg
    args = ap.parse_args()

    if args.cpu_leg:
        _cpu_leg(args.model, args.samples, args.keep, args.cpu_leg, budget_s=args.cpu_budget)
        return 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.micro_batch <= 0:
        args.micro_batch = default_micro_batch(args.global_batch, world, args.samples)
    if args.dry:
        return dry_run(args, world, rank)
    assert args.backend == "nccl", "the GPU path runs over RCCL; gloo is for --dry"
    torch.cuda.set_device(local)
    if args.debug_set:
        from animal2vec_amd import _lib
        for kv in args.debug_set:
            k, v = kv.split("=")
            _lib.load().a2v_debug_set(int(k), int(v))
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("A2V_BENCH_FORCE_DIST") == "1"   # the latter: RCCL path on a one-GPU box
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                         # A2V_BENCH_FORCE_DIST without a launcher: a one-rank group on this host
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as s_:
                    s_.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        dist.init_process_group("nccl", device_id=dev)

    from animal2vec_amd import ops
    from animal2vec_amd.config import base_star_config, large_44k_config, large_config
    from animal2vec_amd.trainer import PretrainTrainer

    cfg = {"large": large_config, "base": base_star_config, "large44k": large_44k_config}.get(args.model, base_star_config)()
    select_pmc_files(args.model)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    assert args.global_batch % (world * args.micro_batch) == 0, "global batch must split into micro-batches per rank"
    n_mb = args.global_batch // (world * args.micro_batch)
    tr = PretrainTrainer(cfg, dtype=dtype, device=dev, seed=0, grad_wire=torch.bfloat16 if args.grad_wire == "bf16" else None)
    L, K = args.samples, args.keep
    mbs = [make_microbatch(cfg, args.micro_batch, L, K, 1234 + 1000 * rank + 10 * j, dev, masks=args.masks) for j in range(n_mb)]

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed, before the warm-up: size the allocator's pools at a kept-frame count above what the sampler draws (K varies from step
    # to step; an unseen larger K costs a multi-second hipMalloc on a fresh process -- trainer.prime_memory)
    k_primed = tr.prime_memory(args.micro_batch, L) if args.masks == "sampled" else None
    if os.environ.get("A2V_BENCH_STEP_TIMES") == "1":      # diagnosis only: wall time of every step, synchronised (not a measurement)
        for i in range(args.warmup + args.steps + 4):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            tr.train_step(mbs)
            torch.cuda.synchronize()
            print(f"[step {i}] {1e3 * (time.perf_counter() - t_):.1f} ms  K={tr.mask_sampler.last_k}  "
                  f"reserved={torch.cuda.memory_reserved() / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        tr.train_step(mbs)
    sync()
    smp = tr.mask_sampler
    k0 = (smp.k_sum, smp.k_count)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_step(mbs)
    sync()
    dt = time.perf_counter() - t0
    if args.masks == "sampled":           # mean kept frames per row over the timed steps (host ints: read back by the sampler)
        k_mean = (smp.k_sum - k0[0]) / max(1, smp.k_count - k0[1])
        K = int(round(k_mean))
    else:
        k_mean = float(K)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    met = tr.metrics()

    roof = None
    if not args.no_roofline and rank == 0:
        rec = GemmRecorder()
        rec.wrap(ops)
        side, tr.engine.tstream = tr.engine.tstream, None     # per-kernel timing: one stream, nothing overlapping the timed launches
        try:
            tr.engine.stats.zero_()
            if args.masks == "sampled":
                from animal2vec_amd.masking import mask_seed
                mk = smp.take(args.micro_batch * cfg.clone_batch, cfg.frame_counts(L)[-1], mask_seed(0, 10 ** 6, 0, rank))
            else:
                mk = (mbs[0]["mask_u8"], mbs[0]["ids_keep"], mbs[0]["ids_restore"])
            tr.engine.forward_backward(mbs[0]["source"], mk[0], mk[1], mk[2], (cfg.audio.mask_noise_std, 4321))
            smp.end_step()
            s = rec.summary()
            hbm = rec.hbm_summary()
            if args.gemm_table:
                with open(args.gemm_table, "w") as f:
                    json.dump(rec.table(), f, indent=0)
        finally:
            rec.unwrap(ops)
            tr.engine.tstream = side
        nt = s["nt"]
        traffic, _, traffic_src = pmc_traffic("gemm_nt_bf16_p8_kernel") if dtype == torch.bfloat16 else (None, None, None)
        # (round 5: the bf16 LayerNorms run in ln_fwd_stream_kernel / ln_bwd_stream_kernel; the row-per-wave kernels keep the rest)
        for name, sub in (("layernorm_fwd", ("layernorm_fwd_kernel", "ln_fwd_stream_kernel")),
                          ("layernorm_bwd", ("layernorm_bwd_kernel", "ln_bwd_stream_kernel")),
                          ("sinc_fused_fwd", "sinc_fused_fwd_kernel"), ("sinc_fused_bwd", "sinc_fused_bwd_kernel")):
            if name in hbm:      # the same kernels' fabric bytes / duration from the PMC passes (not an algorithmic count)
                hbm[name]["pmc_bytes_per_launch"], hbm[name]["pmc_GBps"], _ = pmc_traffic(sub)
        roof = dict(bound="mfma", kernel="a2v_conv_gemm family: gemm_nt_bf16_p8_kernel<*> (Linears) + conv_nt_slab2_kernel / conv_dec_kernel (grouped convs) + gemm_nt_bf16_glds_kernel (front end)" if dtype == torch.bfloat16 else "gemm_nt_f32_kernel",
                    achieved=round(nt["tflops"], 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                    frac=round(nt["tflops"] / PEAK_BF16_TFLOPS, 4), traffic=traffic, traffic_unit="bytes/launch (HBM+MALL, PMC)",
                    traffic_source=traffic_src, algorithmic_bytes_per_launch=round(nt["alg_bytes"]),
                    launches_per_microbatch=nt["launches"], avg_launch_us=round(nt["avg_us"], 2),
                    wgrad_kernel=dict(tflops=round(s["tn"]["tflops"], 2), launches=s["tn"]["launches"],
                                      avg_launch_us=round(s["tn"]["avg_us"], 2)) if "tn" in s else None,
                    # rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES share of the kernels' cycles (the file select_pmc_files() picked, named in `source`)
                    mfma_busy_share={"gemm_nt_bf16_p8_kernel": pmc_mfma("gemm_nt_bf16_p8_kernel"), "conv_nt_slab2_kernel": pmc_mfma("conv_nt_slab"), "conv_dec_kernel": pmc_mfma("conv_dec_kernel"),
                                     "gemm_tn_bf16_p8_kernel": pmc_mfma("gemm_tn_bf16_p8_kernel"), "attn_fwd32_bf16_kernel": pmc_mfma("attn_fwd32_bf16_kernel"),
                                     "source": PMC_MFMA} if dtype == torch.bfloat16 else None,
                    # what the matrix pipe of this pool's boxes sustains under its power cap (tools/native/mfma_power.hip: independent
                    # v_mfma_f32_16x16x32_bf16 on register operands, all CUs, seconds long -- a 1.85 GHz clock at a full pipe); `peak` above
                    # stays the guide's nominal figure
                    sustained_peak=dict(tflops=SUSTAINED_BF16_TFLOPS, frac=round(nt["tflops"] / SUSTAINED_BF16_TFLOPS, 4),
                                        source="profiles/r05_mfma_power.txt") if dtype == torch.bfloat16 else None,
                    # the HBM-bound families of the same micro-batch (algorithmic bytes / HIP-event time, peak 8 TB/s)
                    hbm_kernels=hbm)
    if use_dist:
        dist.barrier()

    if rank == 0:
        audio_s = args.global_batch * L / cfg.audio.sample_rate
        value = audio_s * args.steps / dt
        gf_fwd, gf_fb = gflop_per_clip(cfg, L, k_mean)      # the student's share follows the kept frames actually drawn
        line = {
            "metric": {"large": "audio-seconds/s pretrain (10s@8kHz, large model)",
                       "base": "audio-seconds/s pretrain (10s@8kHz, base* model)",
                       "large44k": "audio-seconds/s pretrain (44.1 kHz, large model, BASELINE config 5)"}[args.model],
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"animal2vec-{args.model} pretrain step (student fwd+bwd, EMA-teacher targets, grad "
                                   f"all-reduce, clip+Adam, EMA) on {L}-sample clips @ {cfg.audio.sample_rate} Hz, "
                                   f"{k_mean:.1f}/{cfg.frame_counts(L)[-1]} frames kept, clone_batch {cfg.clone_batch}",
                       "global_batch": args.global_batch, "micro_batch": args.micro_batch, "micro_batches_per_rank": n_mb,
                       "parallelism": f"dp{world}", "grad_all_reduce": f"{args.grad_wire} buckets, overlapped with backward",
                       "dropout": "on (p=0.1, 5 sites)", "mixup": "on (BC, A-weighted gain)",
                       "masks": ("span masks (mask_prob %g, length %d) + gather ids sampled on the device INSIDE every timed step, fresh per "
                                 "step and micro-batch (a2v_span_mask on a side stream, one step ahead; K read back through pinned "
                                 "memory); the reference samples them on the host in every forward (base.py:401-413)"
                                 % (cfg.audio.mask_prob, cfg.audio.mask_length)) if args.masks == "sampled" else
                                "seeded keep-masks / gather ids precomputed once per micro-batch OUTSIDE the timed loop (A/B form)",
                       "kept_frames_mean": round(k_mean, 2), "allocator_primed_at_kept_frames": k_primed,
                       "global_batch_note": "strong scaling at a fixed global batch; 128 clips ~ the reference recipe's update "
                                            "(4 GPUs x 5 clips x update_freq 5 ~ 102 clips, a2v_large_pretrain_best.yaml:34-36,45-47,62)",
                       "observed_world_size": dist.get_world_size() if use_dist else 1,
                       "collectives": ("on (RCCL process group, forced at world size 1)" if world == 1 else "on (RCCL)") if (use_dist and tr.sync.active)
                       else "off (one rank, no process group)"},
            "model_tflops_per_s": round(gf_fb * args.global_batch * args.steps / dt / 1e3, 2),
            "mfu_vs_bf16_dense_peak": round(gf_fb * args.global_batch * args.steps / dt / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
            # (model FLOPs = the reference's nominal count: every product dense over all T frames of every clone.  Six of the fifteen
            #  positional-encoder products -- the first layer's, whose input is zero at the masked frames, and the last layer's, whose output
            #  is only used at the kept frames -- are evaluated over the kept rows only: 5.6 % of the nominal FLOPs are not executed)
            "model_flops_note": "nominal (the reference's dense count); 5.6 % of it -- the positional encoder's first / last layer on masked "
                                "clones -- is evaluated over the kept rows only",
            "loss_per_sample": round(met["loss"], 5), "pred_var": round(met["pred_var"], 5),
            "target_var": round(met["target_var"], 5),
            "roofline": roof,
        }
        if args.cpu_baseline == "auto" and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.model, L, K, cfg.audio.sample_rate, procs=args.cpu_procs,
                                                leg_timeout=90 if args.cpu_procs <= 4 else 240)
        else:
            line["cpu_baseline"] = None
        # the mode that carries the north-star's 1e-3 (the bf16 headline above is held to bf16-level bounds), timed on a
        # bounded sample of the same workload: never `value`
        line["parity_mode"] = parity_mode_leg(cfg, L, dev, args) if (args.parity_leg == "auto" and world == 1 and args.dtype == "bf16") else None
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
