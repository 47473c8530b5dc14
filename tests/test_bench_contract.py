"""CPU: the committed default bench line (the newest profiles/rNN_bench_default.json, written by `python bench.py` on the GPU box)
carries every field of the driver's contract, with consistent values; bench.py spawns its own ranks."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench_line():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_default.json")))[-1]


def test_default_bench_line_contract():
    with open(_newest_bench_line()) as f:
        j = json.load(f)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"].split(" at ")[0] in base["metric"] or base["metric"].startswith(j["metric"].split("(")[0])
    assert j["unit"] == "audio-s/s" and j["higher_is_better"] is True and j["n_gpus"] == 1 and j["data"] == "synthetic"
    assert j["vs_baseline"] is None and not base.get("published")          # no published number for this metric
    assert "workload" in j["config"] and "model" not in j["config"]
    # value = clips * seconds per clip * steps / time
    audio_s = j["config"]["global_batch"] * 10.0
    assert abs(j["value"] - audio_s / (j["ms_per_step"] / 1e3)) / j["value"] < 1e-3
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["mfma_busy_share"]["gemm_nt_bf16_p8_kernel"] > 0.3 and "masks" in j["config"]
    for fam in ("layernorm_fwd", "layernorm_bwd", "sinc_fused_fwd", "sinc_fused_bwd"):
        assert r["hbm_kernels"][fam]["achieved_GBps"] > 0 and "pmc_GBps" in r["hbm_kernels"][fam]
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and len(c["legs"]) >= 1
    assert "median" in c["sample"] and "warm-up" in c["sample"]
    assert j["value"] / c["value"] > 10          # the north-star's >= 10x the host-CPU path on one GPU
    if "r03" <= os.path.basename(_newest_bench_line())[:3]:
        # round 3: masks sampled inside the timed step, the 1e-3 mode's rate beside the headline
        assert "INSIDE every timed step" in j["config"]["masks"] and 120 < j["config"]["kept_frames_mean"] < 170
        pm = j["parity_mode"]
        assert pm["value"] > 0 and pm["unit"] == "audio-s/s" and "1e-3" in pm["tolerance"] and pm["value"] > pm["f32_mfma_mode"]["value"]
    if "r05" <= os.path.basename(_newest_bench_line())[:3]:
        # round 5: the 1e-3 mode measured at the headline's protocol; honest CPU legs (16 / 32 / 64 threads + concurrent processes)
        pm = j["parity_mode"]
        assert pm["config"]["global_batch"] == j["config"]["global_batch"] and pm["steps"] >= 5 and pm["warmup"] >= 2
        assert abs(pm["value"] - pm["config"]["global_batch"] * 10.0 / (pm["ms_per_step"] / 1e3)) / pm["value"] < 1e-2
        th = [l["threads"] for l in c["legs"]]
        assert th[:3] == [16, 32, 64] and any(l.get("processes", 1) > 1 for l in c["legs"])


def _run_bench(*argv, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env,
                          timeout=300)


def test_bench_spawns_its_own_ranks_gloo_dry():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver used at N = 1): the parent spawns two
    ranks before touching any GPU, rank 0 prints ONE JSON line with n_gpus = 2 and the observed world size."""
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["observed_world_size"] == 2 and j["dry"] is True and j["grad_sum_ok"] is True
    assert j["steps"] == 3 and j["warmup"] == 1 and j["config"]["micro_batch"] == 64 and j["config"]["micro_batches_per_rank"] == 1 and j["config"]["global_batch"] == 128
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in j, k


def test_bench_spawn_propagates_rank_failure():
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--global-batch", "7", "--micro-batch", "2")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_world_size_mismatch_is_an_error():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_cpu_baseline_median_and_best_leg(monkeypatch):
    """cpu_baseline(): one-clip legs on 16 / 32 / 64 threads (each 1 warm-up + up to 3 timed runs in a child, median), one leg of
    concurrent 16-thread processes (aggregate = sum of the processes' rates); value = the best leg, cores = its threads."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench

    class R:
        def __init__(self, out):
            self.stdout, self.returncode = out, 0

    per_thread_median = {16: 9.0, 32: 8.0, 64: 12.0}

    def rec(th, med):
        return json.dumps(dict(threads=th, warmup_s=12.0, runs_s=[med + 1.0, med - 1.0, med])) + "\n"

    def fake_run(cmd, **kw):
        th = int(cmd[cmd.index("--cpu-leg") + 1])
        if th == 64:
            raise subprocess.TimeoutExpired(cmd, kw.get("timeout"))
        return R(rec(th, per_thread_median[th]))

    class FakePopen:
        def __init__(self, cmd, **kw):
            self.th = int(cmd[cmd.index("--cpu-leg") + 1])

        def communicate(self, timeout=None):
            return rec(self.th, 10.0), ""            # four at once: each a little slower than alone

        def kill(self):
            pass

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    monkeypatch.setattr(os, "cpu_count", lambda: 256)
    monkeypatch.setattr(bench, "_host_memory_available", lambda: float(1 << 40))
    c = bench.cpu_baseline("large", 80000, 152)
    assert [l["threads"] for l in c["legs"]] == [16, 32, 64, 64] and c["legs"][2]["audio_s_per_s"] is None
    assert abs(c["legs"][0]["audio_s_per_s"] - 10.0 / 9.0) < 1e-3 and abs(c["legs"][1]["audio_s_per_s"] - 1.25) < 1e-3
    multi = c["legs"][3]
    assert multi["processes"] == 4 and abs(multi["audio_s_per_s"] - 4.0) < 1e-3
    assert c["value"] == multi["audio_s_per_s"] and c["cores"] == 64 and c["host_cores"] == 256 and c["kind"] == "port"
    assert "warm-up" in c["sample"] and "median" in c["sample"]
    # the concurrent leg is bounded by the memory the process may use (20 GB per oracle process): --cpu-procs 16 on a 256-core host
    # with 50 GB to spare runs two
    monkeypatch.setattr(bench, "_host_memory_available", lambda: 50.0 * (1 << 30))
    c2 = bench.cpu_baseline("large", 80000, 152, procs=16)
    assert c2["legs"][3]["processes"] == 2 and c2["legs"][3]["threads"] == 32
    monkeypatch.setattr(bench, "_host_memory_available", lambda: float(1 << 40))
    # a small host: thread legs collapse to what exists, no multi-process leg
    monkeypatch.setattr(os, "cpu_count", lambda: 8)
    per_thread_median[8] = 20.0
    c = bench.cpu_baseline("large", 80000, 152)
    assert [l["threads"] for l in c["legs"]] == [8] and c["cores"] == 8 and abs(c["value"] - 0.5) < 1e-3


def test_default_micro_batch_per_world_size():
    """one micro-batch per rank for the driver's 1 / 2 / 4 / 8-GPU runs of the 64-clip batch; long clips keep to 8"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert [b.default_micro_batch(64, w, 80000) for w in (1, 2, 4, 8)] == [64, 32, 16, 8]
    assert b.default_micro_batch(64, 1, 441000) == 8 and b.default_micro_batch(4, 1, 441000) == 4
    assert b.default_micro_batch(48, 1, 80000) == 16 and b.default_micro_batch(7, 2, 80000) == 1
