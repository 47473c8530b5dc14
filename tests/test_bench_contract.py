"""CPU: the committed default bench line (profiles/r01_bench_default.json, written by `python bench.py` on the GPU box)
carries every field of the driver's contract, with consistent values."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_contract():
    with open(os.path.join(ROOT, "profiles", "r01_bench_default.json")) as f:
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
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert j["value"] / c["value"] > 10          # the north-star's >= 10x the host-CPU path on one GPU
