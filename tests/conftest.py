import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="session")
def tiny_golden():
    return load_golden("tiny_pretrain.npz")


def split_golden(fx):
    p = {k[2:]: v for k, v in fx.items() if k.startswith("p.")}
    ema = {k[4:]: v for k, v in fx.items() if k.startswith("ema.")}
    g = {k[2:]: v for k, v in fx.items() if k.startswith("g.")}
    return p, ema, g


def parity_report(name, dtype, errs, **extra):
    """With A2V_PARITY_REPORT=<path> set, append the worst measured errors of a parity test to that file (jsonl): the
    bf16 bounds in the tests are set from these measurements (profiles/r02_parity_report.jsonl), not guessed."""
    path = os.environ.get("A2V_PARITY_REPORT")
    if not path:
        return
    import json
    es = sorted(((float(e), str(k)) for e, k in errs), reverse=True)
    rec = dict(test=name, dtype=str(dtype), n=len(es), worst=es[:6], median=es[len(es) // 2][0] if es else None,
               **{k: float(v) for k, v in extra.items()})
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")
