"""GPU, one rank, backend nccl (= RCCL): the trainer's bucketed asynchronous all-reduces run on RCCL's stream while
backward keeps issuing kernels.  With one rank the reduction is the identity, so two optimizer steps with the
collectives forced on (A2V_DDP_FORCE=1) must leave the parameters (up to atomic-order rounding) of the same two steps without them -- any
missing stream dependency between the engine's kernels and the collective shows up as a difference.  (The
world_size-2 semantics are covered on CPU by test_ddp_gloo.py; multi-GPU boxes are not available to the tests.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["A2V_ROOT"])
sys.path.insert(0, os.path.join(os.environ["A2V_ROOT"], "tests"))
from animal2vec_amd.config import tiny_config
from animal2vec_amd.trainer import PretrainTrainer, OptimConfig
from animal2vec_amd import masking

def run(force, wire=None):
    os.environ["A2V_DDP_FORCE"] = "1" if force else "0"
    cfg = tiny_config()
    tr = PretrainTrainer(cfg, OptimConfig(lr=1e-3, warmup_updates=1, max_update=10), dtype=torch.float32, seed=3, grad_wire=wire)
    assert tr.sync.active == force
    L = 3200
    T = cfg.frame_counts(L)[-1]
    B = 2
    g = torch.Generator().manual_seed(5)
    mbs = []
    for j in range(2):
        src = torch.randn(B, L, generator=g).cuda()
        K = max(2, T // 4)
        m, keep, restore = masking.make_maskinfo(masking.seeded_keep_mask(B * cfg.clone_batch, T, K, 11 + j, device="cuda"))
        mbs.append(dict(source=src, noise=torch.randn(B * cfg.clone_batch, T - K, cfg.embed_dim, generator=g).cuda(),
                        mask_u8=m, ids_keep=keep, ids_restore=restore))
    for _ in range(2):
        tr.train_step(mbs)
    torch.cuda.synchronize()
    return tr.store.flat.clone(), tr.store.teacher.clone(), tr.metrics()

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a, ta, ma = run(True)
b, tb, mb = run(False)
c, tc, mc = run(True, torch.bfloat16)
dist.destroy_process_group()
# fp32 atomics in the weight-gradient kernels make two runs differ in the last bits (~1e-7 after Adam); a bucket
# reduced before its producer finished would move parameters by O(lr) = 1e-3
assert float((a - b).abs().max()) < 5e-6 and float((ta - tb).abs().max()) < 5e-6, float((a - b).abs().max())
assert abs(ma["loss"] - mb["loss"]) < 1e-5 * abs(mb["loss"])
# bf16 buckets on the wire: every gradient is rounded to bf16 (2^-9 relative) before Adam, whose normalised step moves by that
# fraction of lr at most where the second moment is dominated by the current gradient: parameters within a few lr * 2^-8 = 4e-6
# ... and exactly the fp32 run's where a bucket was never converted back (a missed copy leaves the gradient un-reduced: identical)
dc = float((c - b).abs().max())
assert 0.0 < dc < 5e-5, dc
assert abs(mc["loss"] - mb["loss"]) < 1e-3 * abs(mb["loss"])
print("RCCL_PATH_OK", ma["loss"], dc)
"""


def test_trainer_over_rccl_one_rank(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", A2V_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    f = tmp_path / "rccl_one_rank.py"
    f.write_text(_SCRIPT)
    r = subprocess.run([sys.executable, str(f)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_itself_runs_its_rccl_path_on_one_gpu():
    """the RCCL code path of bench.py itself (process-group set-up from the launcher's environment, GradSync's
    bucketed asynchronous all-reduces inside the timed steps, the packed statistics all-reduce, max-over-ranks timing) with one
    rank on the one GPU of the test box: `A2V_BENCH_FORCE_DIST=1 A2V_DDP_FORCE=1 python bench.py --gpus 1 --steps 2`.  The JSON
    line must keep the driver's contract and say that the gradients went through the collectives."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(A2V_BENCH_FORCE_DIST="1", A2V_DDP_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--global-batch", "8",
                        "--cpu-baseline", "skip", "--no-roofline", "--parity-leg", "skip"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["unit"] == "audio-s/s" and j["dtype"] == "bf16"
    assert j["config"]["global_batch"] == 8 and j["config"]["observed_world_size"] == 1
    assert abs(j["value"] - 80.0 / (j["ms_per_step"] / 1e3)) / j["value"] < 1e-3
    assert "overlapped with backward" in j["config"]["grad_all_reduce"]
    assert j["config"].get("collectives") == "on (RCCL process group, forced at world size 1)", j["config"]
    assert 0 < j["loss_per_sample"] < 1e4
