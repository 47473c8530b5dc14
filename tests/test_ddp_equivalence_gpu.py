"""GPU, two ranks sharing the one device, backend gloo (RCCL refuses two ranks on one GPU; gloo stages CUDA tensors
through the host): the data-parallel step is equivalent to gradient accumulation -- two ranks with one micro-batch
each end on the same parameters, teacher and statistics as one rank with both micro-batches.  This runs the real
PretrainTrainer / GradSync code path with world size 2 on device tensors: bucketed asynchronous all-reduce of the flat
gradient issued from the engine's grad-ready events, the packed statistics all-reduce, 1/sample_size and clip scaling
from the global counts, Adam, EMA.  (Dropout and mixup are off: their random streams are per rank by design.)  The finetune variant does the same with
FinetuneTrainer: classifier-only update, then encoder updates with layer_decay, clip-norm over encoder + head."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["A2V_ROOT"])
from animal2vec_amd.config import tiny_config
from animal2vec_amd.trainer import PretrainTrainer, OptimConfig
from animal2vec_amd import masking

def batches():
    cfg = tiny_config()
    L = 3200
    T = cfg.frame_counts(L)[-1]
    g = torch.Generator().manual_seed(5)
    out = []
    for j in range(2):
        src = torch.randn(2, L, generator=g).cuda()
        K = max(2, T // 4)
        m, keep, restore = masking.make_maskinfo(masking.seeded_keep_mask(2 * cfg.clone_batch, T, K, 11 + j, device="cuda"))
        out.append(dict(source=src, noise=torch.randn(2 * cfg.clone_batch, T - K, cfg.embed_dim, generator=g).cuda(),
                        mask_u8=m, ids_keep=keep, ids_restore=restore))
    return cfg, out

def run(mbs_of_step):
    cfg, mbs = batches()
    tr = PretrainTrainer(cfg, OptimConfig(lr=1e-3, warmup_updates=1, max_update=10), dtype=torch.float32, seed=3,
                         dropout=False, mixup=False)
    for _ in range(3):
        tr.train_step(mbs_of_step(mbs))
    torch.cuda.synchronize()
    return tr

def run_finetune(mbs_of_step):
    from animal2vec_amd.finetune import FinetuneConfig, FinetuneTrainer
    cfg, mbs = batches()
    T = cfg.frame_counts(3200)[-1]
    g = torch.Generator().manual_seed(9)
    fmbs = [dict(source=mb["source"], target=(torch.rand(2, T, 5, generator=g) < 0.2).float().cuda()) for mb in mbs]
    f = FinetuneConfig(n_labels=5, apply_mask=False, layerdrop=0.0, dropout=0.0, activation_dropout=0.0, attention_dropout=0.0,
                       source_mixup=-1.0, freeze_finetune_updates=1, layer_decay=0.75, optim_overrides=True, warmup_updates=1, lr=1e-3, min_lr=1e-4,
                       max_update=10, clip_norm=1.0)
    tr = FinetuneTrainer(cfg, f, dtype=torch.float32, seed=3)
    for _ in range(3):                      # one classifier-only update, then two with the encoder
        tr.train_step(mbs_of_step(fmbs))
    torch.cuda.synchronize()
    return tr

torch.cuda.set_device(0)
world = int(os.environ["WORLD_SIZE"])
rank = int(os.environ["RANK"])
if os.environ.get("A2V_MODE") == "finetune":
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = run_finetune((lambda mbs: [mbs[rank]]) if world > 1 else (lambda mbs: mbs))
    if rank == 0:
        torch.save(dict(flat=tr.store.flat.cpu(), teacher=torch.cat([tr.head.w.flatten(), tr.head.b]).cpu(), met=tr.metrics()),
                   os.environ["A2V_OUT"])
    if world > 1:
        assert tr.sync.active
        dist.barrier()
        dist.destroy_process_group()
elif world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = run(lambda mbs: [mbs[rank]])
    assert tr.sync.active and tr.world == 2
    if rank == 0:
        torch.save(dict(flat=tr.store.flat.cpu(), teacher=tr.store.teacher.cpu(), met=tr.metrics()), os.environ["A2V_OUT"])
    dist.barrier()
    dist.destroy_process_group()
else:
    tr = run(lambda mbs: mbs)
    torch.save(dict(flat=tr.store.flat.cpu(), teacher=tr.store.teacher.cpu(), met=tr.metrics()), os.environ["A2V_OUT"])
print("DONE", rank)
"""


@pytest.mark.parametrize("mode", ["pretrain", "finetune"])
def test_two_ranks_equal_gradient_accumulation(tmp_path, mode):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = tmp_path / "ddp_equiv.py"
    f.write_text(_SCRIPT)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), A2V_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0",
                A2V_MODE=mode)
    single = tmp_path / "single.pt"
    r = subprocess.run([sys.executable, str(f)], env=dict(base, WORLD_SIZE="1", RANK="0", A2V_OUT=str(single)),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    multi = tmp_path / "multi.pt"
    procs = [subprocess.Popen([sys.executable, str(f)], env=dict(base, WORLD_SIZE="2", RANK=str(k), LOCAL_RANK="0", A2V_OUT=str(multi)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-1000:] + se[-3000:]
    a, b = torch.load(single), torch.load(multi)
    assert float((a["flat"] - b["flat"]).abs().max()) < 5e-6, float((a["flat"] - b["flat"]).abs().max())
    assert float((a["teacher"] - b["teacher"]).abs().max()) < 5e-6
    for k in (("loss", "sample_size", "pred_var", "target_var", "gnorm") if mode == "pretrain" else ("loss", "sample_size", "gnorm")):
        assert abs(a["met"][k] - b["met"][k]) <= 1e-5 * max(1.0, abs(a["met"][k])), (k, a["met"][k], b["met"][k])
    # and the step really moved the parameters (lr 1e-3 x 3 Adam steps)
    assert float(a["flat"].abs().max()) > 0
