"""CPU, world_size 2, gloo: the bucketed gradient synchronisation used by the trainer (GradSync) reduces
every slice of the flat gradient exactly once, in backward order, and the packed statistics in one call."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from animal2vec_amd import config as C
from animal2vec_amd.ddp import GradSync, make_buckets
from animal2vec_amd.params import build_specs, bucket_starts


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = C.tiny_config()
        specs, _, total = build_specs(cfg)
        buckets = make_buckets(bucket_starts(cfg, specs), total)
        g = torch.Generator().manual_seed(100 + rank)
        grad = torch.randn(total, generator=g)
        local = grad.clone()
        sync = GradSync(grad, buckets)
        assert sync.world == world
        # an accumulation micro-batch that must NOT reduce
        sync.arm(False)
        for ev in buckets:
            sync.on_ready(ev)
        assert torch.equal(grad, local)
        # last micro-batch: events arrive in backward order; 'frontend' is deliberately never fired
        sync.arm(True)
        for ev in list(buckets)[:-1]:
            sync.on_ready(ev)
        stats = torch.tensor([1.0 + rank, 10.0], dtype=torch.float64)
        sync.finish(stats)
        others = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        assert torch.allclose(grad, sum(others), atol=1e-6)
        assert stats.tolist() == [sum(1.0 + r for r in range(world)), 10.0 * world]
        with pytest.raises(AssertionError):
            sync.arm(True)
            sync.on_ready("decoder")
            sync.on_ready("decoder")
        # bf16 on the wire: every rank ends on the sum of the bf16-rounded gradients (two ranks: one rounding of an exact sum),
        # whether a bucket went out during backward or in finish(); the statistics stay in their own dtype
        grad_w = local.clone()
        sync_w = GradSync(grad_w, buckets, wire_dtype=torch.bfloat16)
        sync_w.arm(True)
        for ev in list(buckets)[1:]:
            sync_w.on_ready(ev)
        stats_w = torch.tensor([2.0, 3.0], dtype=torch.float64)
        sync_w.finish(stats_w)
        want = (others[0].bfloat16() + others[1].bfloat16()).float()
        assert torch.equal(grad_w, want)
        assert stats_w.tolist() == [2.0 * world, 3.0 * world]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
