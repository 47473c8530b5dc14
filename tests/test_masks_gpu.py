"""Device span-mask sampler (a2v_span_mask / a2v_maskinfo, csrc/a2v_mask.hip) -- SURVEY row a8.

Integer / index work: bit-exact against the oracle's restatement of the same counter stream (oracle/a2v_oracle.py::span_mask)
and against make_maskinfo; the LAW (what fairseq's compute_mask_indices draws for the reference's arguments, base.py:401-413)
is checked statistically, because fairseq's numpy stream itself is third party and unpinned.
"""
import numpy as np
import pytest
import torch

import a2v_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from animal2vec_amd import ops as o
    return o


def _sample(ops, N, T, **kw):
    mask, keep, restore, kdev, _ = ops.span_mask(N, T, **kw)
    torch.cuda.synchronize()
    K = int(kdev)
    return mask.cpu(), keep[:N * K].view(N, K).cpu(), restore.cpu(), K


@pytest.mark.parametrize("N,T,prob,length,drop,seed", [
    (96, 2000, 1.5, 2, 0.0, 12345),                 # the large recipe, 8 clips x clone_batch 12
    (768, 2000, 1.5, 2, 0.0, (7 << 40) + 99),       # one 64-clip micro-batch; 64-bit seed
    (16, 200, 0.65, 5, 0.1, 7),                     # overlapping long spans + mask_dropout
    (5, 37, 0.5, 3, 0.0, 1),                        # ragged: fewer frames than threads
    (3, 11025, 1.5, 2, 0.0, 2),                     # BASELINE config 5's frame count
    (4, 16384, 0.3, 10, 0.0, 3),                    # the largest T the kernel takes (80 KiB of LDS)
    (2, 50, 8.0, 4, 0.0, 5),                        # more spans asked for than starts exist (min_len shrinks)
    (1, 2, 0.5, 1, 0.0, 0),                         # smallest
])
def test_span_mask_bit_exact_vs_oracle(ops, N, T, prob, length, drop, seed):
    mask, keep, restore, K = _sample(ops, N, T, mask_prob=prob, mask_length=length, mask_dropout=drop, seed=seed)
    ref = O.span_mask(N, T, prob, length, seed, mask_dropout=drop)
    assert torch.equal(mask.bool(), torch.from_numpy(ref)), "mask differs from the oracle's restatement of the stream"
    ik, ir = O.make_maskinfo(torch.from_numpy(ref))
    assert K == ik.shape[1]
    assert torch.equal(keep.long(), ik) and torch.equal(restore.long(), ir)
    # the reference's own definition on top (base.py:433-441): argsort of the mask, kept frames first
    assert torch.equal(torch.gather(mask.long(), 1, keep.long()), torch.zeros(N, K, dtype=torch.long))
    assert bool((mask.sum(1) == T - K).all())


@pytest.mark.parametrize("N,T,prob,length,min_masks,seed", [
    (64, 2000, 0.825, 4, 1, 99),                    # finetune time masks of one 64-clip micro-batch (finetune_mixup_100.yaml: 0.825 / 4)
    (64, 1024, 0.5, 64, 0, 100),                    # finetune channel masks over D = 1024 (0.5 / 64, min_masks 0: base.py:471-484)
    (8, 768, 0.5, 64, 0, 101),                      # base model's D
    (4, 1024, 0.01, 64, 0, 102),                    # 0.16 spans per row: rows that draw NO span (min_masks 0) -> nothing is masked anywhere
])
def test_finetune_mask_shapes_bit_exact_vs_oracle(ops, N, T, prob, length, min_masks, seed):
    """the finetune path's apply_mask draws (reference nn/modalities/base.py:370-425) come from the same device
    sampler as the pretraining masks; time masks with min_masks 1, channel masks with min_masks 0."""
    mask = ops.span_mask(N, T, mask_prob=prob, mask_length=length, min_masks=min_masks, seed=seed)[0]
    torch.cuda.synchronize()
    ref = O.span_mask(N, T, prob, length, seed, min_masks=min_masks)
    assert torch.equal(mask.cpu().bool(), torch.from_numpy(ref))
    cnt = mask.sum(1)
    assert bool((cnt == cnt[0]).all())              # require_same_masks


def test_finetune_make_aug_draws_on_the_device_and_repeats_per_seed(ops):
    from animal2vec_amd.finetune import FinetuneConfig, make_aug
    from animal2vec_amd.masking import mask_seed
    f = FinetuneConfig()
    B, T, D = 8, 500, 1024
    seeds = (mask_seed(3, 17, 0, 1), mask_seed(3, 17, 1, 1))
    a1 = make_aug(f, B, T, D, 8, 24, np.random.default_rng(0), 0.01, mask_seeds=seeds)
    a2 = make_aug(f, B, T, D, 8, 24, np.random.default_rng(5), 0.01, mask_seeds=seeds)        # another host RNG, the same mask seeds
    a3 = make_aug(f, B, T, D, 8, 24, np.random.default_rng(0), 0.01, mask_seeds=(mask_seed(3, 18, 0, 1), mask_seed(3, 18, 1, 1)))
    for k, n in (("mask_t", T), ("mask_c", D)):
        assert a1[k].is_cuda and a1[k].dtype == torch.uint8 and tuple(a1[k].shape) == (B, n)
        assert torch.equal(a1[k], a2[k]) and not torch.equal(a1[k], a3[k])
    assert torch.equal(a1["mask_t"].cpu().bool(), torch.from_numpy(O.span_mask(B, T, f.mask_prob, f.mask_length, seeds[0], min_masks=1)))
    assert torch.equal(a1["mask_c"].cpu().bool(),
                       torch.from_numpy(O.span_mask(B, D, f.mask_channel_prob, f.mask_channel_length, seeds[1], min_masks=0)))
    frac = float(a1["mask_t"].float().mean())
    assert 0.4 < frac < 0.7, frac                    # 0.825 / 4 with overlaps: 1 - exp(-0.825) = 0.56 of the frames
    mc = a1["mask_c"].cpu().numpy()
    assert 0.2 < mc.mean() < 0.5                     # 0.5 / 64 with overlapping spans
    runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], mc[0], [0]]))))[::2]
    assert runs.mean() > 2.0                         # spans (thinned by require_same_masks' equalisation), not isolated channels
    assert a1["noise_std"] == 0.01


def test_same_seed_same_masks_other_seed_other_masks(ops):
    a = _sample(ops, 24, 2000, mask_prob=1.5, mask_length=2, seed=42)
    b = _sample(ops, 24, 2000, mask_prob=1.5, mask_length=2, seed=42)
    c = _sample(ops, 24, 2000, mask_prob=1.5, mask_length=2, seed=43)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert not torch.equal(a[0], c[0])


def test_maskinfo_kernel_vs_argsort(ops):
    from animal2vec_amd.masking import make_maskinfo, seeded_keep_mask
    for N, T, K in ((24, 2000, 140), (3, 11025, 772), (7, 33, 5), (2, 8, 8), (2, 8, 0)):
        m = seeded_keep_mask(N, T, K, 3 + N)
        mu, keep, restore = make_maskinfo(m.cuda())
        ik, ir = O.make_maskinfo(m)
        assert torch.equal(keep.cpu().long(), ik) and torch.equal(restore.cpu().long(), ir)
    bad = seeded_keep_mask(4, 64, 10, 1)
    bad[2, :] = 1                                     # a row that keeps nothing: the reference's equal-count assumption
    with pytest.raises(ValueError):
        make_maskinfo(bad.cuda())


def test_span_count_law(ops):
    """num = int(mask_prob * T / mask_length + U): with unit spans nothing merges, and with one row nothing is equalised, so the
    masked count IS num -- floor or floor + 1, the latter with probability frac."""
    T, prob = 1000, 0.3003                            # 300.3 spans
    counts = np.array([int(_sample(ops, 1, T, mask_prob=prob, mask_length=1, seed=s)[0].sum()) for s in range(400)])
    assert set(counts.tolist()) <= {300, 301}
    frac = float((counts == 301).mean())
    assert abs(frac - 0.3) < 4 * (0.3 * 0.7 / 400) ** 0.5, frac


def test_distribution_matches_the_host_statement_of_the_law(ops):
    """Per-frame masking frequency and the kept count against animal2vec_amd.masking.compute_mask_indices (the numpy statement
    of fairseq's law the round-1/2 trainer used) on the same shapes; both are estimates from 2048 rows."""
    from animal2vec_amd.masking import compute_mask_indices
    N, T, prob, length = 2048, 200, 0.65, 5
    dev_mask, _, _, K = _sample(ops, N, T, mask_prob=prob, mask_length=length, seed=11)
    host = compute_mask_indices((N, T), prob, length, np.random.default_rng(5))
    fd, fh = dev_mask.float().mean(0).numpy(), host.mean(0)
    sigma = (np.maximum(fh * (1 - fh), 0.02) * 2 / N) ** 0.5
    assert np.all(np.abs(fd - fh) < 5 * sigma), float(np.max(np.abs(fd - fh) / sigma))
    assert abs(K - (T - int(host[0].sum()))) <= 4                # batch minimum of ~2048 rows: a narrow extreme-value law
    # run structure after the cut-down: every frame of K the batch minimum moves (an extreme value of 2048 rows) is one more
    # random hole per row, i.e. ~4 % more runs -- compared loosely here, exactly (before the cut-down) in the next test
    rd, rh = _runs(dev_mask.numpy()), _runs(host)
    assert abs(rd.mean() - rh.mean()) < 0.12 * rh.mean()
    assert abs(len(rd) - len(rh)) < 0.10 * len(rh)


def _runs(m):
    m = np.asarray(m, dtype=np.int8)
    d = np.diff(np.concatenate([np.zeros((m.shape[0], 1), np.int8), m, np.zeros((m.shape[0], 1), np.int8)], 1), axis=1)
    lens = []
    for r in range(m.shape[0]):
        s, e = np.nonzero(d[r] == 1)[0], np.nonzero(d[r] == -1)[0]
        lens.append(e - s)
    return np.concatenate(lens)


def test_spans_before_the_cut_down(ops):
    """One row = nothing to equalise: the mask is a union of `num` spans of mask_length frames, so every masked run is at least
    mask_length long, no start lies beyond T - mask_length, and the masked count is at most num * mask_length."""
    T, prob, length = 300, 0.5, 6
    starts_seen = np.zeros(T, dtype=np.int64)
    for s in range(120):
        m = _sample(ops, 1, T, mask_prob=prob, mask_length=length, seed=1000 + s)[0].numpy()
        r = _runs(m)
        assert r.min() >= length, r
        assert int(m.sum()) <= 25 * length
        d = np.diff(np.concatenate([[0], m[0].astype(np.int8)]))
        starts_seen[np.nonzero(d == 1)[0]] += 1
    assert starts_seen[T - length + 1:].sum() == 0
    # run starts spread over the whole admissible range (uniform starts): both halves are hit about equally
    h = (T - length) // 2
    a, b = starts_seen[:h].sum(), starts_seen[h:T - length + 1].sum()
    assert abs(a - b) < 5 * (a + b) ** 0.5, (a, b)


def test_large_recipe_shape_properties(ops):
    """T = 2000, mask_prob 1.5, length 2 (a2v_large_pretrain_best.yaml:127-130): ~93.75 % masked before the cut-down, every row
    equal afterwards, K in the band SURVEY 8c quotes widened for 768 rows (the batch minimum moves out with the row count)."""
    for N, lo, hi in ((96, 132, 156), (768, 140, 165)):
        mask, keep, restore, K = _sample(ops, N, 2000, mask_prob=1.5, mask_length=2, seed=1000 + N)
        assert bool((mask.sum(1) == 2000 - K).all())
        assert lo <= K <= hi, K
        assert bool((keep[:, 1:] > keep[:, :-1]).all())           # ascending kept frames
        inv = torch.empty_like(restore)
        inv.scatter_(1, restore.long(), torch.arange(2000, dtype=restore.dtype).expand(N, -1))   # restore is a permutation
        assert bool((inv.sort(1).values == torch.arange(2000, dtype=restore.dtype)).all())


def test_sampler_runs_one_step_ahead_and_reuses_slots():
    from animal2vec_amd.masking import DeviceMaskSampler, mask_seed
    smp = DeviceMaskSampler("cuda", mask_prob=1.5, mask_length=2)
    N, T = 24, 500
    got = []
    smp.request(N, T, mask_seed(0, 0, 0, 0))
    for u in range(4):
        mask, keep, restore = smp.take(N, T, mask_seed(0, u, 0, 0))
        smp.request(N, T, mask_seed(0, u + 1, 0, 0))
        got.append((mask.clone(), keep.clone(), restore.clone(), smp.last_k))
        smp.end_step()
    torch.cuda.synchronize()
    for u, (mask, keep, restore, K) in enumerate(got):
        ref = O.span_mask(N, T, 1.5, 2, mask_seed(0, u, 0, 0))
        assert torch.equal(mask.cpu().bool(), torch.from_numpy(ref)) and keep.shape == (N, K)
    assert len({id(s) for pool in smp._free.values() for s in pool} | {id(s) for _, s in smp._pending}) <= 3
    # a request nobody takes (shape changed) is dropped, and an unrequested shape is drawn on the spot
    m2, k2, r2 = smp.take(12, 300, 77)
    assert torch.equal(m2.cpu().bool(), torch.from_numpy(O.span_mask(12, 300, 1.5, 2, 77)))


def test_trainer_samples_its_masks_in_the_step_and_per_rank_streams_survive_a_resume(tmp_path):
    """(1) A step given sources only draws its masks on the device; feeding the SAME masks explicitly (the oracle's restatement
    of the stream) gives the same parameters.  (2) ADVICE round 2: every draw is a function of (seed, num_updates, micro-batch,
    rank), so a checkpoint written by rank 0 resumes rank 1's streams too -- emulated by setting the trainer's rank."""
    from animal2vec_amd import checkpoint as ck
    from animal2vec_amd.config import tiny_config
    from animal2vec_amd.masking import make_maskinfo, mask_seed
    from animal2vec_amd.trainer import OptimConfig, PretrainTrainer
    cfg = tiny_config()
    oc = OptimConfig(lr=1e-3, warmup_updates=2, max_update=20)
    L = 3200
    T = cfg.frame_counts(L)[-1]
    g = torch.Generator().manual_seed(2)
    src = [torch.randn(4, L, generator=g).cuda() for _ in range(2)]

    def run(rank, steps, tr=None, explicit=False):
        tr = tr or PretrainTrainer(cfg, oc, dtype=torch.float32, seed=4)
        tr.rank = rank
        for _ in range(steps):
            mbs = []
            for j, s in enumerate(src):
                mb = dict(source=s, noise=None)
                if explicit:
                    a = cfg.audio
                    ref = O.span_mask(4 * cfg.clone_batch, T, a.mask_prob, a.mask_length, mask_seed(4, tr.num_updates, j, rank),
                                      mask_dropout=a.mask_dropout)
                    mb["mask_u8"], mb["ids_keep"], mb["ids_restore"] = make_maskinfo(torch.from_numpy(ref).cuda())
                mbs.append(mb)
            tr.train_step(mbs)
        torch.cuda.synchronize()
        return tr

    a = run(1, 4)
    e = run(1, 4, explicit=True)
    assert float((a.store.flat - e.store.flat).abs().max()) < 5e-6, "in-step sampling != the same masks fed explicitly"
    b = run(1, 2)
    path = tmp_path / "ck.pt"
    ck.save_checkpoint(path, b.store, num_updates=b.num_updates, optimizer_state=b.optimizer_state())
    state = ck.load_checkpoint(path)

    def resumed(rank):
        c = PretrainTrainer(cfg, oc, dtype=torch.float32, seed=4)
        ck.load_pretrained(c.store, state)
        c.engine.refresh_student_packs()
        c.engine.refresh_teacher_packs()
        c.load_optimizer_state(state["last_optimizer_state"])
        return run(rank, 2, tr=c)

    assert float((a.store.flat - resumed(1).store.flat).abs().max()) < 5e-6
    assert float((a.store.flat - resumed(0).store.flat).abs().max()) > 1e-5, "rank 0 and rank 1 drew the same masks / mixups"
