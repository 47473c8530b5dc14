"""CPU: the real-data input pipeline (manifest TSV, PCM16 wav, event labels -> frame targets, collate, prefetch)
against a restatement of the reference's own steps (nn/audio_tasks.py:235-465) on files written by the test."""
import os
import wave

import numpy as np
import pytest
import torch
from scipy import interpolate

from animal2vec_amd import data as D

LAYERS = [(127, 63, 1), (512, 10, 5), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 3, 1), (512, 2, 1), (512, 2, 1)]
LABELS = ["beep", "cc", "oth", "sn", "focal"]


def write_wav(path, x, sr=8000, nch=1):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with wave.open(path, "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())


def reference_targets(wav_len, lab, unique_labels, layers):
    """audio_tasks.py:345-384, literally (dense source vector + interp1d)"""
    size = wav_len
    for _, k, s in layers:
        if s == 1:
            continue
        p = int(np.ceil(s / 2))
        size = int(np.floor((size + 2 * p - (min(10, k) - 1) - 1) / s + 1))
    source_vector = np.zeros((wav_len, len(unique_labels)), dtype=np.int64)
    tidx = np.round(np.linspace(0, wav_len, size, endpoint=False)).astype(np.int64)
    for ii, (s, e, l) in enumerate(zip(lab["start_frame_lbl"], lab["end_frame_lbl"], lab["lbl_cat"])):
        source_vector[int(s):int(e), int(l)] = 1
        if unique_labels[-1].lower() == "focal" and lab["foc"][ii] == 1:
            source_vector[int(s):int(e), -1] = 1
    f = interpolate.interp1d(np.arange(wav_len), source_vector, axis=0, kind="linear")
    return np.round(f(tidx), decimals=0).astype(np.int64)


@pytest.fixture
def corpus(tmp_path):
    rng = np.random.default_rng(0)
    root = tmp_path / "meerkat" / "wav"
    lens = [80000, 80000, 64000, 80000, 3000]
    lines = [str(root)]
    labs = {}
    for i, n in enumerate(lens):
        fn = f"grp{i % 2}/chunk_{i:03d}.wav"
        x = (rng.standard_normal(n) * 0.1).astype(np.float32)
        write_wav(str(root / fn), x)
        lines.append(f"{fn}\t{n}")
        if i == 3:
            continue                                # no label file -> skipped
        k = 6
        s = np.sort(rng.integers(0, n - 2000, k))
        e = s + rng.integers(50, 6000, k)
        lab = dict(start_frame_lbl=s.astype(np.float64), end_frame_lbl=np.minimum(e, n).astype(np.float64),
                   lbl_cat=rng.integers(0, 4, k).astype(np.float64), foc=rng.integers(0, 2, k).astype(np.float64))
        lp = D.filename_audio2label(str(root / fn), lblext="npz")
        os.makedirs(os.path.dirname(lp), exist_ok=True)
        np.savez(lp, **lab)
        labs[fn] = lab
    man = tmp_path / "train_0.tsv"
    man.write_text("\n".join(lines) + "\n")
    return str(man), labs, lens


def test_label_path_and_frame_count():
    assert D.filename_audio2label("/d/MeerKAT/wav/a/b/x_001.wav") == "/d/MeerKAT/lbl/a/b/x_001.h5"
    assert D.filename_audio2label("/d/WAV/x.flac", lblext="npz") == "/d/lbl/x.npz"
    with pytest.raises(RuntimeError):
        D.filename_audio2label("/d/audio/x.wav")
    assert D.target_frames(80000, LAYERS) == 2000 and D.target_frames(64000, LAYERS) == 1600


def test_manifest_and_items(corpus):
    man, labs, lens = corpus
    m = D.read_manifest(man, min_sample_size=4000)
    assert m.skipped == 2 and len(m.fnames) == 3            # one too short, one without labels
    assert m.label_dir.endswith("meerkat") and m.root_dir.endswith("wav")
    assert D.read_manifest(man, min_sample_size=0, require_labels=False).skipped == 0
    ds = D.AudioLabelDataset(man, sample_rate=8000, conv_feature_layers=LAYERS, unique_labels=LABELS, min_sample_size=4000)
    it = ds[0]
    assert it["source"].shape == (80000,) and abs(float(it["source"].mean())) < 1e-4 and abs(float(it["source"].std()) - 1) < 1e-3
    for i in range(len(ds)):
        it = ds[i]
        ref = reference_targets(len(it["source"]), labs[ds.m.fnames[i]], LABELS, LAYERS)
        assert it["target"].shape == ref.shape and np.array_equal(it["target"], ref)
        assert it["target"].sum() > 0
    with pytest.raises(Exception):
        D.AudioLabelDataset(man, sample_rate=16000, conv_feature_layers=LAYERS, unique_labels=LABELS, min_sample_size=4000)[0]


def test_wav_decoding_matches_soundfile_convention(tmp_path):
    x = np.array([0.0, 0.5, -0.5, 0.999, -1.0], dtype=np.float32)
    p = str(tmp_path / "wav" / "t.wav")
    write_wav(p, x)
    y, sr = D.read_wav(p)
    assert sr == 8000 and y.dtype == np.float32
    assert np.array_equal(y, (np.clip(x, -1, 1 - 1 / 32768) * 32768).astype("<i2").astype(np.float32) / 32768.0)
    st = np.stack([x, -x], 1).reshape(-1)
    write_wav(p, st, nch=2)
    y2, _ = D.read_wav(p)
    assert y2.shape == (5, 2)
    assert torch.allclose(D.postprocess(torch.from_numpy(y2), 8000, 8000, False), torch.zeros(5), atol=1e-4)


def test_collate_crop_to_min_and_prefetch(corpus):
    man, labs, lens = corpus
    ds = D.AudioLabelDataset(man, sample_rate=8000, conv_feature_layers=LAYERS, unique_labels=LABELS, min_sample_size=4000)
    col = ds.collater([ds[i] for i in range(3)], np.random.default_rng(1))
    src, tgt = col["net_input"]["source"], col["target"]
    assert src.shape == (3, 64000) and tgt.shape == (3, 1600, 5) and col["ntokens"] == 3 * 1600
    k = [i for i in range(3) if ds.sizes[i] == 64000][0]
    assert torch.equal(src[k], ds[k]["source"]) and np.array_equal(tgt[k].numpy(), ds[k]["target"])
    # a cropped clip keeps its labels aligned with its audio: the crop offset maps to the matching frame
    j = [i for i in range(3) if ds.sizes[i] == 80000][0]
    full = ds[j]
    win = full["source"].unfold(0, 64, 1)[:16001]
    off = int(np.flatnonzero((win == src[j][:64]).all(1).numpy())[0])
    assert torch.equal(full["source"][off:off + 64000], src[j])
    f0 = min(int(round(off * 2000 / 80000)), 400)
    assert np.array_equal(tgt[j].numpy(), full["target"][f0:f0 + 1600])
    # batching by size + sharding
    bs = D.batch_by_size(ds.sizes, max_tokens=160000)
    assert sorted(sum(bs, [])) == [0, 1, 2] and all(min(ds.sizes[i] for i in b) * len(b) <= 160000 for b in bs)
    r0 = D.batch_by_size(ds.sizes, 80000, shard=(0, 2))
    r1 = D.batch_by_size(ds.sizes, 80000, shard=(1, 2))
    assert len(r0) == len(r1) == 1 and set(sum(r0, [])).isdisjoint(sum(r1, []))
    got = list(D.DevicePrefetcher(ds, bs, device=None, depth=2))
    assert len(got) == len(bs) and got[0]["target"].dtype == torch.float32 and got[0]["source"].dim() == 2
    with pytest.raises(IndexError):
        list(D.DevicePrefetcher(ds, [[99]], device=None))


# ------------------------------------------------------------------------------------------ pinned against the reference
def _rebuild_corpus(fx, split, root):
    """Write the fixture's synthetic corpus (tests/golden/data_pipeline.npz, dumped by oracle/gen_golden.py from the
    corpus it served through the LIVE reference's FileAudioLabelDataset) back to disk in the reference's layout."""
    import wave
    os.makedirs(os.path.join(root, "wav"), exist_ok=True)
    os.makedirs(os.path.join(root, "lbl"), exist_ok=True)
    names = sorted({k.split(".")[2] for k in fx if k.startswith(f"{split}.wav.")})
    for name in names:
        x = fx[f"{split}.wav.{name}"].numpy()
        with wave.open(os.path.join(root, "wav", name + ".wav"), "wb") as w:
            w.setnchannels(x.shape[1])
            w.setsampwidth(2)
            w.setframerate(8000)
            w.writeframes(x.astype("<i2").tobytes())
        if f"{split}.lbl.{name}.start" in fx:
            with open(os.path.join(root, "lbl", name + ".h5"), "wb") as f:      # the reference's file name; an .npz container
                np.savez(f, start_frame_lbl=fx[f"{split}.lbl.{name}.start"].numpy(), end_frame_lbl=fx[f"{split}.lbl.{name}.end"].numpy(),
                         lbl_cat=fx[f"{split}.lbl.{name}.cat"].numpy(), foc=fx[f"{split}.lbl.{name}.foc"].numpy())
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.npz"))
    with open(os.path.join(root, "train.tsv"), "w") as f:
        f.write(root + "\n" + "\n".join(str(l) for l in z[f"{split}.manifest"]) + "\n")


def _task(root, crop):
    import nn
    from animal2vec_amd.config import LARGE_LAYERS_STR
    cfg = nn.AudioConfigCCAS(data=root, sample_rate=8000, normalize=True, with_labels=True, min_sample_size=1000,
                             max_sample_size=None, conv_feature_layers=LARGE_LAYERS_STR, unique_labels="['a', 'b', 'c', 'focal']",
                             use_focal_loss=True, do_focal_prediction=True, collate_crop=crop)
    return nn.AudioTaskCCAS.setup_task(cfg)


@pytest.mark.parametrize("split", ["equal", "ragged"])
def test_dataset_and_collater_match_the_reference(tmp_path, split):
    """AudioTaskCCAS.load_dataset -> data.AudioLabelDataset against FileAudioLabelDataset's own outputs: which manifest
    entries survive (no label file / shorter than min_sample_size are dropped), every item's waveform (PCM16 -> float,
    stereo mean, whole-clip layer norm) and frame targets (event list -> multi-hot through the reference's interp1d
    path), and the collated batch -- equal lengths: a copy; different lengths: the reference's per-clip and
    per-label-column random crops reproduced draw for draw (collate_crop="reference", numpy global RNG seeded alike)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.npz"))
    fx = {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}
    root = str(tmp_path / split)
    _rebuild_corpus(fx, split, root)
    task = _task(root, "reference")
    ds = task.load_dataset("train")
    assert task.dataset("train") is ds
    kept = [str(k) for k in fx[f"{split}.kept"]]
    assert ds.m.fnames == kept
    assert np.array_equal(ds.sizes, fx[f"{split}.sizes"].numpy())
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        assert torch.allclose(it["source"], fx[f"{split}.item{i}.source"], rtol=0, atol=1e-6), i
        assert np.array_equal(it["target"], fx[f"{split}.item{i}.target"].numpy()), i
    np.random.seed(123)
    col = ds.collater(items)
    assert torch.equal(col["id"], fx[f"{split}.collate.id"])
    assert torch.allclose(col["net_input"]["source"], fx[f"{split}.collate.source"], rtol=0, atol=1e-6)
    assert torch.equal(col["target"], fx[f"{split}.collate.target"]) and col["target"] is col["net_input"]["target"]
    assert int(col["ntokens"]) == int(fx[f"{split}.collate.ntokens"])


def test_aligned_collate_differs_from_the_reference_only_in_the_label_crop(tmp_path):
    """collate_crop="aligned" (the default): one crop offset per clip and the labels cut at the matching frame -- each row
    of the batch is a contiguous window of its item, and its labels are the item's labels of that window; the
    reference's independent per-column offsets do not have that property on the same corpus (fixture)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.npz"))
    fx = {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}
    root = str(tmp_path / "ragged")
    _rebuild_corpus(fx, "ragged", root)
    ds = _task(root, "aligned").load_dataset("train")
    items = [ds[i] for i in range(len(ds))]
    col = ds.collater(items, np.random.default_rng(5))
    L = col["net_input"]["source"].shape[1]
    assert L == 7200 and col["target"].shape[1] == 180

    def window_start(row, full):
        full = full.numpy() if torch.is_tensor(full) else full
        row = row.numpy()
        hits = [s for s in range(len(full) - len(row) + 1) if np.array_equal(full[s:s + len(row)], row)]
        return hits

    for i, it in enumerate(items):
        s_hits = window_start(col["net_input"]["source"][i], it["source"])
        assert len(s_hits) == 1
        t_hits = window_start(col["target"][i], torch.from_numpy(it["target"]))
        assert t_hits, i                                   # ALL label columns come from one window ...
        f0 = int(round(s_hits[0] * it["target"].shape[0] / len(it["source"])))
        assert min(f0, it["target"].shape[0] - 180) in t_hits          # ... the one under the audio crop
    # the reference's batch on this corpus: at least one row whose label columns are NOT one window of the item
    ref_t = fx["ragged.collate.target"]
    broken = [i for i, it in enumerate(items)
              if it["target"].shape[0] > 180 and not window_start(ref_t[i], torch.from_numpy(it["target"]))]
    assert broken, "expected the reference's per-column crops to misalign at least one ragged row"
