"""Real-data input pipeline: manifest TSV -> PCM16 wav chunks + event labels -> frame-rate
multi-hot targets -> crop-to-min collate -> device batches, overlapped with compute.  Host-side restatement of the
reference's FileAudioLabelDataset (nn/audio_tasks.py:196-465) for the hot path's two consumers (PretrainTrainer:
source only; FinetuneTrainer: source + target); no fairseq dataset / iterator machinery.

File formats (written by the reference's scripts/animal2vec_manifest.py:184-217 and prepare scripts):
  manifest: line 1 = root directory of the audio files; then `relative/path.wav<TAB>num_samples` per line
  audio:    PCM16 wav, mono (multi-channel files are averaged, fairseq RawAudioDataset.postprocess)
  labels:   .../wav/.../name.wav -> .../lbl/.../name.h5 with datasets start_frame_lbl, end_frame_lbl, lbl_cat, foc
            (audio_tasks.py:335-343).  h5py is optional here: the same four arrays in a `.npz` next to it are accepted.
"""
from __future__ import annotations

import os
import queue
import re
import threading
import wave
from dataclasses import dataclass
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

_AUDIO2LABEL = re.compile(r"(?P<pre>.*)(?P<dir>wav)(?P<post>/.*\.)(?P<ext>[a-z0-9]+)$", re.IGNORECASE)


def filename_audio2label(audiofile: str, lbldir: str = "lbl", lblext: str = "h5") -> str:
    """audio_tasks.py:295-314: .../wav/.../file.ext -> .../lbl/.../file.<lblext>"""
    m = _AUDIO2LABEL.match(audiofile)
    if m is None:
        raise RuntimeError(f"Cannot derive label file from: {audiofile}")
    return m.expand(rf"\g<pre>{lbldir}\g<post>{lblext}")


@dataclass
class Manifest:
    root_dir: str
    label_dir: str
    fnames: List[str]
    sizes: np.ndarray
    skipped: int


def _label_path(label_dir: str, fn: str) -> Optional[str]:
    for ext in ("h5", "npz"):
        try:
            p = filename_audio2label(os.path.join(label_dir, fn), lblext=ext)
        except RuntimeError:          # no .../wav/... component on this spelling of the path
            return None
        if os.path.isfile(p):
            return p
    return None


def read_manifest(path: str, *, min_sample_size: Optional[int] = 0, min_label_size: int = 0,
                  require_labels: bool = True) -> Manifest:
    """audio_tasks.py:235-266: entries shorter than min_sample_size, or (when labels are required) whose label file is
    missing / not larger than min_label_size bytes, are skipped."""
    fnames, sizes, skipped = [], [], 0
    with open(path, "r") as f:
        root = f.readline().strip()
        parents, last = os.path.split(root)
        label_dir = parents if last in ("wav", "flac", "audio") else root
        for line in f:
            if not line.strip():
                continue
            items = line.strip().split("\t")
            assert len(items) == 2, line
            sz = int(items[1])
            ok = not (min_sample_size is not None and sz < min_sample_size)
            if ok and require_labels:
                lp = _label_path(root, items[0])
                ok = lp is not None and os.path.getsize(lp) > min_label_size
            if not ok:
                skipped += 1
                continue
            fnames.append(items[0])
            sizes.append(sz)
    return Manifest(root, label_dir, fnames, np.asarray(sizes, dtype=np.int64), skipped)


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """PCM wav -> float32 in [-1, 1) as soundfile.read(dtype="float32") returns it; (frames,) or (frames, channels)."""
    with wave.open(path, "rb") as w:
        nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    return (x.reshape(-1, nch) if nch > 1 else x), sr


def postprocess(feats: torch.Tensor, curr_sample_rate: int, sample_rate: int, normalize: bool) -> torch.Tensor:
    """fairseq RawAudioDataset.postprocess: channel mean, sample-rate check, optional whole-clip layer norm."""
    if feats.dim() == 2:
        feats = feats.mean(-1)
    if curr_sample_rate != sample_rate:
        raise Exception(f"sample rate: {curr_sample_rate}, need {sample_rate}")
    assert feats.dim() == 1, feats.dim()
    if normalize:
        with torch.no_grad():
            feats = torch.nn.functional.layer_norm(feats, feats.shape)
    return feats


def read_labels(path: str) -> Dict[str, np.ndarray]:
    keys = ("start_frame_lbl", "end_frame_lbl", "lbl_cat", "foc")
    if path.endswith(".npz"):
        z = np.load(path)
        return {k: np.asarray(z[k]) for k in keys if k in z}
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic[:2] == b"PK":                        # an .npz container under the .h5 name (test corpora, converted labels)
        z = np.load(path)
        return {k: np.asarray(z[k]) for k in keys if k in z}
    try:
        import h5py
    except ImportError as e:                      # pragma: no cover - h5py is not in this image
        raise ImportError(f"{path}: reading .h5 labels needs h5py; convert to .npz with the same four arrays") from e
    with h5py.File(path, "r") as f:               # pragma: no cover
        return {k: np.asarray(f[k]) for k in keys if k in f}


def target_frames(wav_len: int, conv_feature_layers: Sequence[Tuple[int, int, int]]) -> int:
    """audio_tasks.py:345-349 with nn/utils.py:80-98 (get_conv_size, dim=1: kernel clipped to 10, padding ceil(s/2),
    stride-1 layers leave the size unchanged)."""
    size = wav_len
    for _, k, s in conv_feature_layers:
        if s == 1:
            continue
        p = int(np.ceil(s / 2))
        size = int(np.floor((size + 2 * p - (min(10, k) - 1) - 1) / s + 1))
    return size


def frame_targets(wav_len: int, labels: Dict[str, np.ndarray], unique_labels: Sequence[str],
                  conv_feature_layers: Sequence[Tuple[int, int, int]], *, use_focal_loss: bool = True,
                  do_focal_prediction: bool = True) -> np.ndarray:
    """audio_tasks.py:345-384: event list (sample units) -> per-frame targets, (T, C) multi-hot int64 for the focal
    loss or (T,) class index + 1.  The reference builds a (wav_len, C) vector and runs interp1d at
    round(linspace(0, wav_len, T, endpoint=False)); those abscissae are integers, so the interpolation is a gather --
    done here directly on the event list (same result, no 7.7 MB temporary per clip)."""
    T = target_frames(wav_len, conv_feature_layers)
    idx = np.round(np.linspace(0, wav_len, T, endpoint=False)).astype(np.int64)
    C = len(unique_labels)
    out = np.zeros((T, C) if use_focal_loss else (T,), dtype=np.int64)
    s_, e_, l_ = labels.get("start_frame_lbl", ()), labels.get("end_frame_lbl", ()), labels.get("lbl_cat", ())
    if len(s_) == 0 or len(e_) == 0 or len(l_) == 0:
        return out
    foc = labels.get("foc")
    focal_col = use_focal_loss and do_focal_prediction and unique_labels[-1].lower() == "focal"
    for ii, (s, e, l) in enumerate(zip(s_, e_, l_)):
        s, e, l = int(s), int(e), int(l)
        if s < 0:                                  # numpy slicing semantics of source_vector[s:e]
            s += wav_len
        if e < 0:
            e += wav_len
        lo, hi = np.searchsorted(idx, s, side="left"), np.searchsorted(idx, min(e, wav_len), side="left")
        if use_focal_loss:
            out[lo:hi, l] = 1
            if focal_col and foc is not None and int(foc[ii]) == 1:
                out[lo:hi, -1] = 1
        else:
            out[lo:hi] = l + 1
    return out


class AudioLabelDataset:
    """Index -> {"id", "source" (L,) fp32, "target" (T, C) int64}; audio_tasks.py:316-386."""

    def __init__(self, manifest_path: str, *, sample_rate: int, conv_feature_layers, unique_labels: Optional[Sequence[str]] = None,
                 return_labels: bool = True, normalize: bool = True, max_sample_size: Optional[int] = None,
                 min_sample_size: Optional[int] = 0, min_label_size: int = 0, use_focal_loss: bool = True,
                 do_focal_prediction: bool = True, require_label_files: Optional[bool] = None, crop: str = "aligned"):
        """require_label_files: the reference drops every manifest entry without a label file larger than
        min_label_size bytes even when labels are not returned (audio_tasks.py:253-262); None = only when labels are
        returned (label-free pretraining corpora).  crop: see collater()."""
        self.m = read_manifest(manifest_path, min_sample_size=min_sample_size, min_label_size=min_label_size,
                               require_labels=return_labels if require_label_files is None else require_label_files)
        assert crop in ("aligned", "reference")
        self.crop = crop
        self.sample_rate, self.normalize = sample_rate, normalize
        self.conv_feature_layers = [tuple(x) for x in conv_feature_layers]
        self.unique_labels = list(unique_labels or [])
        self.return_labels = return_labels
        self.max_sample_size = max_sample_size if max_sample_size is not None else int(np.iinfo(np.int64).max)
        self.use_focal_loss, self.do_focal_prediction = use_focal_loss, do_focal_prediction

    def __len__(self):
        return len(self.m.fnames)

    @property
    def sizes(self):
        return self.m.sizes

    def __getitem__(self, index: int) -> dict:
        fn = self.m.fnames[index]
        wav, sr = read_wav(os.path.join(self.m.root_dir, fn))
        feats = postprocess(torch.from_numpy(np.ascontiguousarray(wav)).float(), sr, self.sample_rate, self.normalize)
        d = {"id": index, "source": feats}
        if self.return_labels:
            lp = _label_path(self.m.label_dir, fn) or _label_path(self.m.root_dir, fn)
            wav_len = wav.shape[0]
            d["target"] = frame_targets(wav_len, read_labels(lp), self.unique_labels, self.conv_feature_layers,
                                        use_focal_loss=self.use_focal_loss, do_focal_prediction=self.do_focal_prediction)
        return d

    # ------------------------------------------------------------------ collate (audio_tasks.py:388-465, pad = False)
    def collater(self, samples: List[dict], rng: Optional[np.random.Generator] = None) -> dict:
        """Crop every clip to the shortest one (and to max_sample_size).  Equal-length chunks (the shipped MeerKAT
        layout) are copied as they are.  When clips differ in length the reference crops the sources at a random offset
        and then every label COLUMN of every longer clip at its OWN random offset (collate_samples ->
        crop_to_max_size per column, audio_tasks.py:388-431), which misaligns labels and audio:
          crop="aligned" (default): one offset per clip, labels cropped at the matching frame;
          crop="reference": the reference's draws, call for call, from numpy's global RNG (np.random.seed decides) --
                            bit-equal to FileAudioLabelDataset.collater (tests/golden/data_pipeline.npz)."""
        samples = [s for s in samples if s.get("source") is not None]
        if not samples:
            return {}
        sizes = [len(s["source"]) for s in samples]
        tsz = min(min(sizes), self.max_sample_size)
        src = torch.zeros(len(samples), tsz)
        if self.crop == "reference":
            return self._collate_reference(samples, sizes, tsz, src)
        rng = rng or np.random.default_rng()
        starts = []
        for i, (s, size) in enumerate(zip(samples, sizes)):
            st = int(rng.integers(0, size - tsz + 1)) if size > tsz else 0
            starts.append(st)
            src[i] = s["source"][st:st + tsz]
        out = {"id": torch.tensor([s["id"] for s in samples], dtype=torch.long), "net_input": {"source": src}}
        if self.return_labels:
            tl = [s["target"].shape[0] for s in samples]
            tt = min(min(tl), target_frames(tsz, self.conv_feature_layers))
            tgt = torch.zeros((len(samples), tt) + tuple(samples[0]["target"].shape[1:]), dtype=torch.long)
            for i, s in enumerate(samples):
                f0 = int(round(starts[i] * tl[i] / sizes[i]))
                f0 = min(f0, tl[i] - tt)
                tgt[i] = torch.from_numpy(s["target"][f0:f0 + tt])
            out["net_input"]["target"] = tgt
            out["target"] = tgt
            out["ntokens"] = int(tgt.shape[0] * tgt.shape[1])
        return out

    def _collate_reference(self, samples, sizes, tsz, src):
        def crop(t, target_size):                 # fairseq RawAudioDataset.crop_to_max_size (numpy global RNG)
            diff = t.shape[0] - target_size
            if diff <= 0:
                return t
            start = np.random.randint(0, diff + 1)
            return t[start:t.shape[0] - diff + start]

        for i, (s, size) in enumerate(zip(samples, sizes)):
            src[i] = s["source"] if size == tsz else crop(s["source"], tsz)
        out = {"id": torch.tensor([s["id"] for s in samples], dtype=torch.long), "net_input": {"source": src}}
        if self.return_labels:
            tg = [torch.as_tensor(s["target"]) for s in samples]
            tl = [t.shape[0] for t in tg]
            tt = min(min(tl), self.max_sample_size)
            tgt = torch.zeros((len(samples), tt) + tuple(tg[0].shape[1:]), dtype=torch.long)
            for i, t in enumerate(tg):
                if tl[i] == tt:
                    tgt[i] = t
                elif t.dim() > 1:
                    tgt[i] = torch.stack([crop(col, tt) for col in t.t()], dim=1)
                else:
                    tgt[i] = crop(t, tt)
            out["net_input"]["target"] = tgt
            out["target"] = tgt
            out["ntokens"] = int(tgt.shape[0] * tgt.shape[1])
        return out


def batch_by_size(sizes: np.ndarray, max_tokens: int, *, shuffle_seed: Optional[int] = None, shard: Tuple[int, int] = (0, 1),
                  max_sentences: Optional[int] = None) -> List[List[int]]:
    """Batches of clips whose cropped size (shortest clip x clips) stays within max_tokens samples
    (finetune_mixup_100.yaml:33-35), clips ordered by length as fairseq's ordered_indices does, batches optionally
    shuffled, then sharded round-robin over data-parallel ranks (shard = (rank, world))."""
    order = np.argsort(sizes, kind="stable")[::-1]
    batches, cur = [], []
    for i in order:
        cand = cur + [int(i)]
        if cur and (min(sizes[j] for j in cand) * len(cand) > max_tokens or (max_sentences and len(cand) > max_sentences)):
            batches.append(cur)
            cur = [int(i)]
        else:
            cur = cand
    if cur:
        batches.append(cur)
    if shuffle_seed is not None:
        np.random.default_rng(shuffle_seed).shuffle(batches)
    rank, world = shard
    n = len(batches) // world * world if world > 1 else len(batches)
    return batches[rank:n:world] if world > 1 else batches


class DevicePrefetcher:
    """Reads + collates batches on a worker thread into pinned host memory and copies them to the GPU on a side
    stream, `depth` batches ahead, so that file I/O, label rasterisation and the PCIe copy overlap the previous
    step's kernels (the boundary takes device pointers; see DESIGN.md section 6 for the PCIe-inclusive rate)."""

    def __init__(self, dataset: AudioLabelDataset, batches: Iterable[List[int]], *, device=None, depth: int = 2,
                 float_targets: bool = True, seed: int = 0):
        self.ds, self.batches, self.depth = dataset, list(batches), depth
        self.device = torch.device(device) if device is not None else None
        self.float_targets = float_targets
        self.rng = np.random.default_rng(seed)
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self.stream = torch.cuda.Stream(self.device) if self.device is not None and self.device.type == "cuda" else None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        try:
            if self.stream is not None:
                torch.cuda.set_device(self.device)      # pin_memory() in this thread must not open a context on GPU 0
            for b in self.batches:
                col = self.ds.collater([self.ds[i] for i in b], self.rng)
                src = col["net_input"]["source"]
                tgt = col.get("target")
                if tgt is not None and self.float_targets:
                    tgt = tgt.float()
                item = {"id": col["id"], "ntokens": col.get("ntokens")}
                if self.stream is not None:
                    src = src.pin_memory()
                    tgt = tgt.pin_memory() if tgt is not None else None
                    with torch.cuda.stream(self.stream):
                        item["source"] = src.to(self.device, non_blocking=True)
                        item["target"] = tgt.to(self.device, non_blocking=True) if tgt is not None else None
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    item["_ready"], item["_host"] = ev, (src, tgt)
                else:
                    item["source"], item["target"] = src, tgt
                self.q.put(item)
            self.q.put(None)
        except BaseException as e:          # surface worker errors in the consumer
            self.q.put(e)

    def __iter__(self) -> Iterator[dict]:
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            ev = item.pop("_ready", None)
            item.pop("_host", None)
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                # allocated on the prefetch stream, consumed on `cur`: without record_stream the caching allocator
                # would hand the blocks back to the prefetch stream's pool as soon as the consumer drops them, and
                # the worker's next H2D copy could overwrite them under kernels still queued on `cur`
                for k in ("source", "target"):
                    if item.get(k) is not None:
                        item[k].record_stream(cur)
            yield item
