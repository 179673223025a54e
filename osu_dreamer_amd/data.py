"""Feeder for the denoiser: cached latent encodings -> `LatentBatch`es.

Mirrors osu_dreamer/data/modules/latent.py:20-149 (on-disk format, window cropping, shuffle
buffer, mapset hold-out from data/modules/beatmap.py:33-71) so `encode-latents` output is read
unchanged.  MI355X-side differences: the stream is sharded by (rank, worker) — the reference
shards by worker only, so DDP ranks would see identical data — and there is no Lightning
dependency.  `write_synthetic_dataset` produces the same layout from seeded noise for
BASELINE configs[0] (no dataset ships with the repo).
"""
from __future__ import annotations

import random
from pathlib import Path
from typing import Iterator, List, NamedTuple, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, IterableDataset

NUM_LABELS = 5


class LatentBatch(NamedTuple):
    h: torch.Tensor        # (A, l) audio features at latent rate
    z: torch.Tensor        # (E, l) chart latent
    s: torch.Tensor        # (S,)   per-map style code
    labels: torch.Tensor   # (NUM_LABELS,)


def load_latents(latent_file: Path) -> LatentBatch:
    """`<map>.latent.npz` {z, s, labels} + sibling `h.npy` (latent.py:74-80)."""
    with np.load(latent_file) as d:
        z, s, labels = (torch.from_numpy(d[k]).float() for k in ("z", "s", "labels"))
    h = torch.from_numpy(np.load(latent_file.parent / "h.npy")).float()
    return LatentBatch(h, z, s, labels)


def split_mapsets(mapsets: List[Path], counts: List[int], max_val_count: int, max_val_frac: float) -> Tuple[List[Path], List[Path]]:
    """The hold-out rule of data/modules/beatmap.py:38-71 on an ORDERED list of mapsets with their map counts: walk the
    list, a mapset goes to validation while the validation set stays within min(max_val_count, int(total * max_val_frac))
    maps, otherwise to training."""
    full = sum(counts)
    if full == 0:
        raise ValueError("data dir is empty, generate dataset first")
    if max_val_count <= 0:
        raise ValueError(f"invalid {max_val_count=}")
    if not (0 < max_val_frac < 1):
        raise ValueError(f"invalid {max_val_frac=}")
    cap = min(max_val_count, int(full * max_val_frac))
    if not (0 < cap < full):
        raise ValueError(f"invalid max_val_size={cap} given full_size={full} {max_val_count=} {max_val_frac=}")
    train, val, nval = [], [], 0
    for mapset, n in zip(mapsets, counts):
        if nval + n > cap:
            train.append(mapset)
        else:
            val.append(mapset)
            nval += n
    return train, val


def hold_out_mapsets(data_dir: Path, pattern: str, max_val_count: int, max_val_frac: float) -> Tuple[List[Path], List[Path]]:
    """Whole mapsets (directories) are held out so train/val never share audio.  The reference walks
    `data_dir.iterdir()` in file-system order, so its split differs from machine to machine; here the listing is sorted,
    which every rank of a data-parallel run needs anyway (all ranks must hold out the SAME mapsets)."""
    if not data_dir.exists():
        raise ValueError(f"data dir `{data_dir}` does not exist, generate dataset first")
    mapsets = sorted(data_dir.iterdir())
    counts = [sum(1 for _ in m.glob(pattern)) for m in mapsets] if mapsets else []
    extra = sum(1 for _ in data_dir.rglob(pattern)) - sum(counts)      # maps nested deeper count towards the total (rglob)
    if sum(counts) + extra == 0:
        raise ValueError(f"data dir `{data_dir}` is empty, generate dataset first")
    if extra:
        mapsets, counts = mapsets + [None], counts + [extra]
    train, val = split_mapsets(mapsets, counts, max_val_count, max_val_frac)
    return [m for m in train if m is not None], [m for m in val if m is not None]


class LatentDataset(IterableDataset):
    def __init__(self, mapsets: List[Path], seq_len: Optional[int] = None, shuffle_buffer_size: int = 1,
                 max_per_map: int = -1, rank: int = 0, world_size: int = 1):
        super().__init__()
        self.mapsets, self.seq_len = mapsets, seq_len
        self.shuffle_buffer_size = shuffle_buffer_size
        self.max_per_map = max_per_map if max_per_map > 0 else float("inf")
        self.rank, self.world_size = rank, world_size

    def _files(self) -> Iterator[Path]:
        return (f for m in self.mapsets for f in sorted(m.glob("*.latent.npz")))

    def _stream(self, nshards: int, shard: int) -> Iterator[LatentBatch]:
        for i, f in enumerate(self._files()):
            if i % nshards == shard:
                yield from self.make_samples(f)

    def __iter__(self):
        info = torch.utils.data.get_worker_info()
        nw, wid, seed = (1, 0, torch.initial_seed()) if info is None else (info.num_workers, info.id, info.seed)
        random.seed(seed + 7919 * self.rank)
        stream = self._stream(nw * self.world_size, self.rank * nw + wid)
        if self.shuffle_buffer_size <= 1:
            yield from stream
            return
        buf: List[LatentBatch] = []
        for sample in stream:
            if len(buf) < self.shuffle_buffer_size:
                buf.append(sample)
                continue
            j = random.randrange(len(buf))
            yield buf[j]
            buf[j] = sample
        random.shuffle(buf)
        yield from buf

    def make_samples(self, latent_file: Path) -> Iterator[LatentBatch]:
        h, z, s, labels = load_latents(latent_file)
        if self.seq_len is None:
            yield LatentBatch(h, z, s, labels)
            return
        end = z.size(-1) - self.seq_len + 1
        if end < 1:
            return
        start = int(torch.randint(0, min(self.seq_len, end), ()).item())
        idxs = torch.arange(start, end, self.seq_len)
        idxs = idxs[torch.randperm(len(idxs))[: int(min(self.max_per_map, len(idxs)))]]
        for i in idxs:
            yield LatentBatch(h[..., i:i + self.seq_len].clone(), z[..., i:i + self.seq_len].clone(), s, labels)


class LatentDataModule:
    """Same constructor keys as the reference's LatentDataModule (they are the YAML `data:` block)."""

    def __init__(self, batch_size: int, seq_len: int, num_workers: int, max_val_count: int = 512,
                 max_val_frac: float = .3, data_path: str = "./data", shuffle_buffer_size: int = 1,
                 max_per_map: int = -1, rank: int = 0, world_size: int = 1):
        self.batch_size, self.seq_len, self.num_workers = batch_size, seq_len, num_workers
        train, val = hold_out_mapsets(Path(data_path), "*.latent.npz", max_val_count, max_val_frac)
        self.train_set = LatentDataset(train, seq_len, shuffle_buffer_size, max_per_map, rank, world_size)
        self.val_set = LatentDataset(val)

    def train_dataloader(self):
        return DataLoader(self.train_set, batch_size=self.batch_size, num_workers=self.num_workers, pin_memory=True,
                          persistent_workers=self.num_workers > 0, drop_last=True)

    def val_dataloader(self):
        return DataLoader(self.val_set, batch_size=1, num_workers=min(1, self.num_workers), pin_memory=True,
                          persistent_workers=self.num_workers > 0)


def write_synthetic_dataset(data_path: str, n_maps: int = 8, frames=4096, a_dim: int = 128, emb_dim: int = 6,
                            style_dim: int = 32, seed: int = 0):
    """`n_maps` mapset dirs each with h.npy (A, frames) and 0.latent.npz {z (E, frames), s (S,), labels (5,)}:
    h ~ N(0,1), z per-frame RMS-normalised, s RMS-normalised.  `frames` may be a list (one length per map)."""
    rng = np.random.default_rng(seed)
    root = Path(data_path)
    lengths = list(frames) if isinstance(frames, (list, tuple)) else [frames] * n_maps
    for i in range(n_maps):
        d = root / f"{i:04d}"
        d.mkdir(parents=True, exist_ok=True)
        frames = lengths[i]
        z = rng.standard_normal((emb_dim, frames)).astype(np.float32)
        z /= np.sqrt((z * z).mean(0, keepdims=True) + 1e-6)
        s = rng.standard_normal(style_dim).astype(np.float32)
        s /= np.sqrt((s * s).mean() + 1e-6)
        np.save(d / "h.npy", rng.standard_normal((a_dim, frames)).astype(np.float32))
        np.savez(d / "0.latent.npz", z=z, s=s, labels=(rng.random(NUM_LABELS) * 10).astype(np.float32))
