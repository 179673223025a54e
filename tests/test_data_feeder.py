"""Pins osu_dreamer_amd/data.py to the reference's LatentDataset / hold_out_mapsets
(data/modules/latent.py:83-149, data/modules/beatmap.py:33-71): tests/golden/feeder_stream.npz holds the sample stream the
REFERENCE produced (oracle/make_golden.py:gen_feeder) on a synthetic dataset this repo's writer re-creates bit for bit.

Where the rank-aware feeder must equal the reference: world_size 1, one worker — same window offsets, same per-map
permutation, same max_per_map cut, same shuffle-buffer order, same tensors.  Where it must differ: mapsets are listed
sorted (the reference's order is the file system's), ranks read disjoint files, and each rank's shuffle seed is offset."""
import os
from pathlib import Path

import numpy as np
import torch

from osu_dreamer_amd.data import LatentDataModule, LatentDataset, hold_out_mapsets, split_mapsets, write_synthetic_dataset

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dataset(tmp_path, fx):
    write_synthetic_dataset(str(tmp_path), n_maps=len(fx["frames"]), frames=[int(f) for f in fx["frames"]],
                            a_dim=int(fx["a_dim"]), emb_dim=int(fx["emb_dim"]), style_dim=int(fx["style_dim"]), seed=int(fx["seed"]))
    return sorted(Path(tmp_path).iterdir())


def _records(ds, mapsets):
    full = {int(m.name): (np.load(m / "h.npy"), np.load(m / "0.latent.npz")) for m in mapsets}
    recs, sums = [], []
    for smp in ds:
        idx = next(i for i, (_, d) in full.items() if np.allclose(d["s"], smp.s.numpy()))
        h, z = full[idx][0], full[idx][1]["z"]
        l = smp.z.shape[-1]
        starts = [i for i in range(z.shape[-1] - l + 1) if np.array_equal(z[:, i:i + l], smp.z.numpy())]
        assert len(starts) == 1 and np.array_equal(h[:, starts[0]:starts[0] + l], smp.h.numpy())
        assert np.array_equal(full[idx][1]["labels"], smp.labels.numpy())
        recs.append((idx, starts[0], l))
        sums.append(float(smp.h.double().sum() + smp.z.double().sum()))
    return np.array(recs), np.array(sums)


def test_stream_equals_reference(tmp_path):
    fx = np.load(os.path.join(GOLDEN, "feeder_stream.npz"))
    mapsets = _dataset(tmp_path, fx)
    for tag, kw in (("plain", dict(seq_len=48, shuffle_buffer_size=1, max_per_map=-1)),
                    ("shuffled", dict(seq_len=32, shuffle_buffer_size=4, max_per_map=3)),
                    ("fullmaps", dict(seq_len=None))):
        torch.manual_seed(int(fx["seed"]))
        recs, sums = _records(LatentDataset(mapsets, **kw), mapsets)
        assert np.array_equal(recs, fx[f"{tag}.stream"]), tag
        assert np.allclose(sums, fx[f"{tag}.sums"], rtol=0, atol=1e-9), tag


def test_split_rule_equals_reference(tmp_path):
    fx = np.load(os.path.join(GOLDEN, "feeder_stream.npz"))
    mapsets = _dataset(tmp_path, fx)
    by_name = {m.name: m for m in mapsets}
    # the reference's rule on the reference's own (file-system) listing order
    listing = [by_name[n] for n in fx["split_listing"].tolist()]
    train, val = split_mapsets(listing, [1] * len(listing), 3, .3)
    assert [m.name for m in train] == fx["split_train"].tolist() and [m.name for m in val] == fx["split_val"].tolist()
    # the module applies the same rule to the SORTED listing
    train, val = hold_out_mapsets(Path(tmp_path), "*.latent.npz", 3, .3)
    st, sv = split_mapsets(mapsets, [1] * len(mapsets), 3, .3)
    assert train == st and val == sv and len(val) == 2 and not set(train) & set(val)


def test_ranks_read_disjoint_files(tmp_path):
    fx = np.load(os.path.join(GOLDEN, "feeder_stream.npz"))
    mapsets = _dataset(tmp_path, fx)
    seen = []
    for rank in range(2):
        torch.manual_seed(5)
        recs, _ = _records(LatentDataset(mapsets, seq_len=32, rank=rank, world_size=2), mapsets)
        seen.append(set(int(r[0]) for r in recs))
    assert not (seen[0] & seen[1]) and seen[0] | seen[1] == set(range(7))
    torch.manual_seed(5)
    world1, _ = _records(LatentDataset(mapsets, seq_len=32), mapsets)
    assert set(int(r[0]) for r in world1) == seen[0] | seen[1]


def test_datamodule_batches(tmp_path):
    fx = np.load(os.path.join(GOLDEN, "feeder_stream.npz"))
    _dataset(tmp_path, fx)
    dm = LatentDataModule(batch_size=2, seq_len=32, num_workers=0, max_val_count=3, max_val_frac=.3, data_path=str(tmp_path))
    torch.manual_seed(0)
    b = next(iter(dm.train_dataloader()))
    assert [tuple(t.shape) for t in b] == [(2, 4, 32), (2, 3, 32), (2, 2), (2, 5)]
    v = next(iter(dm.val_dataloader()))
    assert v[0].shape[0] == 1 and v[0].shape[-1] in (200, 96)
