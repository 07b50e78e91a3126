"""world_size-2 gloo tests of the multi-GPU helpers (runs on CPU): shard -> gather order, histogram all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lbm_amd.pipeline import allreduce_hist, gather_epoch_outputs, gather_pipeline_outputs, perplexity_from_counts, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, E, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tokens_all = torch.arange(E * 5, dtype=torch.int64).view(E, 5)            # stand-in for per-epoch tokens
        feats_all = torch.arange(E * 3, dtype=torch.float32).view(E, 3) * 0.5
        lo, hi = shard_range(E, rank, world)
        tok = gather_epoch_outputs(tokens_all[lo:hi].clone())
        ft = gather_epoch_outputs(feats_all[lo:hi].clone())
        hist = torch.bincount(tokens_all[lo:hi].reshape(-1) % 7, minlength=7)
        hist = allreduce_hist(hist)
        ok = torch.equal(tok, tokens_all) and torch.equal(ft, feats_all) and \
            torch.equal(hist, torch.bincount(tokens_all.reshape(-1) % 7, minlength=7))
        # Scaler-style global standardisation: all-reduced {sum, sum of squares, count} -> the statistics of the whole job
        from lbm_amd.epochs import Scaler
        g = torch.Generator().manual_seed(5)
        xs = torch.randn(E, 3, 11, generator=g, dtype=torch.float64) * torch.tensor([1.0, 2.0, 0.5]).view(1, 3, 1) + 0.25
        mine = xs[lo:hi]
        mom = torch.stack([mine.sum(dim=(0, 2)), (mine * mine).sum(dim=(0, 2)),
                           torch.full((3,), float(mine.shape[0] * mine.shape[2]), dtype=torch.float64)], dim=1)
        sc = Scaler().fit_from_moments(Scaler.reduce_moments(mom))
        ok = ok and torch.allclose(sc.mean_.double(), xs.mean(dim=(0, 2)), atol=1e-6) and \
            torch.allclose(sc.scale_.double(), xs.std(dim=(0, 2), unbiased=False), atol=1e-6)
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([int(ok), perplexity_from_counts(hist)]))
    finally:
        dist.destroy_process_group()


def _run(E, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, E, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(os.path.join(tmp_path, f"ok{i}.npy")) for i in range(2)]
    assert r[0][0] == 1 and r[1][0] == 1
    assert r[0][1] == r[1][1]   # every rank ends with the same global perplexity


def test_gather_equal_shards(tmp_path):
    _run(8, tmp_path)


def test_gather_unequal_shards(tmp_path):
    _run(7, tmp_path)


def _job_arrays(E, C=3, F=5, S=4):
    g = torch.Generator().manual_seed(11)
    return {"tokens": torch.randint(0, 512, (E, S), generator=g, dtype=torch.int64),
            "psd": torch.rand((E, C, F), generator=g, dtype=torch.float32),
            "bandpower": torch.rand((E, 6), generator=g, dtype=torch.float32),
            "keep": (torch.rand((E,), generator=g) > 0.3).to(torch.uint8)}


def _worker_features(rank, world, port, E, out_dir):
    """north_star: all-gather of token AND feature arrays.  Gathered PSD / band features / keep mask must come back in
    np.concatenate order (training/model.py:423), with and without dropping rejected epochs (unequal shards)."""
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = _job_arrays(E)
        lo, hi = shard_range(E, rank, world)
        mine = {k: v[lo:hi].clone() for k, v in full.items()}
        got = gather_pipeline_outputs(mine)
        ok = all(torch.equal(got[k], full[k]) for k in full)
        kept = gather_pipeline_outputs(mine, drop_rejected=True)
        sel = full["keep"].to(torch.bool)
        ok = ok and all(torch.equal(kept[k], full[k][sel]) for k in full)
        # bench.py's grouped exchange (weak scaling: every rank holds E epochs; int16 tokens travel as bytes)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        per_rank = []
        for w in range(world):
            g = torch.Generator().manual_seed(100 + w)
            per_rank.append({"tokens": torch.randint(0, 512, (E, 4), generator=g).to(torch.int16),
                             "psd": torch.rand((E, 3, 5), generator=g), "keep": (torch.rand((E,), generator=g) > 0.5).to(torch.uint8)})
        xch = bench.Exchange(dist, "gloo", world, per_rank[rank], torch.device("cpu"))
        bounds = [0, E // 3, E // 3 + 1, E]                   # ragged groups, one of a single epoch
        for a, b_ in zip(bounds[:-1], bounds[1:]):
            xch.gather_group(a, b_)
        hist = torch.full((7,), rank + 1, dtype=torch.int64)
        xch.finish(hist)
        for k in per_rank[0]:
            want = torch.cat([per_rank[w][k] for w in range(world)], dim=0)
            ok = ok and torch.equal(xch.gathered(k), want) and xch.gathered(k).dtype == want.dtype
        ok = ok and torch.equal(hist, torch.full((7,), sum(range(1, world + 1)), dtype=torch.int64))
        np.save(os.path.join(out_dir, f"feat{rank}.npy"), np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


def test_gather_features_in_concatenate_order(tmp_path):
    for E in (9, 8):
        port = _free_port()
        mp.spawn(_worker_features, args=(2, port, E, str(tmp_path)), nprocs=2, join=True)
        assert all(int(np.load(os.path.join(tmp_path, f"feat{i}.npy"))[0]) == 1 for i in range(2))


def test_single_process_passthrough():
    t = torch.arange(6).view(3, 2)
    assert gather_epoch_outputs(t) is t
    h = torch.tensor([1, 2, 3])
    assert allreduce_hist(h) is h
