"""CPU, world_size 2 over gloo: candidate sharding by score group + ONE all-gather of the
packed tables reproduces the unsharded result.  The per-rank tables are filled by the
oracle here (no GPU in this container); on the GPU the same code path is fed by the kernels
(bench.py --gpus N)."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from alphadia_amd.distributed import (
    merge_gathered,
    packed_layout,
    precursor_bounds,
    rebuild_local_columns,
    shard_bounds,
    slice_soa,
    window_owner,
)
from torch_transport import DeviceTables, PipelinedGather, all_gather_rows, all_gather_tables


def test_shard_bounds_keep_score_groups_intact():
    sg = np.array([0, 0, 0, 1, 2, 2, 3, 3, 3, 3, 4], dtype=np.uint32)
    for world in (1, 2, 3, 4, 8):
        cuts = [shard_bounds(sg, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(sg)
        for (a0, b0), (a1, b1) in zip(cuts[:-1], cuts[1:]):
            assert b0 == a1
        for a, b in cuts:
            if 0 < a < len(sg):
                assert sg[a] != sg[a - 1]
    assert shard_bounds(np.zeros(0, np.uint32), 0, 2) == (0, 0)


def test_packed_layout_is_aligned_and_disjoint():
    offsets, nbytes = packed_layout(1000, 12)
    spans = sorted((off, off + int(np.prod(shape)) * dt.itemsize) for off, shape, dt in offsets.values())
    for (a0, b0), (a1, b1) in zip(spans[:-1], spans[1:]):
        assert b0 <= a1
    assert all(off % 256 == 0 for off, _, _ in offsets.values()) and spans[-1][1] <= nbytes


def _worker(rank, world, port, tmpdir, golden="handler_default"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle

    g = H.load_scoring_golden(golden)
    soa_all = H.soa_for(g, g.config)
    a, b = shard_bounds(soa_all["score_group_idx"], rank, world)
    local, _ = H.oracle_score(oracle, g, g.config, soa=slice_soa(soa_all, a, b))
    n_rows = -(-len(soa_all["precursor_idx"]) // world) + 3
    tables = DeviceTables(n_rows, int(g.config.top_k_fragments), "cpu", with_stats=False)
    tables.load_host(local)
    # only the computed tables travel; candidate ids are rebuilt from the candidate table
    gathered = all_gather_tables(tables.wire, world)
    rows = [shard_bounds(soa_all["score_group_idx"], r, world) for r in range(world)]
    merged = merge_gathered([tables.to_host(gathered[r]) for r in range(world)], [e - s for s, e in rows])
    assert "precursor_idx" not in merged and "fragment_rank" not in merged and "fragment_mz" not in merged
    from alphadia_amd.scoring import fragment_columns

    merged = rebuild_local_columns(merged, soa_all["precursor_idx"], soa_all["rank"], soa_all["flags"],
                                   frag_start=soa_all["frag_start_idx"],
                                   fragment_cols=fragment_columns(g.library.fragment_df, "mz_library"))
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), **merged)
    dist.destroy_process_group()


@pytest.mark.parametrize("golden", ["handler_default", "multiplex"])
def test_two_rank_sharding_and_all_gather(tmp_path, oracle_lib, golden):
    world = 2
    port = 29500 + (os.getpid() % 2000) + (7 if golden == "multiplex" else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path), golden), nprocs=world, join=True)
    g = H.load_scoring_golden(golden)
    full, _ = H.oracle_score(oracle_lib, g, g.config)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for k, v in full.items():
            assert np.array_equal(z[k], v, equal_nan=True), (r, k)


def test_precursor_and_window_partitions_cover_everything():
    for n, world in ((0, 2), (1, 2), (7, 3), (100, 8), (100001, 8)):
        cuts = [precursor_bounds(n, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(b0 == a1 for (_, b0), (a1, _) in zip(cuts[:-1], cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1
    own = window_owner(61, 8)
    assert own.shape == (61,) and set(own) == set(range(8)) and np.bincount(own).max() - np.bincount(own).min() <= 1


def _select_worker(rank, world, port, tmpdir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_selection as TS
    from alphadia_amd.scoring import fragment_columns
    from oracle import oracle

    z, dia, fdf, pdf = TS._load()
    pdf = pdf.sort_values("precursor_idx").reset_index(drop=True)
    cfg = TS._cfg(z, "default")
    a, b = precursor_bounds(len(pdf), rank, world)
    local = oracle.select(dia, fragment_columns(fdf, "mz_library"), TS._pack(pdf.iloc[a:b]), cfg,
                          z["default_kernel"], n_threads=2)
    cc = int(cfg.candidate_count)
    rows = [(precursor_bounds(len(pdf), r, world)[1] - precursor_bounds(len(pdf), r, world)[0]) * cc
            for r in range(world)]
    merged = {k: all_gather_rows(v, rows) for k, v in local.items()}
    np.savez(os.path.join(tmpdir, f"sel{rank}.npz"), **merged)
    dist.destroy_process_group()


def test_two_rank_selection_sharding(tmp_path, oracle_lib):
    """Candidate selection sharded by precursor + one gather per column == the unsharded table."""
    import test_selection as TS
    from alphadia_amd.scoring import fragment_columns

    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_select_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z, dia, fdf, pdf = TS._load()
    full = oracle_lib.select(dia, fragment_columns(fdf, "mz_library"), TS._pack(pdf), TS._cfg(z, "default"),
                             z["default_kernel"], n_threads=2)
    for r in range(world):
        got = np.load(tmp_path / f"sel{r}.npz")
        for k, v in full.items():
            assert np.array_equal(got[k], v), (r, k)


def _pipeline_worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg = PipelinedGather(50, 4, "cpu", world, with_stats=False)
    seen = []
    for batch in range(5):
        tables = pg.begin()
        tables.zero_()
        tables.buffer[:16] = batch * 10 + rank  # stands in for the kernels filling the tables
        pg.end()
        if batch >= 2:  # the slot about to be reused must already hold its complete gather
            prev = pg.gathered[pg.slot ^ 1]
            w = pg.pending[pg.slot ^ 1]
            if w is not None:
                w.wait()
            seen.append(prev[:, 0].clone().numpy())
    last = pg.finish()
    np.savez(os.path.join(tmpdir, f"pipe{rank}.npz"), last=last[:, :16].numpy(), seen=np.stack(seen))
    dist.destroy_process_group()


def test_pipelined_gather_two_ranks(tmp_path):
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        z = np.load(tmp_path / f"pipe{r}.npz")
        assert np.array_equal(z["last"], np.array([[40] * 16, [41] * 16], dtype=np.uint8))
        # while batch b is being filled, the other slot holds the gather of batch b - 1
        assert np.array_equal(z["seen"], np.array([[10, 11], [20, 21], [30, 31]], dtype=np.uint8))
