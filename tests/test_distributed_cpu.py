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


@pytest.mark.parametrize("golden,world", [("handler_default", 2), ("multiplex", 2), ("handler_default", 8), ("multiplex", 8)])
def test_two_rank_sharding_and_all_gather(tmp_path, oracle_lib, golden, world):
    """World 2, and world 8 - the split BASELINE configs[2] names: score-group shards of UNEVEN size (the padded
    gather count is agreed on, the tail of a short shard is dropped when the tables are merged)."""
    port = 29500 + (os.getpid() % 2000) + (7 if golden == "multiplex" else 0) + 13 * world
    mp.spawn(_worker, args=(world, port, str(tmp_path), golden), nprocs=world, join=True)
    g = H.load_scoring_golden(golden)
    soa_all = H.soa_for(g, g.config)
    sizes = [b - a for a, b in (shard_bounds(soa_all["score_group_idx"], r, world) for r in range(world))]
    assert sum(sizes) == len(soa_all["precursor_idx"]) and min(sizes) > 0
    if world == 8 and golden == "multiplex":
        assert len(set(sizes)) > 1, sizes  # uneven: whole score groups of several channels
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
    # the product's sharding (alphadia_amd/selection.py::select_sharded, what HipCandidateSelection runs with a
    # communicator) with the oracle as the per-rank selection and gloo as the gather
    from alphadia_amd.selection import select_sharded

    merged = select_sharded(
        len(pdf), int(cfg.candidate_count), rank, world,
        lambda a, b: oracle.select(dia, fragment_columns(fdf, "mz_library"), TS._pack(pdf.iloc[a:b]), cfg,
                                   z["default_kernel"], n_threads=2),
        all_gather_rows)
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


def test_packed_layout_is_the_librarys(tmp_path):
    """``packed_layout`` (what the gloo tests pack and unpack with) IS ``adh_table_layout``: computed tables
    first, 256-byte aligned, the wire prefix ends where the first rebuildable column starts."""
    from alphadia_amd import _abi, runtime
    from alphadia_amd.distributed import LOCAL_COLUMNS, wire_bytes

    fields, total, wire = runtime.table_layout(1000, 12)
    names = [f["name"] for f in fields]
    assert set(names) == set(dict(_abi.output_shapes(1, 1))) | {"fragment_lib_slot", "stat_matched_peaks"}
    seen_local = False
    for f in fields:
        assert f["offset"] % 256 == 0
        seen_local = seen_local or not f["wire"]
        assert f["wire"] != seen_local  # wire columns first, then only local ones
        assert (f["name"] in LOCAL_COLUMNS) == (not f["wire"])
    offsets, nbytes = packed_layout(1000, 12)
    assert nbytes == total and wire_bytes(offsets) == wire
    assert wire == sum((1000 * f["row_elems"] * f["elem_bytes"] + 255) // 256 * 256 for f in fields if f["wire"])
    per_row = sum(f["row_elems"] * f["elem_bytes"] for f in fields if f["wire"])
    assert per_row == 449  # bytes per candidate on the wire and over PCIe at top_k = 12


def _rendezvous_worker(rank, world, path_env, out_dir, nonce, linger_for=None):
    """``linger_for``: a file this process waits for before it exits (rank 0 of a real run is alive while its peers
    read the id - it waits for them inside the collective communicator creation)."""
    os.environ["ADH_RUN_NONCE"] = nonce
    os.environ["MASTER_PORT"] = "29555"
    os.environ.pop("LOCAL_WORLD_SIZE", None)
    if path_env:
        os.environ["ADH_RENDEZVOUS_FILE"] = path_env
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from alphadia_amd import runtime

    uid = runtime.rendezvous_unique_id(rank, world, timeout=30.0, make_id=lambda: bytes([7 + rank]) * 128)
    with open(os.path.join(out_dir, f"id_{rank}"), "wb") as f:
        f.write(uid)
    if linger_for:
        import time

        t0 = time.time()
        while not all(os.path.exists(p) for p in linger_for) and time.time() - t0 < 60:
            time.sleep(0.05)
        if os.environ.get("ADH_TEST_HARD_EXIT"):
            os._exit(0)  # (a crash: no exit handlers, the id file stays)


def test_rendezvous_hands_rank0s_id_to_every_rank(tmp_path):
    """The RCCL id travels through a node-local file: every rank ends up with rank 0's 128 bytes, a stale file
    of another launch (other nonce) or a truncated one is never accepted, the file is private to the user."""
    import stat

    from alphadia_amd import runtime

    # (a) default location: a 0700 directory of this user, the name carries the launch nonce
    os.environ["ADH_RUN_NONCE"] = "launch-A"
    os.environ.pop("ADH_RENDEZVOUS_FILE", None)
    p_a = runtime.rendezvous_path()
    assert stat.S_IMODE(os.stat(os.path.dirname(p_a)).st_mode) == 0o700
    os.environ["ADH_RUN_NONCE"] = "launch-B"
    assert runtime.rendezvous_path() != p_a
    # (b) a stale file of an earlier launch sits where this launch will look: same path (forced), other nonce
    path = str(tmp_path / "rccl_id")
    with open(path, "wb") as f:
        f.write(b"\0" * 32 + b"\x55" * 128)
    procs = [mp.get_context("spawn").Process(target=_rendezvous_worker, args=(r, 3, path, str(tmp_path), "launch-C"))
             for r in (1, 2)]
    for p in procs:
        p.start()
    import time

    time.sleep(1.0)  # the readers are polling: they must not have taken the stale id
    assert not os.path.exists(tmp_path / "id_1") and not os.path.exists(tmp_path / "id_2")
    p0 = mp.get_context("spawn").Process(target=_rendezvous_worker, args=(0, 3, path, str(tmp_path), "launch-C",
                                                                           [str(tmp_path / "id_1"), str(tmp_path / "id_2")]))
    p0.start()
    for p in procs + [p0]:
        p.join(60)
        assert p.exitcode == 0
    ids = [open(tmp_path / f"id_{r}", "rb").read() for r in range(3)]
    assert ids[0] == bytes([7]) * 128 and ids[1] == ids[0] and ids[2] == ids[0]
    assert not os.path.exists(path)  # (rank 0 removes its id file when it exits, if comm_init has not already)
    # (d) a restart under the same launcher: the previous attempt's file carries the SAME nonce but is older than
    # the restarted readers - they wait for the id rank 0 writes now; a symlinked rendezvous directory is refused
    path_d = str(tmp_path / "rccl_id_d")
    out_d = tmp_path / "d"
    out_d.mkdir()
    ctx_mp = mp.get_context("spawn")
    os.environ["ADH_TEST_HARD_EXIT"] = "1"
    first = ctx_mp.Process(target=_rendezvous_worker, args=(0, 2, path_d, str(out_d), "launch-D", [str(out_d / "id_0")]))
    first.start()
    first.join(60)  # the "previous attempt": its rank 0 crashed and left a file with this launch's nonce
    os.environ.pop("ADH_TEST_HARD_EXIT")
    assert first.exitcode == 0 and os.path.exists(path_d)
    os.remove(out_d / "id_0")
    stat_mode = stat.S_IMODE(os.stat(path_d).st_mode)
    assert stat_mode == 0o600
    # (the crash was SECONDS ago - the file is fresh and carries the right nonce; its writer is gone)
    reader = ctx_mp.Process(target=_rendezvous_worker, args=(1, 2, path_d, str(out_d), "launch-D"))
    reader.start()
    time.sleep(1.5)
    assert not os.path.exists(out_d / "id_1")  # right nonce, fresh, but nobody behind it: not this attempt's
    old_t = time.time() - 3600.0
    os.utime(path_d, (old_t, old_t))  # (... and an old one is refused on its age as well)
    writer = ctx_mp.Process(target=_rendezvous_worker, args=(0, 2, path_d, str(out_d), "launch-D", [str(out_d / "id_1")]))
    writer.start()
    for p in (reader, writer):
        p.join(60)
        assert p.exitcode == 0
    assert open(out_d / "id_1", "rb").read() == bytes([7]) * 128
    # (e) staggered start: rank 0 wrote its id seconds before rank 1 even started - rank 1 must take it
    # (the lower bound of an attempt is the launcher's start / the stagger window, not the reader's own start)
    path_e = str(tmp_path / "rccl_id_e")
    out_e = tmp_path / "e"
    out_e.mkdir()
    w0 = ctx_mp.Process(target=_rendezvous_worker, args=(0, 2, path_e, str(out_e), "launch-E", [str(out_e / "id_1")]))
    w0.start()
    t_w = time.time()
    while not os.path.exists(out_e / "id_0") and time.time() - t_w < 60:
        time.sleep(0.05)
    time.sleep(3.0)
    late = ctx_mp.Process(target=_rendezvous_worker, args=(1, 2, path_e, str(out_e), "launch-E"))
    late.start()
    late.join(20)
    w0.join(20)
    assert w0.exitcode == 0
    assert late.exitcode == 0 and open(out_e / "id_1", "rb").read() == bytes([7]) * 128
    # ... and under a launcher (no ADH_RUN_NONCE: the parent's start time bounds the attempt) as well
    assert runtime._attempt_lower_bound() <= runtime._PROCESS_T0 <= time.time()
    # (c) more than one node is refused (the file is node-local)
    os.environ["LOCAL_WORLD_SIZE"] = "4"
    try:
        with pytest.raises(runtime.HipBackendError, match="one node"):
            runtime.rendezvous_unique_id(1, 8, timeout=0.1)
    finally:
        os.environ.pop("LOCAL_WORLD_SIZE", None)
        os.environ.pop("ADH_RUN_NONCE", None)


def _fragcomp_worker(rank, world, port, tmpdir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pandas as pd

    from alphadia_amd.fragcomp import compete_sharded, competition_plan
    from oracle import oracle

    z = np.load(H.golden_path("fragcomp.npz"))
    psm = pd.DataFrame({k[4:]: z[k] for k in z.files if k.startswith("psm_")})
    frag = pd.DataFrame({k[5:]: z[k] for k in z.files if k.startswith("frag_")})
    plan = competition_plan(psm["precursor_idx"].values, psm["rank"].values, psm["mz_observed"].values, psm["proba"].values,
                            frag["precursor_idx"].values, frag["rank"].values, z["cycle"])
    # the product's sharding (what FragmentCompetition runs with a communicator): the oracle competes in this
    # rank's windows, gloo gathers the flags
    valid = compete_sharded(plan, psm["rt_observed"].values[plan.rows], frag["mz_observed"].values, 3, 15, rank, world,
                            lambda *a: oracle.fragcomp(*a, n_threads=2), all_gather_rows)
    np.savez(os.path.join(tmpdir, f"fc{rank}.npz"), valid=valid, rows=plan.rows)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fragment_competition_sharded_by_window(tmp_path, oracle_lib, world):
    """Fragment competition with the DIA windows dealt over the ranks + one gather of the flags == the
    competition on one rank == the reference's survivors (golden fragcomp.npz)."""
    import pandas as pd

    from alphadia_amd.fragcomp import competition_plan

    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_fragcomp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = np.load(H.golden_path("fragcomp.npz"))
    psm = pd.DataFrame({k[4:]: z[k] for k in z.files if k.startswith("psm_")})
    frag = pd.DataFrame({k[5:]: z[k] for k in z.files if k.startswith("frag_")})
    plan = competition_plan(psm["precursor_idx"].values, psm["rank"].values, psm["mz_observed"].values, psm["proba"].values,
                            frag["precursor_idx"].values, frag["rank"].values, z["cycle"])
    one = oracle_lib.fragcomp(plan.window_start, plan.window_stop, psm["rt_observed"].values[plan.rows], plan.frag_start,
                              plan.frag_stop, frag["mz_observed"].values, 3, 15, n_threads=2)
    assert len(plan.window_start) > world and 0 < one.sum() < len(one)
    for r in range(world):
        got = np.load(tmp_path / f"fc{r}.npz")
        assert np.array_equal(got["valid"], one), r
        survivors = psm.iloc[got["rows"][got["valid"]]]
        assert np.array_equal(survivors["precursor_idx"].values, z["surviving_precursor_idx"])


def _shared_case_worker(rank, world, tmpdir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["ADH_RUN_NONCE"] = "shared-case-test-" + os.path.basename(tmpdir)
    import bench

    case = bench.shared_case(300, 40, 2, rank, rank, world)
    np.savez(os.path.join(tmpdir, f"case{rank}.npz"), mz=np.asarray(case.dia.mz_values), inten=np.asarray(case.dia.intensity_values),
             start=np.asarray(case.dia.peak_start_idx_list), rt=np.asarray(case.dia.rt_values),
             cand=case.candidates_df["frame_start"].values, lib=case.library.precursor_df["mz_library"].values,
             mapped=np.array([isinstance(case.dia.mz_values, np.memmap)]))
    if rank == 0:
        import time

        t0 = time.time()
        while not os.path.exists(os.path.join(tmpdir, "case1.npz")) and time.time() - t0 < 120:
            time.sleep(0.1)
        bench.release_shared_case(0, world, 300, 40)


def test_bench_generates_the_run_once_per_node(tmp_path):
    """bench.py --gpus N: local rank 0 generates the synthetic run, the other ranks map it from /dev/shm and
    build only the (cheap, deterministic) library and candidate table themselves - same arrays on every rank."""
    import synthetic as syn

    mp.spawn(_shared_case_worker, args=(2, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "case0.npz"), np.load(tmp_path / "case1.npz")
    ref = syn.make_case(300, 40, config_id=2, per_precursor=3, threads=2)
    assert not a["mapped"][0] and b["mapped"][0]
    for k, v in (("mz", ref.dia.mz_values), ("inten", ref.dia.intensity_values), ("start", ref.dia.peak_start_idx_list),
                 ("rt", ref.dia.rt_values), ("cand", ref.candidates_df["frame_start"].values),
                 ("lib", ref.library.precursor_df["mz_library"].values)):
        assert np.array_equal(a[k], v) and np.array_equal(b[k], v), k
    import glob

    assert not glob.glob("/dev/shm/adh_bench_*shared-case-test*") and not glob.glob("/dev/shm/adh_bench_*_300_40.*")
