"""GPU tests of the host -> host pipeline behind ``adh_score_candidates``: chunking, page-locked
buffers, the plan built on the device, tables left in HBM, the RCCL all-gather (one rank) and the
invalidation rules of the resident candidate table."""

import ctypes as C
import os

import numpy as np
import pytest

import helpers as H
import synthetic as syn
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled

pytestmark = pytest.mark.gpu

TABLES = list(H.OUT_NAMES) + ["stat_matched_peaks", "fragment_lib_slot"]


@pytest.fixture(scope="module")
def ctx():
    from alphadia_amd import runtime

    return runtime.get_context(0)


@pytest.fixture(scope="module")
def case():
    return syn.make_case(2500, 400, config_id=2, per_precursor=3, threads=8)


def _cfg(**kw):
    cfg = CandidateScoringConfig()
    cfg.update(dict(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                         experimental_xic=True), **kw))
    return cfg


def _stage(ctx, case):
    ctx.stage_run(case.dia, force=True)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"), force=True)


def _same(a: dict, b: dict, names=TABLES):
    for k in names:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("experimental_xic", [True, False])
def test_chunked_pipeline_is_bitwise_the_single_pass(ctx, case, monkeypatch, experimental_xic):
    """Chunk boundaries (short first chunk, ragged last one), page-locked or pageable host buffers:
    every table identical to the one-chunk result."""
    cfg = _cfg(experimental_xic=experimental_xic).to_jitclass()
    _stage(ctx, case)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    n = len(soa["precursor_idx"])
    monkeypatch.setenv("ADH_CHUNK", str(10 * n))
    ref = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
    assert ref["valid"].sum() > n // 4
    for chunk in (1024, 1777, 4096):
        monkeypatch.setenv("ADH_CHUNK", str(chunk))
        got = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
        _same(got, ref)
        pinned_soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library", pool=ctx.pinned)
        got = ctx.score_host(pack_assembled(pinned_soa), cfg, with_stats=True, reuse_buffers=True)
        _same(got, ref)
    # the order of the copy-in burst (behind / in front of the plan of chunk 1) and the length of the first chunk
    monkeypatch.setenv("ADH_CHUNK", "1500")
    for env in (dict(ADH_H2D_BURST_LATE="0"), dict(ADH_FIRST_CHUNK_DIV="2"), dict(ADH_FIRST_CHUNK_DIV="1"),
                dict(ADH_FIRST_CHUNK_DIV="7", ADH_H2D_BURST_LATE="0")):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            got = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
            _same(got, ref)


def test_host_rebuilt_columns_equal_the_copied_ones(ctx, case, monkeypatch):
    """adh_score_candidates copies only the computed tables back; ids and library columns are rebuilt in the
    caller's buffers from fragment_lib_slot by host threads while later chunks are in flight.  Byte-identical
    to copying every table (ADH_DEBUG_COPY_ALL), for chunked calls, skipped rows, with and without a host
    buffer for the slots, at any thread count; and a third of the bytes stay off the link."""
    from alphadia_amd import _abi

    cfg = _cfg().to_jitclass()
    _stage(ctx, case)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    n = len(soa["precursor_idx"])
    soa["flags"] = soa["flags"].copy()
    soa["flags"][::17] |= 1  # ADH_FLAG_SKIP: such rows stay zero everywhere
    monkeypatch.setenv("ADH_DEBUG_COPY_ALL", "1")
    monkeypatch.setenv("ADH_CHUNK", "1500")
    ctx.d2h_bytes(reset=True)
    ref = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
    copied = ctx.d2h_bytes(reset=True)
    monkeypatch.delenv("ADH_DEBUG_COPY_ALL")
    assert ref["valid"].sum() > n // 4 and (ref["precursor_idx"][::17] == 0).all()
    # (round 6) a team of fewer than six threads does not rebuild: the device writes the columns, the link carries them
    # (host_rebuild_pays; ADH_REBUILD_MIN_THREADS moves the threshold)
    monkeypatch.setenv("ADH_HOST_THREADS", "2")
    ctx.d2h_bytes(reset=True)
    got = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
    assert ctx.d2h_bytes(reset=True) == copied
    _same(got, ref)
    monkeypatch.setenv("ADH_REBUILD_MIN_THREADS", "0")  # from here on: the team rebuilds whatever its size
    for threads, chunk in (("1", "1500"), ("3", "1777"), ("16", str(10 * n))):
        monkeypatch.setenv("ADH_HOST_THREADS", threads)
        monkeypatch.setenv("ADH_CHUNK", chunk)
        ctx.d2h_bytes(reset=True)
        got = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
        rebuilt = ctx.d2h_bytes(reset=True)
        _same(got, ref)
        assert rebuilt < 0.75 * copied
        # production form: page-locked buffers, no host table for the slots (an internal staging buffer is used)
        got = ctx.score_host(pack_assembled(soa), cfg, reuse_buffers=True)
        _same(got, ref, [k for k in TABLES if k in got])
        # opt-in: the fragment tables leave packed (filled slots only, offsets first) and the host team writes
        # the padded rows - fewer bytes still, the same tables
        with monkeypatch.context() as mp:
            mp.setenv("ADH_COMPACT_COPY_OUT", "1")
            ctx.d2h_bytes(reset=True)
            got = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
            packed = ctx.d2h_bytes(reset=True)
            _same(got, ref)
            assert packed < 0.9 * rebuilt
            got = ctx.score_host(pack_assembled(soa), cfg, reuse_buffers=True)
            _same(got, ref, [k for k in TABLES if k in got])


def test_device_plan_matches_oracle_on_mixed_classes(ctx, oracle_lib, monkeypatch):
    """Candidates of every kernel class (cycle counts 3..40, one and two observations, short library
    slices, skipped score groups) in one table, several chunks: the device-built plan routes each
    to its kernel and the rows come back in input order."""
    g = H.load_scoring_golden("edges")
    monkeypatch.setenv("ADH_CHUNK", "1024")
    cfg = g.config
    soa = H.soa_for(g, cfg)
    _stage(ctx, g)
    got = ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=True)
    exp, _ = H.oracle_score(oracle_lib, g, cfg, soa=soa, with_stats=True)
    assert np.array_equal(got["valid"], exp["valid"])
    assert np.array_equal(got["stat_matched_peaks"], exp["stat_matched_peaks"])
    v = exp["valid"].astype(bool)
    assert H.rel_err(got["features"][v][:, [0, 1, 2, 3, 17, 20, 28]], exp["features"][v][:, [0, 1, 2, 3, 17, 20, 28]]).max() == 0


def test_tables_stay_in_hbm_and_gather_with_one_rank(ctx, case):
    """``device_tables`` is the hand-over to an on-device stage; with a communicator attached the
    computed tables are all-gathered (world = 1: the gathered slice is the rank's own)."""
    cfg = _cfg().to_jitclass()
    _stage(ctx, case)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    n = len(soa["precursor_idx"])
    host = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
    dev = ctx.device_tables_to_host()
    _same(dev, host)
    from alphadia_amd import runtime

    uid = C.create_string_buffer(128)
    runtime._check(runtime.lib.adh_comm_unique_id(uid), "adh_comm_unique_id")
    ctx.comm_init(0, 1, n + 100, unique_id=uid.raw)
    try:
        for _ in range(3):  # both table slots, gather of call i overlapping call i + 1
            host2 = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
        _same(host2, host)
        ctx.comm_wait()
        wire = ctx.gathered_tables(0, rows=n)
        for k in ("valid", "features", "fragment_mz_observed", "fragment_height", "fragment_intensity",
                  "fragment_mass_error", "fragment_correlation", "fragment_lib_slot"):
            assert np.array_equal(wire[k], host[k], equal_nan=True), k
        assert "precursor_idx" not in wire and "fragment_mz" not in wire  # rebuilt locally, never on the wire
        assert ctx.all_reduce_max(3.5) == 3.5
        ctx.barrier()
    finally:
        ctx.comm_destroy()
    # the columns that did not travel come back from the candidate table and the staged library
    from alphadia_amd.distributed import rebuild_local_columns

    full = rebuild_local_columns(wire, soa["precursor_idx"], soa["rank"], soa["flags"], frag_start=soa["frag_start_idx"],
                                 fragment_cols=fragment_columns(case.library.fragment_df, "mz_library"))
    _same(full, host, names=H.OUT_NAMES)


def test_selection_and_fragment_competition_under_a_communicator(ctx, case):
    """The two other stages that shard - candidate selection by precursor range, fragment competition by DIA
    window - take their rank and world from the communicator on the context and exchange their results through
    `adh_comm_all_gather_host`.  With a one-rank RCCL communicator (what a single GPU offers) the exchange runs
    for real and both operators must return what they return without one; the row partitions themselves are
    covered at world 2 and 3 over gloo (tests/test_distributed_cpu.py)."""
    import pandas as pd

    from alphadia_amd import runtime
    from alphadia_amd.fragcomp import FragmentCompetition
    from alphadia_amd.selection import CandidateSelectionConfig, HipCandidateSelection

    z = np.load(H.golden_path("fragcomp.npz"))
    psm = pd.DataFrame({k[4:]: z[k] for k in z.files if k.startswith("psm_")})
    frag = pd.DataFrame({k[5:]: z[k] for k in z.files if k.startswith("frag_")})
    scfg = CandidateSelectionConfig()
    scfg.update(dict(rt_tolerance=60.0, candidate_count=2))
    names = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
                 fragment_mz_column="mz_library")
    selector = HipCandidateSelection(case.dia, case.library.precursor_df, case.library.fragment_df, scfg, device=0, **names)
    plain_sel = selector()
    plain_fc = FragmentCompetition(device=0)(psm, frag, z["cycle"])
    uid = C.create_string_buffer(128)
    runtime._check(runtime.lib.adh_comm_unique_id(uid), "adh_comm_unique_id")
    ctx.comm_init(0, 1, 1000, unique_id=uid.raw)
    try:
        rows = np.arange(21, dtype=np.float32).reshape(7, 3)
        assert np.array_equal(ctx.all_gather_rows(rows, [7]), rows)           # through ncclAllGather
        assert ctx.all_gather_rows(np.zeros((0, 3), np.uint8), [0]).shape == (0, 3)
        with pytest.raises(ValueError):
            ctx.all_gather_rows(rows, [7, 7])                                   # one entry per rank
        comm_sel = selector()
        comm_fc = FragmentCompetition(device=0)(psm, frag, z["cycle"])
        with pytest.raises(runtime.HipBackendError, match="communicator of 2 ranks"):
            FragmentCompetition(device=0, rank=0, world=2)(psm, frag, z["cycle"])
    finally:
        ctx.comm_destroy()
    pd.testing.assert_frame_equal(comm_sel, plain_sel)
    pd.testing.assert_frame_equal(comm_fc, plain_fc)
    assert len(plain_sel) > 1000 and np.array_equal(plain_fc["precursor_idx"].values, z["surviving_precursor_idx"])


def test_uneven_shards_share_one_layout_under_a_communicator(ctx, case):
    """All ranks must lay their tables out alike (the all-gather moves equal byte counts): the layout follows
    ``max_rows_per_rank`` and the agreed table width, not the shard.  A short shard scored through a
    communicator sized for a longer one gives the rows of the unsharded call; RCCL itself reports the ranks it
    sees; more rows than the communicator was sized for are refused (mismatched counts are undefined)."""
    from alphadia_amd import runtime
    from alphadia_amd.distributed import shard_bounds, slice_soa

    cfg = _cfg().to_jitclass()
    _stage(ctx, case)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    n = len(soa["precursor_idx"])
    full = ctx.score_host(pack_assembled(soa), cfg, with_stats=True)
    sg = np.cumsum(np.r_[0, np.diff(soa["precursor_idx"].astype(np.int64)) != 0])
    bounds = [shard_bounds(sg, r, 3) for r in range(3)]
    longest = max(b - a for a, b in bounds)
    a, b = bounds[1][0], bounds[1][0] + (bounds[1][1] - bounds[1][0]) // 2  # a shard half as long as the longest
    uid = C.create_string_buffer(128)
    runtime._check(runtime.lib.adh_comm_unique_id(uid), "adh_comm_unique_id")
    ctx.comm_init(0, 1, longest, unique_id=uid.raw)
    try:
        assert ctx.comm_info() == (0, 1)
        part = ctx.score_host(pack_assembled(slice_soa(soa, a, b)), cfg, with_stats=True)
        for k in TABLES:
            assert np.array_equal(part[k], full[k][a:b], equal_nan=True), k
        ctx.comm_wait()
        wire = ctx.gathered_tables(0, rows=b - a)
        assert np.array_equal(wire["features"], full["features"][a:b], equal_nan=True)
        # the gathered slice of a rank is as long as the layout of the LONGEST shard says
        _, _, wire_bytes = runtime.table_layout(longest, part["fragment_mz"].shape[1])
        assert wire_bytes % 256 == 0 and wire_bytes > (b - a) * 449
        with pytest.raises(runtime.HipBackendError, match="max_rows_per_rank"):
            ctx.score_host(pack_assembled(soa), cfg)  # n > longest
    finally:
        ctx.comm_destroy()
    assert ctx.comm_info() == (0, 1)


def test_restaging_invalidates_the_resident_table(ctx, case):
    """A candidate table uploaded for one run / library must not be scored against another one
    (its bounds were checked against the old arrays)."""
    from alphadia_amd.runtime import HipBackendError

    cfg = _cfg().to_jitclass()
    _stage(ctx, case)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    host = ctx.score_host(pack_assembled(soa), cfg)
    view = ctx.device_tables()
    ctx.upload_candidates(pack_assembled(soa))
    ctx.zero_device_tables(ctx.stream_handle())
    ctx.score_uploaded(cfg, view, ctx.stream_handle())
    ctx.synchronize()
    _same(ctx.device_tables_to_host(), host, names=H.OUT_NAMES)
    small = case.library.fragment_df.iloc[: len(case.library.fragment_df) // 2]
    ctx.stage_fragments(*fragment_columns(small, "mz_library"), force=True)
    with pytest.raises(HipBackendError, match="no candidate table uploaded"):
        ctx.score_uploaded(cfg, view, ctx.stream_handle())
    ctx.upload_candidates(pack_assembled(soa))
    with pytest.raises(HipBackendError, match="fragment slice"):  # the plan re-validates against the new library
        ctx.score_uploaded(cfg, view, ctx.stream_handle())
    _stage(ctx, case)
    ctx.upload_candidates(pack_assembled(soa))
    ctx.stage_run(case.dia, force=True)
    with pytest.raises(HipBackendError, match="no candidate table uploaded"):
        ctx.score_uploaded(cfg, view, ctx.stream_handle())


def test_event_list_stays_bounded(ctx, case):
    """Timing events are folded into running sums: many calls without reading the timers do not
    accumulate events."""
    cfg = _cfg().to_jitclass()
    _stage(ctx, case)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    sub = {k: (v[:600] if isinstance(v, np.ndarray) and v.shape[:1] == (len(soa["precursor_idx"]),) else v)
           for k, v in soa.items()}
    packed = pack_assembled(sub)
    ctx.kernel_time_ms(reset=True)
    for _ in range(300):
        ctx.score_host(packed, cfg, reuse_buffers=True)
    g, f, n = ctx.kernel_time_ms(reset=True)
    assert n == 300 and g > 0 and f > 0


def test_hip_dense_tile_matches_reference_alpharaw(ctx):
    """The tile the production gather kernel builds, dumped through ``adh_debug_get_dense``,
    against the 32 ``AlphaRawJIT.get_dense`` outputs of the reference: intensity plane bit for
    bit in all cases, absolute-m/z plane bit for bit where the golden was made with
    ``absolute_masses=True`` (the only mode the scoring path uses, candidate.py:213-246)."""
    z = np.load(H.golden_path("get_dense_alpharaw.npz"))
    ctx.stage_run(H.dia_from_npz(z), force=True)
    hits = 0
    for i in range(int(z["n_cases"])):
        fl, quad, e = z[f"q{i}_frame_limits"], z[f"q{i}_quad"], z[f"q{i}_dense"]
        dense, obs = ctx.debug_get_dense(fl[0, 0], fl[0, 1], z[f"q{i}_mz"], z[f"q{i}_tol"], quad[0, 0], quad[0, 1])
        assert np.array_equal(obs, z[f"q{i}_pidx"]), i
        assert dense.shape == e.shape[:3] + (1,) + e.shape[4:], (dense.shape, e.shape)
        for slot in (0, 1):  # the reference writes the same value to both scan slots
            assert np.array_equal(dense[0, :, :, 0, :], e[0, :, :, slot, :]), f"intensity, case {i}"
            if bool(z[f"q{i}_absolute"]):
                assert np.array_equal(dense[1, :, :, 0, :], e[1, :, :, slot, :]), f"m/z, case {i}"
        hits += int((e[0] > 0).sum())
    assert hits > 100


def test_hip_dense_tile_matches_reference_timstof(ctx):
    """The same for the 16 ``TimsTOFTransposeJIT.get_dense`` goldens (both planes, every scan)."""
    import test_oracle_golden as TG

    z, dia, *_ = TG._tims_golden()
    ctx.stage_run(dia, force=True)
    hits = 0
    for i in range(int(z["n_cases"])):
        fl, sl, quad, e = z[f"q{i}_frame_limits"], z[f"q{i}_scan_limits"], z[f"q{i}_quad"], z[f"q{i}_dense"]
        dense, obs = ctx.debug_get_dense(fl[0, 0], fl[0, 1], z[f"q{i}_mz"], z[f"q{i}_tol"], quad[0, 0], quad[0, 1],
                                         scan_start=sl[0, 0], scan_stop=sl[0, 1])
        if e.size == 0:  # no push matches the quadrupole: bruker_jit.py:363-366
            assert dense.size == 0
            continue
        assert np.array_equal(obs, z[f"q{i}_pidx"]), i
        assert dense.shape == e.shape and np.array_equal(dense, e), f"case {i}"
        hits += int((e[0] > 0).sum())
    assert hits > 20


def test_staging_cache_notices_edited_arrays(ctx, case):
    """``stage_run`` skips the upload when the very same arrays are staged already; an in-place edit
    of a staged array (same address, same shape) must not be mistaken for "already staged"."""
    import copy

    dia = copy.copy(case.dia)
    dia.intensity_values = case.dia.intensity_values.copy()
    assert ctx.stage_run(dia, force=True) is True
    assert ctx.stage_run(dia) is False
    dia.intensity_values[::7] *= 2.0
    assert ctx.stage_run(dia) is True
    assert ctx.stage_run(dia) is False


@pytest.mark.parametrize("world", [4, 8])
def test_score_group_shards_reassemble_to_the_unsharded_tables(ctx, world):
    """BASELINE configs[4] in small (3-plex + reference channel, channels of one elution group scored
    as ONE score group, groups without their reference channel skipped) cut into `world` score-group
    shards the way bench.py / a multi-GPU search does: every shard goes through the production call
    on its own and the shards, put back in rank order, are the unsharded tables bit for bit - no group
    is split and no row depends on its neighbours."""
    from alphadia_amd.distributed import merge_gathered, shard_bounds, slice_soa

    g = H.load_scoring_golden("multiplex")
    cfg = g.config
    soa = H.soa_for(g, cfg)
    _stage(ctx, g)
    full = ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=True)
    bounds = [shard_bounds(soa["score_group_idx"], r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == len(soa["precursor_idx"])
    parts = []
    for a, b in bounds:
        if a > 0:  # a cut never falls inside a score group
            assert soa["score_group_idx"][a] != soa["score_group_idx"][a - 1]
        parts.append(ctx.score_host(pack_assembled(slice_soa(soa, a, b)), cfg.to_jitclass(), with_stats=True))
    merged = merge_gathered(parts, [b - a for a, b in bounds])
    _same(merged, full)
    assert full["valid"].sum() > 50 and (soa["flags"] != 0).sum() > 0
