"""BASELINE.json's configurations at their full sizes, inside the GPU suite.

configs[2] (1e6-precursor library x 3 candidates vs the 2 h run, on one GPU) and configs[1] (its first
100 000 precursors: the same run, the 100k library) through size-independent properties - permutation
invariance, sub-table == rows of the full table, planted precursors found - and a strided oracle sample that
includes the ppm features; configs[3] (918 scans x 2 000 cycles, ion mobility) with its 200 000 precursors and an
oracle sample; configs[4] (75 000 elution groups x 4 label channels, the requantification config) with an
oracle sample and its four-way score-group split.  The run generators are the bench's (tests/synthetic.py);
generation dominates the time.
"""

import os

import numpy as np
import pytest

import helpers as H
import synthetic as syn
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled
from test_gpu_parity import PPM_ABS_TOL_ORACLE, compare

pytestmark = pytest.mark.gpu

THREADS = max(1, min(32, os.cpu_count() or 8))


@pytest.fixture(scope="module")
def ctx():
    from alphadia_amd import runtime

    return runtime.get_context(0)


def _cfg():
    cfg = CandidateScoringConfig()
    # ClassicExtractionHandler defaults (extraction_handler.py:370-376,400-409), as bench.py
    cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1, precursor_mz_tolerance=10,
                    fragment_mz_tolerance=15, exclude_shared_ions=True, quant_window=3, quant_all=True,
                    experimental_xic=True, top_k_fragments=12))
    return cfg


ORACLE_STRIDE_CONFIG2 = int(os.environ.get("ADH_TEST_ORACLE_STRIDE", "5"))
# The knife-edge masks of compare() (features 16, 18, 19: 0 / 0 forms that hang on the last bit of a float64 exp).
# Most ELIGIBLE rows are candidates without signal, where both sides agree anyway; what is bounded here is the share of
# the compared rows a mask actually RESCUED - rows that would have failed the comparison of that feature.  Measured on
# the pool (gpurun_out/parity_masks.jsonl, round 6) - see DESIGN.md section 5; the bound is the measured share plus a
# margin, not a quarter of the sample.
RESCUED_BOUND = 5e-4  # measured: 0 of 546 158 rows (configs[2]), 38 of 469 512 = 8.1e-5 (configs[3], feature 19), 0 of 209 666 (configs[4])


def _bound_masked(m: dict) -> None:
    print(f"[full size] knife-edge rows of {m['rows']} compared: {m}")
    for f, n in m["rescued"].items():
        assert n <= RESCUED_BOUND * m["rows"], (f, m)


def _rows(soa: dict, idx) -> dict:
    n = len(soa["precursor_idx"])
    return {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (n,) else v) for k, v in soa.items()}


@pytest.fixture(scope="module")
def headline(ctx):
    """The bench workload: 1e6 precursors x 3 candidates against 4 800 cycles x 61 spectra (4.9e8 peaks)."""
    case = syn.make_case(1_000_000, 4800, config_id=2, per_precursor=3, threads=THREADS)
    cfg = _cfg()
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    ctx.stage_run(case.dia, force=True)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"), force=True)
    got = ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=True)
    return case, cfg, soa, got


def test_config2_full_size_one_gpu(ctx, oracle_lib, headline):
    """configs[2] on one GPU: all 3 000 000 candidates; every 5th row (600 000) against the oracle (all 46 features
    incl. the ppm ones, every fragment table, matched-peak counts; bench.py compares all 3 M); planted precursors
    are found."""
    case, cfg, soa, got = headline
    n = len(soa["precursor_idx"])
    assert n == 3_000_000 and case.dia.mz_values.size > 4.5e8
    assert 0.85 < got["valid"].mean() <= 1.0
    idx = np.arange(0, n, ORACLE_STRIDE_CONFIG2)
    exp, _ = H.oracle_score(oracle_lib, case, cfg, soa=_rows(soa, idx), n_threads=THREADS, with_stats=True)
    compare({k: v[idx] for k, v in got.items()}, exp, PPM_ABS_TOL_ORACLE)
    # the knife-edge masks of compare() (features 16, 18, 19 where the operands are exactly equal) must stay a
    # small share of the sample: a comparison that masks most rows would not be one
    m = compare.last_masked
    print(f"[full size] knife-edge rows of {m['rows']} compared: {m}")
    _bound_masked(m)
    assert np.array_equal(got["stat_matched_peaks"][idx], exp["stat_matched_peaks"])
    planted = case.apex_cycle[soa["precursor_idx"]] >= 0
    r0 = planted & (soa["rank"] == 0)
    assert got["valid"][r0].mean() > 0.99
    assert np.nanmean(got["features"][r0][:, 20]) > 0.9 > np.nanmean(got["features"][~planted][:, 20])
    # two observations (precursor on an isolation-window boundary) are part of the workload
    assert 0.05 < (got["features"][got["valid"].astype(bool)][:, 17] == 2).mean() < 0.4


def test_config1_full_size(ctx, headline):
    """configs[1]: the 100 000-precursor library (the first 100 000 precursors) against the same 2 h run.
    Scored on its own it gives the rows of the big table, in any order (other chunk cuts, other plan)."""
    case, cfg, soa, got = headline
    n1 = 300_000
    assert len(np.unique(soa["precursor_idx"][:n1])) == 100_000
    sub = _rows(soa, np.arange(n1))
    alone = ctx.score_host(pack_assembled(sub), cfg.to_jitclass(), with_stats=True)
    for k in got:
        assert np.array_equal(alone[k], got[k][:n1], equal_nan=True), k
    perm = np.random.default_rng(1).permutation(n1)
    shuffled = ctx.score_host(pack_assembled(_rows(sub, perm)), cfg.to_jitclass(), with_stats=True)
    for k in got:
        assert np.array_equal(shuffled[k], got[k][:n1][perm], equal_nan=True), k


@pytest.mark.parametrize("world", [3, 8])
def test_config2_score_group_shards_reassemble(ctx, headline, world):
    """The 3 000 000-row table cut into score-group shards as `bench.py --gpus N` cuts it (N = 3: uneven; N = 8:
    the split configs[2] names), every
    shard scored on its own - its own chunk cuts of the host -> host pipeline, its own plan - gives the rows of
    the one-GPU table bit for bit: sharding and chunking at a size where both are in play."""
    from alphadia_amd.distributed import shard_bounds, slice_soa

    case, cfg, soa, got = headline
    n, pos = len(soa["precursor_idx"]), 0
    for rank in range(world):
        a, b = shard_bounds(soa["score_group_idx"], rank, world)
        assert a == pos and b > a
        pos = b
        part = ctx.score_host(pack_assembled(slice_soa(soa, a, b)), cfg.to_jitclass(), with_stats=True)
        for k in got:
            assert np.array_equal(part[k], got[k][a:b], equal_nan=True), (rank, k)
    assert pos == n


def test_config3_full_size_ion_mobility(ctx, oracle_lib):
    """configs[3] as specified: 918 scans x 2 000 cycles (1 MS1 + 8 diaPASEF frames per cycle), 200 000
    precursors x 3 candidates of 17-39 scans x 7-29 cycles; EVERY candidate against the oracle; repeated runs
    identical; permutation invariance."""
    case = syn.make_timstof_case(
        n_precursors=200_000, n_cycles=2000, config_id=4, per_precursor=3, n_ms2_frames=8, windows_per_frame=3,
        scan_max_index=918, n_tof=400_000, events_per_push=30.0, mz_lo=400.0, mz_hi=1000.0, frag_mz_lo=200.0,
        frag_mz_hi=1000.0, tof_mz_lo=195.0, tof_mz_hi=1010.0, planted_fraction=0.3, h_range=(3, 14), hs_range=(9, 19),
        candidates_on_window=True, sorted_noise=True, threads=THREADS,
    )
    assert case.dia.push_indices.size > 4e8 and case.dia.scan_max_index == 918
    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                    experimental_xic=True))
    cfgj = cfg.to_jitclass()
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    n = len(soa["precursor_idx"])
    ctx.stage_run(case.dia, force=True)
    cols = fragment_columns(case.library.fragment_df, "mz_library")
    ctx.stage_fragments(*cols, force=True)
    got = ctx.score_host(pack_assembled(soa), cfgj, with_stats=True)
    v = got["valid"].astype(bool)
    assert n == 600_000 and v.mean() > 0.5 and (got["features"][v][:, 29] != 0).mean() > 0.3
    exp = oracle_lib.score_timstof(case.dia, cols, pack_assembled(soa), cfgj, n_threads=THREADS, with_stats=True)
    compare(got, exp, PPM_ABS_TOL_ORACLE)  # all 600 000 candidates
    _bound_masked(compare.last_masked)
    assert np.array_equal(got["stat_matched_peaks"], exp["stat_matched_peaks"])
    got = {k: np.array(x, copy=True) for k, x in got.items()}
    for _ in range(3):  # run-to-run identical (a register hazard in an unrolled MFMA chain once broke this for feature 29)
        again = ctx.score_host(pack_assembled(soa), cfgj, with_stats=True)
        for k in got:
            assert np.array_equal(again[k], got[k], equal_nan=True), k
    perm = np.random.default_rng(2).permutation(n)
    shuffled = ctx.score_host(pack_assembled(_rows(soa, perm)), cfgj, with_stats=True)
    for k in got:
        assert np.array_equal(shuffled[k], got[k][perm], equal_nan=True), k


def test_config4_full_size_multiplex(ctx, oracle_lib):
    """configs[4] as specified: 75 000 elution groups x label channels {0, 4, 8, 12} = 300 000 precursors against
    the 2 h run, scored as MultiplexingRequantificationHandler scores them (class-default config + score_grouped,
    exclude_shared_ions, reference channel 0: multiplexing_requantification_handler.py:95-140): every score
    group against the oracle, and the four-way score-group split of the configuration (4 GPUs) scored shard by
    shard gives the rows of the unsharded table - no group is cut."""
    from alphadia_amd.distributed import shard_bounds, slice_soa
    from alphadia_amd.scoring import multiplex_candidates

    mc = syn.make_multiplex_case(75_000, 4800, threads=THREADS)
    cfg = CandidateScoringConfig()
    cfg.update(dict(score_grouped=True, exclude_shared_ions=True, reference_channel=0, experimental_xic=True))
    assert cfg.top_k_isotopes == 4 and not cfg.quant_all
    cfgj = cfg.to_jitclass()
    pdf = mc.library.precursor_df.sort_values("precursor_idx").reset_index(drop=True)
    multiplexed = multiplex_candidates(mc.psm_df, pdf, channels=list(mc.channels))
    multiplexed["rank"] = 0
    soa = assemble_candidates(multiplexed, pdf, "mz_library", score_grouped=True, reference_channel=0)
    n = len(soa["precursor_idx"])
    assert n == 300_000 and int(soa["score_group_idx"][-1]) + 1 == 75_000
    ctx.stage_run(mc.dia, force=True)
    cols = fragment_columns(mc.library.fragment_df, "mz_library")
    ctx.stage_fragments(*cols, force=True)
    got = ctx.score_host(pack_assembled(soa), cfgj, with_stats=True)
    got = {k: np.array(v, copy=True) for k, v in got.items()}
    v = got["valid"].astype(bool)
    assert 0.5 < v.mean() < 0.95
    idx = np.arange(n)  # every score group
    exp = oracle_lib.score(mc.dia, cols, pack_assembled(soa), cfgj, n_threads=THREADS, with_stats=True)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    _bound_masked(compare.last_masked)
    # (the diagnostic peak counter on the rows that were scored: the label shift moves some channel copies above the
    # last isolation window - no MS2 observation at all - where the reference still extracts MS1 before the candidate
    # fails at the presence mask (candidate.py:220-329) and the plan on the device drops the candidate at once)
    ev = exp["valid"].astype(bool)
    assert np.array_equal(got["stat_matched_peaks"][idx][ev], exp["stat_matched_peaks"][ev])
    pos = 0
    for rank in range(4):
        a, b = shard_bounds(soa["score_group_idx"], rank, 4)
        assert a == pos and b > a and (a == 0 or soa["score_group_idx"][a] != soa["score_group_idx"][a - 1])
        pos = b
        part = ctx.score_host(pack_assembled(slice_soa(soa, a, b)), cfgj, with_stats=True)
        for k in got:
            assert np.array_equal(part[k], got[k][a:b], equal_nan=True), (rank, k)
    assert pos == n
