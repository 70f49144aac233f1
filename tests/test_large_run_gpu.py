"""Runs with more than 2^32 detector events / peaks (real diaPASEF and long Astral runs have them; the reference's
arrays are int64 throughout: bruker_jit.py:22-55, alpharaw_jit.py:78-97).

The ion-mobility kernels carry event numbers in 64 bits; what they keep in 32 bits counts from the first event
of a TOF bin (index columns) or of a window's first bin (pair ranges).  The golden run is pushed behind two
dummy TOF bins of 2^31 + 1000 events each (below every m/z window: nothing reads their events), so that every
real event number exceeds 2^32: scoring and candidate selection must give what they give on the run itself -
with the search indices and without.
"""

import numpy as np
import pytest

import helpers as H
import synthetic as syn
from alphadia_amd.scoring import assemble_candidates, fragment_columns, pack_assembled
from test_gpu_parity import _tims_case_from_golden

pytestmark = pytest.mark.gpu

DUMMY = (1 << 31) + 1000  # events per dummy bin


def _shifted(dia):
    n = int(dia.push_indices.size)
    try:
        push = np.zeros(2 * DUMMY + n, np.uint32)   # (untouched pages of np.zeros cost nothing on the host)
        inten = np.zeros(2 * DUMMY + n, np.uint16)
    except MemoryError:
        pytest.skip("not enough host address space for a 2^32-event run")
    push[2 * DUMMY:] = dia.push_indices
    inten[2 * DUMMY:] = dia.intensity_values
    indptr = np.concatenate([[0, DUMMY], 2 * DUMMY + np.asarray(dia.tof_indptr, np.int64)])
    mz = np.concatenate([[1.0, 2.0], dia.mz_values])
    return syn.TimsTOFArrays(cycle=dia.cycle, dia_precursor_cycle=dia.dia_precursor_cycle, rt_values=dia.rt_values,
                             mobility_values=dia.mobility_values, mz_values=mz, tof_indptr=indptr, push_indices=push,
                             intensity_values=inten, scan_max_index=dia.scan_max_index, zeroth_frame=dia.zeroth_frame)


@pytest.mark.parametrize("index", ["1", "0"])
def test_event_numbers_beyond_32_bits(monkeypatch, index):
    from alphadia_amd import runtime

    free, _ = runtime.device_memory() if hasattr(runtime, "device_memory") else (1 << 40, 0)
    if free < 40 << 30:
        pytest.skip("needs 30 GB of device memory")
    monkeypatch.setenv("ADH_IM_INDEX", index)
    ctx = runtime.get_context(0)
    z, dia, fragment_df, precursor_df, cand, cfg = _tims_case_from_golden()
    soa = assemble_candidates(cand, precursor_df, "mz_library")
    cols = fragment_columns(fragment_df, "mz_library")
    ctx.stage_run(dia, force=True)
    ctx.stage_fragments(*cols, force=True)
    ref = ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=True)
    big = _shifted(dia)
    assert big.tof_indptr[3] > 1 << 32 and big.push_indices.size > 1 << 32
    ctx.stage_run(big, force=True)
    got = ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=True)
    assert ref["valid"].sum() > 100
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True), k
    ctx.stage_run(dia, force=True)  # (release the 26 GB)


def test_transposition_of_more_than_two_to_the_31_events():
    """`adh_transpose_timstof` on 2^31 + 5e6 detector events (13 GB in, 13 GB out): the reference's `_transpose`
    (alphadia/raw_data/bruker.py:201-274) has no size limit, round 3's transposer stopped at 2^31 - 1.  Checked
    by properties - a NumPy stable sort of 2e9 keys is out of reach of a test: tof_indptr equals the bin counts,
    pushes ascend inside every bin, the (push, value) pairs of a sample of bins are exactly the input's, and the
    sums of all pushes and values are preserved."""
    from alphadia_amd import runtime

    ctx = runtime.get_context(0)
    rng = np.random.default_rng(13)
    n_tof, per_push = 200_000, 2_000
    n = (1 << 31) + 5_000_000
    n_push = (n + per_push - 1) // per_push
    ptr = np.minimum(np.arange(n_push + 1, dtype=np.int64) * per_push, n)
    tof = rng.integers(0, n_tof, n, dtype=np.uint32)
    val = rng.integers(1, 60000, n, dtype=np.uint16)
    push, indptr, out_val = ctx.transpose_timstof(tof, ptr, n_tof, val)
    assert indptr[0] == 0 and indptr[-1] == n
    counts = np.zeros(n_tof, dtype=np.int64)
    for a in range(0, n, 1 << 28):  # (bincount in slices: the int64 copy of 2e9 indices would be 17 GB)
        counts += np.bincount(tof[a:a + (1 << 28)], minlength=n_tof)
    assert np.array_equal(np.diff(indptr), counts)
    assert int(out_val.astype(np.uint64).sum()) == int(val.astype(np.uint64).sum())
    push_of_sum = int((np.arange(n_push, dtype=np.uint64) * np.diff(ptr).astype(np.uint64)).sum())
    assert int(push.astype(np.uint64).sum()) == push_of_sum
    # ascending pushes inside every bin: a descent may only happen where a new bin starts
    desc = np.flatnonzero(push[1:] < push[:-1]) + 1
    assert np.isin(desc, indptr).all()
    for t in (0, 1, 77_777, n_tof - 1):
        where = np.flatnonzero(tof == t)
        a, b = int(indptr[t]), int(indptr[t + 1])
        assert np.array_equal(push[a:b], (where // per_push).astype(np.uint32))
        assert np.array_equal(out_val[a:b], val[where])


def test_alpharaw_run_of_more_than_two_to_the_32_peaks(monkeypatch):
    """An AlphaRaw run of 4.5e9 peaks (36 GB): the golden run pushed behind 192 cycles that hold three dummy
    MS1 spectra of 1.5e9 peaks each (one per group of cycle blocks; below every m/z window of a candidate in
    time, so nothing reads them), so that every real entry number exceeds 2^32.  The bin table counts from the
    first entry of every group of blocks and the run is sorted slab by slab (round 4; round 3 refused such a run):
    the fused kernel, the two-kernel path and candidate selection must give what they give on the run itself."""
    import pandas as pd

    from alphadia_amd import runtime
    from alphadia_amd.selection import CandidateSelectionConfig, HipCandidateSelection
    ctx = runtime.get_context(0)
    from alphadia_amd.scoring import CandidateScoringConfig

    g = syn.make_case(600, 400, config_id=61, per_precursor=2, planted_fraction=0.5, threads=8)
    g.config = CandidateScoringConfig()
    g.config.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                         experimental_xic=True))
    names = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
                 fragment_mz_column="mz_library")
    scfg = CandidateSelectionConfig()
    scfg.update(dict(rt_tolerance=15.0, candidate_count=2))

    cols = fragment_columns(g.library.fragment_df, "mz_library")

    def everything(case_like, soa):
        ctx.stage_run(case_like.dia, force=True)  # (once: 36 GB of peaks go through it for the big run)
        ctx.stage_fragments(*cols, force=True)
        fused = ctx.score_host(pack_assembled(soa), g.config.to_jitclass(), with_stats=True)
        fused = {k: np.array(v, copy=True) for k, v in fused.items()}
        with monkeypatch.context() as mp:
            mp.setenv("ADH_DEBUG_NO_FUSED", "1")
            two = ctx.score_host(pack_assembled(soa), g.config.to_jitclass(), with_stats=True)
            two = {k: np.array(v, copy=True) for k, v in two.items()}
        sel = HipCandidateSelection(case_like.dia, case_like.library.precursor_df, case_like.library.fragment_df, scfg,
                                    device=0, **names)()
        return fused, two, sel

    monkeypatch.setenv("ADH_BLOCK_CYCLES", "8")  # blocks of 8 cycles, groups of 64
    soa0 = H.soa_for(g, g.config)
    ref = everything(g, soa0)
    dia = g.dia
    L = dia.cycle_len
    pad_cycles, dummy = 192, 1_500_000_000
    n0 = int(dia.mz_values.size)
    try:
        mz = np.empty(3 * dummy + n0, np.float32)
        inten = np.empty(3 * dummy + n0, np.float32)
    except MemoryError:
        pytest.skip("not enough host memory for a 2^32-peak run")
    mz[: 3 * dummy] = np.float32(dia.mz_values.min())
    inten[: 3 * dummy] = 1.0
    mz[3 * dummy:] = dia.mz_values
    inten[3 * dummy:] = dia.intensity_values
    n_pad = pad_cycles * L
    counts = np.zeros(n_pad, np.int64)
    counts[[0, 64 * L, 128 * L]] = dummy  # the MS1 spectrum of cycles 0, 64, 128: one dummy spectrum per group of blocks
    pad_stop = np.cumsum(counts)
    start = np.concatenate([pad_stop - counts, 3 * dummy + np.asarray(dia.peak_start_idx_list, np.int64)])
    stop = np.concatenate([pad_stop, 3 * dummy + np.asarray(dia.peak_stop_idx_list, np.int64)])
    # the time axis runs on backwards with the run's own spacing (the smoothing kernel of the selection is sized
    # from the mean cycle time)
    dt = np.float64(dia.rt_values[L] - dia.rt_values[0]) / L
    rt = np.concatenate([(dia.rt_values[0] - dt * np.arange(n_pad, 0, -1)).astype(np.float32), dia.rt_values])
    big = syn.AlphaRawArrays(cycle=dia.cycle, rt_values=rt, peak_start_idx_list=start, peak_stop_idx_list=stop,
                             mz_values=mz, intensity_values=inten)
    assert big.mz_values.size > 1 << 32 and start[n_pad] > 1 << 32
    from types import SimpleNamespace

    shifted = dict(soa0)
    for c in ("frame_start", "frame_stop", "frame_center"):
        shifted[c] = soa0[c] + n_pad
    case_big = SimpleNamespace(dia=big, library=g.library, candidates_df=g.candidates_df)
    got = everything(case_big, shifted)
    assert ref[0]["valid"].sum() > 100 and len(ref[2]) > 100
    for a, b in ((ref[0], got[0]), (ref[1], got[1])):
        for k in a:
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    sel_ref, sel_big = ref[2], got[2].copy()
    for c in ("frame_start", "frame_stop", "frame_center"):
        sel_big[c] = sel_big[c] - n_pad
    # (a precursor whose search window reaches the first cycles of the run sees a run that starts earlier: the
    # others - library RT a minute or more into the run - must get the same boxes)
    late = g.library.precursor_df.loc[g.library.precursor_df["rt_library"].values > 60.0, "precursor_idx"].values
    inner = lambda df: df[df["precursor_idx"].isin(late)].reset_index(drop=True)  # noqa: E731
    pd.testing.assert_frame_equal(inner(sel_ref), inner(sel_big))
    assert len(inner(sel_ref)) > 300
    ctx.stage_run(g.dia, force=True)  # (release the 36 GB)
