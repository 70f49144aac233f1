"""Runs with more than 2^32 detector events (real diaPASEF runs have them; the reference's arrays are int64
throughout: bruker_jit.py:22-55).

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
