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
