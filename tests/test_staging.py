"""f-4: spectrum / peak tables -> run arrays, against outputs of the reference's own
``determine_dia_cycle`` and ``AlphaRaw._preprocess_raw_data`` (tests/golden/staging.npz)."""

import numpy as np
import pytest

import helpers as H
from alphadia_amd.staging import (
    NotValidDiaDataError,
    cycle_is_consistent,
    cycle_length,
    cycle_start,
    determine_dia_cycle,
    preprocess_spectra,
)

CASES = ["plain", "prefix", "irregular_ms1"]


def _tables(z, name):
    spec = {k[len(name) + 6:]: z[k] for k in z.files if k.startswith(f"{name}_spec_")}
    peaks = {k[len(name) + 6:]: z[k] for k in z.files if k.startswith(f"{name}_peak_")}
    return spec, peaks


@pytest.mark.parametrize("name", CASES)
def test_preprocess_matches_reference(name):
    z = np.load(H.golden_path("staging.npz"))
    spec, peaks = _tables(z, name)
    got = preprocess_spectra(spec, peaks)
    assert got.has_ms1 == bool(z[f"{name}_has_ms1"])
    assert got.cycle_start == int(z[f"{name}_cycle_start"]) and got.cycle_len == int(z[f"{name}_cycle_length"])
    assert np.array_equal(got.cycle, z[f"{name}_cycle"]) and got.cycle.dtype == np.float64
    for mine, ref in (("rt_values", "rt_values"), ("peak_start_idx_list", "peak_start_idx_list"),
                      ("peak_stop_idx_list", "peak_stop_idx_list"), ("mz_values", "mz_values"),
                      ("intensity_values", "intensity_values")):
        a, b = getattr(got, mine), z[f"{name}_{ref}"]
        assert a.dtype == b.dtype and np.array_equal(a, b), mine
    assert got.precursor_cycle_max_index == int(z[f"{name}_precursor_cycle_max_index"])
    assert got.frame_max_index == int(z[f"{name}_frame_max_index"])
    for mine in ("max_mz_value", "min_mz_value", "quad_max_mz_value", "quad_min_mz_value"):
        assert getattr(got, mine) == z[f"{name}_{mine}"] and getattr(got, mine).dtype == np.float32, mine
    if name == "prefix":
        assert got.cycle_start == 7
    if name == "irregular_ms1":  # the unusable MS1 spectra are gone: no (-1, -1) row in the cycle
        assert not (got.cycle[0, :, 0, 0] == -1).any()


def test_cycle_detection_pieces():
    sig = np.tile(np.array([-2.0, 810.0, 830.0, 850.0]), 30)
    assert cycle_length(sig) == 4 and cycle_start(sig, 4) == 0 and cycle_is_consistent(sig, 4, 0)
    padded = np.concatenate([np.full(6, 777.0), sig])  # a constant settling stretch is skipped
    assert cycle_length(padded) == 4
    start = cycle_start(padded, 4)
    assert start >= 3 and cycle_is_consistent(padded, 4, start)
    broken = sig.copy()
    broken[61] += 5.0
    assert not cycle_is_consistent(broken, 4, 0)
    with pytest.raises(NotValidDiaDataError, match="not consistent"):
        determine_dia_cycle(broken / 2, broken / 2)
    with pytest.raises(NotValidDiaDataError, match="length"):
        determine_dia_cycle(np.arange(50.0), np.arange(50.0))  # a ramp has no autocorrelation peak
    with pytest.raises(NotValidDiaDataError, match="start"):
        determine_dia_cycle(np.r_[np.zeros(40), 1.0, np.zeros(40)], np.zeros(81))


@pytest.mark.gpu
def test_staged_tables_score_like_the_arrays():
    """``stage_spectra`` puts the run in HBM: scoring through it equals scoring the hand-built arrays."""
    from alphadia_amd import runtime
    from alphadia_amd.scoring import fragment_columns, pack_assembled
    from alphadia_amd.staging import stage_spectra

    g = H.load_scoring_golden("handler_default")
    d = g.dia
    L = d.cycle.shape[1]
    n = d.rt_values.shape[0]
    cyc = d.cycle[0, :, 0, :]
    spec = dict(spec_idx=np.arange(n), rt=d.rt_values.astype(np.float64) / 60.0,
                ms_level=np.where(np.tile(cyc[:, 0], n // L + 1)[:n] == -1, 1, 2),
                precursor_mz=np.tile(cyc.mean(axis=1), n // L + 1)[:n],
                isolation_lower_mz=np.tile(cyc[:, 0], n // L + 1)[:n], isolation_upper_mz=np.tile(cyc[:, 1], n // L + 1)[:n],
                peak_start_idx=d.peak_start_idx_list, peak_stop_idx=d.peak_stop_idx_list)
    arrays = stage_spectra(spec, dict(mz=d.mz_values, intensity=d.intensity_values), device=0)
    assert arrays.cycle_start == 0 and np.array_equal(arrays.cycle, d.cycle)
    ctx = runtime.get_context(0)
    ctx.stage_fragments(*fragment_columns(g.library.fragment_df, "mz_library"), force=True)
    soa = H.soa_for(g, g.config)
    got = ctx.score_host(pack_assembled(soa), g.config.to_jitclass())
    ctx.stage_run(g.dia, force=True)
    ref = ctx.score_host(pack_assembled(soa), g.config.to_jitclass())
    v = ref["valid"].astype(bool)
    assert np.array_equal(got["valid"], ref["valid"]) and v.sum() > 50
    # rt went through minutes and back: features that read rt may move by one float32 ulp of rt
    assert H.rel_err(got["features"][v], ref["features"][v]).max() < 1e-5
    assert np.array_equal(got["fragment_mz_observed"], ref["fragment_mz_observed"])
