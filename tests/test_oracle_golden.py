"""CPU: the oracle (oracle/adh_oracle.cpp) against golden vectors produced by running
the REFERENCE itself (tests/golden/make_golden.py) and against the reference's own
known-answer tests."""

import numpy as np
import pytest

import helpers as H

PPM_FEATURES = [8, 9, 41, 42, 45]
EXACT_FEATURES = [17, 20, 21, 28, 35, 37, 43]
INT_TABLES = ("fragment_precursor_idx fragment_rank fragment_position fragment_number "
              "fragment_type fragment_charge fragment_loss_type").split()


def _compare(got, exp, ppm_tol, rel_tol, corr_abs):
    assert np.array_equal(got["valid"].astype(bool), exp["valid"].astype(bool))
    assert np.array_equal(got["precursor_idx"], exp["precursor_idx"])
    assert np.array_equal(got["rank"], exp["rank"])
    v = exp["valid"].astype(bool)
    for name in INT_TABLES + ["fragment_mz_library", "fragment_mz"]:
        assert np.array_equal(got[name][v], exp[name][v]), name
    gf, ef = got["features"][v], exp["features"][v]
    assert np.array_equal(np.isnan(gf), np.isnan(ef))
    for f in EXACT_FEATURES:
        assert np.array_equal(gf[:, f], ef[:, f]), f
    for f in PPM_FEATURES:
        assert np.nanmax(np.abs(gf[:, f].astype(np.float64) - ef[:, f])) <= ppm_tol, f
    rest = [f for f in range(46) if f not in PPM_FEATURES]
    err = H.rel_err(gf[:, rest], ef[:, rest])
    # the absolute floor is for correlations only (differences of nearly equal sums): H.CORR_FEATURES
    floor = np.array([corr_abs if f in H.CORR_FEATURES else 0.0 for f in rest])
    err = np.where(np.abs(gf[:, rest].astype(np.float64) - ef[:, rest]) <= floor[None, :], 0.0, err)
    assert err.max() <= rel_tol, (err.max(), np.unravel_index(err.argmax(), err.shape))
    for name in ("fragment_mz_observed", "fragment_height", "fragment_intensity", "fragment_correlation"):
        e = H.rel_err(got[name][v], exp[name][v])
        if name == "fragment_correlation":
            e = np.where(np.abs(got[name][v].astype(np.float64) - exp[name][v]) <= corr_abs, 0.0, e)
        assert e.max() <= rel_tol, name
    d = np.abs(got["fragment_mass_error"][v].astype(np.float64) - exp["fragment_mass_error"][v])
    assert d.max() <= ppm_tol


@pytest.mark.parametrize("name", ["handler_default", "class_default", "topk6", "multiplex", "edges", "manyfrag", "manyfrag_class",
                                  "fitted_quadrupole"])
def test_oracle_numba_typing_vs_reference_goldens(oracle_lib, name):
    """Production (Numba) typing vs goldens captured under NumPy typing: validity, every
    integer table and the row order are exact; float features within 1e-4 relative except the
    documented shim artefacts (ppm errors: float32 MS1 collapse / weight normalisation;
    correlations: pairwise float32 sums), see tests/golden/ref_shim.py."""
    g = H.load_scoring_golden(name)
    got, soa = H.oracle_score(oracle_lib, g, g.config)
    assert np.array_equal(soa["precursor_idx"], g.z["order_precursor_idx"])
    assert np.array_equal(soa["rank"], g.z["order_rank"])
    _compare(got, g.expected, ppm_tol=0.15, rel_tol=1e-4, corr_abs=1e-3)


@pytest.mark.parametrize("name", ["handler_default", "class_default", "topk6", "multiplex", "edges", "manyfrag", "manyfrag_class",
                                  "fitted_quadrupole"])
def test_oracle_numpy_typing_pins_every_table(oracle_lib, name):
    """The goldens were produced by the reference running under NumPy (the shim), whose typing
    differs from Numba's at four places: the float32 MS1 collapse, the float32 normalisation of
    the fragment height weights and of the local weights in ``weighted_mean_a1``, and np.sum of a
    float32 row being pairwise (observation importance).  With these four switched to what the
    shim executed, the restatement reproduces the reference BIT FOR BIT on every m/z and mass
    error quantity (features 8, 9, 10, 41, 42, 45, fragment_mz_observed, fragment_mass_error,
    fragment_height; one and several observations) and to one float32 ulp on everything else
    (the remaining pairwise sums of the shim are not modelled): the restatement itself is pinned."""
    g = H.load_scoring_golden(name)
    oracle_lib.set_numpy_typing(True)
    try:
        got, _ = H.oracle_score(oracle_lib, g, g.config)
    finally:
        oracle_lib.set_numpy_typing(False)
    exp = g.expected
    assert np.array_equal(got["valid"].astype(bool), exp["valid"].astype(bool))
    v = exp["valid"].astype(bool)
    assert (exp["features"][v][:, 17] >= 2).sum() >= 5  # candidates seen through two isolation windows
    for f in (8, 9, 10, 41, 42, 45, 17, 20, 21, 28, 35, 37, 43):
        assert np.array_equal(got["features"][v][:, f], exp["features"][v][:, f], equal_nan=True), f
    for t in ("fragment_mz_observed", "fragment_mass_error", "fragment_height", "fragment_mz", "fragment_mz_library"):
        assert np.array_equal(got[t][v], exp[t][v]), t
    a, b = got["features"][v].astype(np.float64), exp["features"][v].astype(np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ad = np.abs(a - b)
    rel = ad / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)
    bad = np.where(np.isnan(a) | (ad <= 2e-6), 0.0, rel)
    assert bad.max() <= 1e-6, (bad.max(), np.unravel_index(bad.argmax(), bad.shape))
    for t in ("fragment_intensity", "fragment_correlation"):
        a, b = got[t][v].astype(np.float64), exp[t][v].astype(np.float64)
        ad = np.abs(a - b)
        rel = np.where(ad <= 2e-6, 0.0, ad / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30))
        assert rel.max() <= 1e-6, t


def test_get_dense_matches_reference(oracle_lib):
    z = np.load(H.golden_path("get_dense_alpharaw.npz"))
    dia = H.dia_from_npz(z)
    n_hits = 0
    for i in range(int(z["n_cases"])):
        fl, quad = z[f"q{i}_frame_limits"], z[f"q{i}_quad"]
        absolute = bool(z[f"q{i}_absolute"])
        dense, pidx = oracle_lib.get_dense(
            dia, fl[0, 0], fl[0, 1], z[f"q{i}_mz"], z[f"q{i}_tol"], quad[0, 0], quad[0, 1], absolute
        )
        e = z[f"q{i}_dense"]
        assert dense.shape == e.shape
        assert np.array_equal(pidx, z[f"q{i}_pidx"])
        assert np.array_equal(dense[0], e[0]), f"intensity channel, case {i}"
        if absolute:  # the channel the scoring path uses: bit exact
            assert np.array_equal(dense[1], e[1]), f"m/z channel, case {i}"
        else:  # ppm channel: Numba types `* 10**6` as float64, the shim ran it in float32
            assert np.allclose(dense[1], e[1], rtol=0, atol=2e-3)
        n_hits += int((e[0] > 0).sum())
    assert n_hits > 100


def test_fragcomp_matches_reference(oracle_lib):
    """The numpy preparation (``competition_plan``) + the oracle's competition loop reproduce the
    reference's surviving PSMs, in the reference's order (golden from FragmentCompetition.__call__)."""
    from alphadia_amd.fragcomp import competition_plan

    z = np.load(H.golden_path("fragcomp.npz"))
    psm = {k[4:]: z[k] for k in z.files if k.startswith("psm_")}
    frag = {k[5:]: z[k] for k in z.files if k.startswith("frag_")}
    plan = competition_plan(psm["precursor_idx"], psm["rank"], psm["mz_observed"], psm["proba"],
                            frag["precursor_idx"], frag["rank"], z["cycle"])
    assert np.all(np.diff(plan.window) >= 0) and plan.window_start[0] == 0 and plan.window_stop[-1] == len(plan.rows)
    valid = oracle_lib.fragcomp(plan.window_start, plan.window_stop, psm["rt_observed"][plan.rows],
                                plan.frag_start, plan.frag_stop, frag["mz_observed"], 3, 15)
    keep = plan.rows[valid]
    assert np.array_equal(psm["precursor_idx"][keep], z["surviving_precursor_idx"])
    assert np.array_equal(psm["rank"][keep], z["surviving_rank"])
    assert 0 < len(keep) < len(psm["precursor_idx"])


def test_competition_plan_edge_cases():
    """PSMs without fragment rows leave the competition; m/z outside every window falls into
    window 0; ties in proba are broken by precursor_idx, then by input position."""
    from alphadia_amd.fragcomp import competition_plan

    cycle = np.zeros((1, 3, 1, 2))
    cycle[0, :, 0, 0] = [-1, 400, 500]
    cycle[0, :, 0, 1] = [-1, 500, 600]
    pidx = np.array([7, 3, 3, 9, 5], dtype=np.uint32)
    rank = np.array([0, 0, 1, 0, 0], dtype=np.uint8)
    mz = np.array([450.0, 550.0, 550.0, 950.0, 420.0], dtype=np.float32)
    proba = np.array([0.2, 0.1, 0.1, 0.3, 0.2])
    fp = np.array([3, 3, 7, 7, 7, 3, 9], dtype=np.uint32)   # precursor 5 has no fragment rows
    fr = np.array([0, 0, 0, 0, 0, 1, 0], dtype=np.uint8)
    plan = competition_plan(pidx, rank, mz, proba, fp, fr, cycle)
    assert plan.rows.tolist() == [3, 0, 1, 2]               # window 0: 9 (outside); window 1: 7; window 2: 3/0, 3/1
    assert plan.window.tolist() == [0, 1, 2, 2]
    assert plan.frag_start.tolist() == [6, 2, 0, 5] and plan.frag_stop.tolist() == [7, 5, 2, 6]
    assert plan.window_start.tolist() == [0, 1, 2] and plan.window_stop.tolist() == [1, 2, 4]
    empty = competition_plan(pidx[:0], rank[:0], mz[:0], proba[:0], fp, fr, cycle)
    assert len(empty.rows) == 0 and len(empty.window_start) == 0


# ---- known-answer tests of the reference, restated against the oracle ----------------

@pytest.mark.parametrize(
    "x, expected",
    [  # tests/unit_tests/search/scoring/test_features.py:7-53
        ([1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1]),
        ([100, 10, 1, 1, 1, 10, 100], [1, 1, 1, 1, 1, 1, 1]),
        ([100, 0, 0, 1, 0, 0, 100], [0, 0, 0, 1, 0, 0, 0]),
        ([1, 1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1, 1]),
        ([100, 10, 1, 1, 1, 1, 10, 100], [1, 1, 1, 1, 1, 1, 1, 1]),
        ([100, 0, 0, 1, 1, 0, 0, 100], [0, 0, 0, 1, 1, 0, 0, 0]),
    ],
)
def test_center_envelope_kat(oracle_lib, x, expected):
    out = oracle_lib.center_envelope_1d(np.array([x], dtype=np.float32))
    np.testing.assert_array_almost_equal(out, np.array([expected], dtype=np.float32))


def test_center_envelope_rows_shrink_edges(oracle_lib):
    rng = np.random.default_rng(3)
    x = rng.random((10, 11)).astype(np.float32)
    out = oracle_lib.center_envelope_1d(x)
    assert np.all(out[:, 0] <= x[:, 0]) and np.all(out[:, -1] <= x[:, -1])


def test_fragment_correlation_kat(oracle_lib):
    # tests/unit_tests/search/scoring/test_scoring_utils.py:183-200
    a = np.array([[[1, 2, 3], [1, 2, 3]], [[3, 2, 1], [1, 2, 3]], [[0, 0, 0], [0, 0, 0]]])
    corr = oracle_lib.fragment_correlation(a)
    assert corr.shape == (2, 3, 3)
    expected = np.array(
        [[[1.0, -1.0, 0.0], [-1.0, 1.0, 0.0], [0.0, 0.0, 0.0]],
         [[1.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 0.0]]]
    )
    assert np.allclose(corr, expected)
    assert np.allclose(oracle_lib.fragment_correlation(np.zeros((10, 10, 10))), 0)


def test_save_corrcoeff_kat(oracle_lib):
    # tests/unit_tests/search/scoring/test_scoring_utils.py:158-180
    up = np.arange(1, 11, dtype=np.float32)
    assert np.isclose(oracle_lib.save_corrcoeff(up, up[::-1].copy()), -1.0)
    assert np.isclose(oracle_lib.save_corrcoeff(up, up), 1.0)
    assert np.isclose(oracle_lib.save_corrcoeff(np.zeros(10), np.zeros(10)), 0.0)


def test_search_sorted_left_kat(oracle_lib):
    # tests/unit_tests/search/jitclasses/test_alpharaw_jit.py:6-9
    assert oracle_lib.search_sorted_left(np.arange(100), 50) == 50


def test_quadrupole_transfer_function_kat(oracle_lib):
    # tests/unit_tests/search/scoring/test_quadrupole.py:59-79
    fake_cycle = np.array([[780.0, 801], [801, 820]])
    fake_cycle = np.repeat(fake_cycle[:, np.newaxis, :], 10, axis=1)[np.newaxis, :, :, :]
    isotope_mz = np.array([800.0, 800.1, 800.2, 802.42944, 802.9311, 803.1])
    qtf = oracle_lib.quadrupole_transfer_function(fake_cycle, np.array([0, 1]), np.arange(2, 9), isotope_mz)
    assert qtf.shape == (6, 2, 7)
    assert np.all(qtf[:3, 0, :] > 0.9) and np.all(qtf[:3, 1, :] < 0.1)
    assert np.all(qtf[3:, 0, :] < 0.1) and np.all(qtf[3:, 1, :] > 0.9)


def test_compete_for_fragments_kat(oracle_lib):
    # tests/unit_tests/fragcomp/test_fragcomp.py:38-57
    rt = np.array([10.0, 20.0, 20.0, 10.0, 10.0, 20])
    valid = oracle_lib.fragcomp(
        np.array([0, 3]), np.array([3, 6]), rt, np.array([0, 10, 20, 30, 40, 50]),
        np.array([10, 20, 30, 40, 50, 60]), np.tile(np.arange(100, 110), 6), 3, 15,
    )
    assert np.all(valid == np.array([True, True, False, True, False, True]))


def test_threads_do_not_change_results(oracle_lib):
    g = H.load_scoring_golden("handler_default")
    a, soa = H.oracle_score(oracle_lib, g, g.config, n_threads=1)
    b, _ = H.oracle_score(oracle_lib, g, g.config, n_threads=4, soa=soa)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


# ---- ion-mobility (timsTOF) layout ---------------------------------------------------

def _tims_golden():
    import pandas as pd

    import synthetic as syn
    from alphadia_amd.scoring import CandidateScoringConfig

    z = np.load(H.golden_path("scoring_timstof.npz"))
    dia = syn.TimsTOFArrays(
        cycle=z["tims_cycle"], dia_precursor_cycle=z["tims_dia_precursor_cycle"],
        rt_values=z["tims_rt_values"], mobility_values=z["tims_mobility_values"],
        mz_values=z["tims_mz_values"], tof_indptr=z["tims_tof_indptr"],
        push_indices=z["tims_push_indices"], intensity_values=z["tims_intensity_values"],
        scan_max_index=int(z["tims_scan_max_index"]), zeroth_frame=bool(z["tims_zeroth_frame"]),
    )
    fragment_df = pd.DataFrame({c: z["frag_" + c] for c in H.FRAG_COLS})
    precursor_df = pd.DataFrame({c: z["prec_" + c] for c in H.PREC_COLS})
    cand = pd.DataFrame({c: z["cand_" + c] for c in H.CAND_COLS})
    cfg = CandidateScoringConfig()
    cfg.update({k: z["cfg_" + k].item() for k in H.CFG_KEYS})
    return z, dia, fragment_df, precursor_df, cand, cfg


def test_timstof_get_dense_matches_reference(oracle_lib):
    z, dia, *_ = _tims_golden()
    hits = 0
    for i in range(int(z["n_cases"])):
        fl, sl, quad = z[f"q{i}_frame_limits"], z[f"q{i}_scan_limits"], z[f"q{i}_quad"]
        dense, pidx = oracle_lib.get_dense_timstof(
            dia, fl[0, 0], fl[0, 1], sl[0, 0], sl[0, 1], z[f"q{i}_mz"], z[f"q{i}_tol"], quad[0, 0], quad[0, 1]
        )
        e = z[f"q{i}_dense"]
        if e.size == 0:  # no push matches the quadrupole: bruker_jit.py:363-366
            assert dense.size == 0
            continue
        assert dense.shape == e.shape and np.array_equal(pidx, z[f"q{i}_pidx"])
        assert np.array_equal(dense, e), f"case {i}"
        hits += int((e[0] > 0).sum())
    assert hits > 20


def test_oracle_numpy_typing_pins_timstof(oracle_lib):
    """The NumPy-typing pin of test_oracle_numpy_typing_pins_every_table for the ion-mobility layout
    (scoring_timstof.npz).  One more typing site shows here: the outer sum of the observation importance,
    np.sum(np.sum(template, axis=-1), axis=-1), runs over the scan axis - two slots on the AlphaRaw layout,
    18 and more here, where NumPy sums pairwise.  With it the restatement reproduces the reference bit for
    bit on every m/z and mass-error quantity, on heights and areas, and to one float32 ulp elsewhere."""
    from alphadia_amd.scoring import assemble_candidates, fragment_columns, pack_assembled

    z, dia, fragment_df, precursor_df, cand, cfg = _tims_golden()
    soa = assemble_candidates(cand, precursor_df, "mz_library")
    oracle_lib.set_numpy_typing(True)
    try:
        got = oracle_lib.score_timstof(
            dia, fragment_columns(fragment_df, "mz_library"), pack_assembled(soa), cfg.to_jitclass()
        )
    finally:
        oracle_lib.set_numpy_typing(False)
    exp = {n: z["out_" + n] for n in H.OUT_NAMES}
    assert np.array_equal(got["valid"].astype(bool), exp["valid"].astype(bool))
    v = exp["valid"].astype(bool)
    assert (exp["features"][v][:, 17] >= 2).sum() >= 5  # candidates seen through two isolation windows
    for f in (8, 9, 10, 41, 42, 45, 17, 20, 21, 28, 35, 37, 43):
        assert np.array_equal(got["features"][v][:, f], exp["features"][v][:, f], equal_nan=True), f
    for t in ("fragment_mz_observed", "fragment_mass_error", "fragment_height", "fragment_intensity", "fragment_mz",
              "fragment_mz_library"):
        assert np.array_equal(got[t][v], exp[t][v]), t
    a, b = got["features"][v].astype(np.float64), exp["features"][v].astype(np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ad = np.abs(a - b)
    rel = np.where(np.isnan(a) | (ad <= 2e-6), 0.0, ad / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30))
    assert rel.max() <= 1e-6, (rel.max(), np.unravel_index(rel.argmax(), rel.shape))
    a, b = got["fragment_correlation"][v].astype(np.float64), exp["fragment_correlation"][v].astype(np.float64)
    assert np.abs(a - b).max() <= 2e-6


def test_timstof_scoring_matches_reference(oracle_lib):
    """Ion-mobility path incl. fragment/template scan correlation (features 29, 30) and
    mobility FWHM (39): same bar as for the AlphaRaw layout."""
    from alphadia_amd.scoring import assemble_candidates, fragment_columns, pack_assembled

    z, dia, fragment_df, precursor_df, cand, cfg = _tims_golden()
    soa = assemble_candidates(cand, precursor_df, "mz_library")
    got = oracle_lib.score_timstof(
        dia, fragment_columns(fragment_df, "mz_library"), pack_assembled(soa), cfg.to_jitclass()
    )
    exp = {n: z["out_" + n] for n in H.OUT_NAMES}
    _compare(got, exp, ppm_tol=0.15, rel_tol=1e-4, corr_abs=1e-3)
    v = exp["valid"].astype(bool)
    assert v.sum() > 100 and (exp["features"][v][:, 29] != 0).sum() > 50
