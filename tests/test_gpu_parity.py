"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle and vs the
golden vectors captured from the reference.  Run with ``-m gpu`` on an MI355X."""

import os

import numpy as np
import pytest

import helpers as H
import synthetic as syn
from alphadia_amd.scoring import FRAGMENT_DF_COLUMNS, CandidateScoringConfig, fragment_columns, pack_assembled

pytestmark = pytest.mark.gpu

# feature classes (indices into OutputPsmDF.features, scoring.py:34-81)
PPM_FEATURES = [8, 9, 41, 42, 45]          # mass errors in ppm: differences of nearly equal m/z
EXACT_FEATURES = [17, 20, 21, 28, 35, 37, 43]  # counts and ratios of counts
REL_TOL = 1e-4                             # BASELINE.json north_star: FP features within 1e-4 relative
PPM_ABS_TOL_ORACLE = 1e-5                  # HIP vs oracle (same typing, same order): 9.5e-7 ppm observed over 3 M rows
PPM_ABS_TOL_GOLDEN = 0.15                  # vs shim goldens: float32 weight normalisation, see ref_shim.py


@pytest.fixture(scope="module")
def ctx():
    from alphadia_amd import runtime

    return runtime.get_context(0)


def hip_score(ctx, case_like, cfg, soa=None, with_stats=False):
    soa = soa if soa is not None else H.soa_for(case_like, cfg)
    ctx.stage_run(case_like.dia, force=True)
    ctx.stage_fragments(*fragment_columns(case_like.library.fragment_df, "mz_library"), force=True)
    return ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=with_stats), soa


def _log_masked(masked: dict) -> None:
    """One line per comparison in gpurun_out/parity_masks.jsonl (the GPU run pulls that directory back): which test,
    how many valid rows, how many of them each knife-edge mask left out of its feature."""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_masks.jsonl"), "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0],
                                "rows": masked["rows"], "eligible_16": masked[16], "eligible_18": masked[18],
                                "eligible_19": masked[19], "differing_16": masked["rescued"][16],
                                "differing_18": masked["rescued"][18], "differing_19": masked["rescued"][19]}) + "\n")
    except OSError:
        pass


def compare(got, exp, ppm_tol, rel_tol=REL_TOL, corr_abs=0.0):
    """Integer tables exact; float tables within tolerance.  Returns the worst relative error."""
    gv, ev = got["valid"].astype(bool), exp["valid"].astype(bool)
    assert np.array_equal(gv, ev), f"valid differs on {np.flatnonzero(gv != ev)[:10]}"
    assert np.array_equal(got["precursor_idx"], exp["precursor_idx"])
    assert np.array_equal(got["rank"], exp["rank"])
    v = ev
    for name in ("fragment_precursor_idx fragment_rank fragment_position fragment_number "
                 "fragment_type fragment_charge fragment_loss_type").split():
        assert np.array_equal(got[name][v], exp[name][v]), name
    for name in ("fragment_mz_library", "fragment_mz"):
        assert np.array_equal(got[name][v], exp[name][v]), name
    if "fragment_lib_slot" in got and "fragment_lib_slot" in exp:  # (the reference goldens have no such column)
        assert np.array_equal(got["fragment_lib_slot"][v], exp["fragment_lib_slot"][v]), "fragment_lib_slot"
    gf, ef = got["features"][v].copy(), exp["features"][v].copy()
    # np.corrcoef of a constant vector is 0/0 (fragment_features.py:340-356, features 18 and 19).  Whether
    # two equal float32 heights / areas are also equal in float64 hangs on the last bit of a float64
    # exp() (device libm vs glibc): rows whose filled slots all show the same value are knife-edge and
    # excluded from the comparison of that feature.
    filled = exp["fragment_type"][v] != 0
    masked, rescued = {}, {}

    def _disagree(a, b):  # rows a comparison would have failed on: NaN on one side only, or beyond the tolerance
        a, b = a.astype(np.float64), b.astype(np.float64)
        with np.errstate(invalid="ignore"):
            far = np.abs(a - b) > rel_tol * np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)
        return (np.isnan(a) != np.isnan(b)) | (far & ~np.isnan(a) & ~np.isnan(b))

    for f, table in ((18, "fragment_intensity"), (19, "fragment_height")):
        x = np.where(filled, exp[table][v], np.nan)
        flat = (np.nanmax(x, axis=1) == np.nanmin(x, axis=1)) if x.shape[1] else np.zeros(len(x), bool)
        rescued[f] = int((flat & _disagree(gf[:, f], ef[:, f])).sum())
        gf[flat, f] = 0.0
        ef[flat, f] = 0.0
        masked[f] = int(flat.sum())
    # feature 16 (correlation of the isotope intensities with the isotope heights, save_corrcoeff) is the
    # same kind of knife edge: isotope planes of equal height (e.g. one event of 5 counts each) give
    # numerator / (denominator + 1e-12) with both ~1e-16 or exactly 0, depending on the last bit of exp()
    knife = ((gf[:, 16] == 0.0) | (ef[:, 16] == 0.0)) & (np.abs(gf[:, 16]) < 1e-3) & (np.abs(ef[:, 16]) < 1e-3)
    rescued[16] = int((knife & _disagree(gf[:, 16], ef[:, 16])).sum())
    gf[knife, 16] = 0.0
    ef[knife, 16] = 0.0
    masked[16] = int(knife.sum())
    # `masked`: rows ELIGIBLE for a mask (mostly candidates without signal: every filled slot 0, both sides NaN or 0
    # alike - the mask changes nothing there); `rescued`: rows the mask actually took out of a comparison that would
    # have failed.  compare.last_masked keeps both, gpurun_out/parity_masks.jsonl logs both, the
    # full-size tests bound the second.
    compare.last_masked = dict(masked, rows=int(v.sum()), rescued=rescued)
    _log_masked(compare.last_masked)
    print(f"[compare] {int(v.sum())} valid rows; knife-edge rows eligible / actually differing: feature 16: {masked[16]} / "
          f"{rescued[16]}, 18: {masked[18]} / {rescued[18]}, 19: {masked[19]} / {rescued[19]}")
    assert np.array_equal(np.isnan(gf), np.isnan(ef)), "NaN pattern differs"
    for f in EXACT_FEATURES:
        assert np.array_equal(gf[:, f], ef[:, f]), f"feature {f} must be exact"
    for f in PPM_FEATURES:
        d = np.nanmax(np.abs(gf[:, f].astype(np.float64) - ef[:, f]))
        assert d <= ppm_tol, f"ppm feature {f}: abs diff {d}"
    rest = [f for f in range(46) if f not in PPM_FEATURES]
    err = H.rel_err(gf[:, rest], ef[:, rest])
    if corr_abs > 0:
        # an absolute floor only where a relative bound means nothing near zero: the correlation features
        # (H.CORR_FEATURES) and the fragment_correlation table
        floor = np.array([corr_abs if f in H.CORR_FEATURES else 0.0 for f in rest])
        ad = np.abs(gf[:, rest].astype(np.float64) - ef[:, rest])
        err = np.where(ad <= floor[None, :], 0.0, err)
    worst = float(err.max()) if err.size else 0.0
    assert worst <= rel_tol, f"feature rel err {worst} at {np.unravel_index(err.argmax(), err.shape)}"
    for name in ("fragment_mz_observed", "fragment_height", "fragment_intensity", "fragment_correlation"):
        e = H.rel_err(got[name][v], exp[name][v])
        if corr_abs > 0 and name == "fragment_correlation":
            e = np.where(np.abs(got[name][v].astype(np.float64) - exp[name][v]) <= corr_abs, 0.0, e)
        assert e.max() <= rel_tol, f"{name}: {e.max()}"
    d = np.abs(got["fragment_mass_error"][v].astype(np.float64) - exp["fragment_mass_error"][v]).max()
    assert d <= ppm_tol, f"fragment_mass_error abs diff {d}"
    return worst


@pytest.mark.parametrize("name", ["handler_default", "class_default", "topk6", "multiplex", "edges", "manyfrag", "manyfrag_class",
                                  "fitted_quadrupole"])
def test_hip_matches_oracle_on_golden_inputs(ctx, oracle_lib, name):
    g = H.load_scoring_golden(name)
    got, soa = hip_score(ctx, g, g.config)
    exp, _ = H.oracle_score(oracle_lib, g, g.config, soa=soa)
    compare(got, exp, PPM_ABS_TOL_ORACLE)


@pytest.mark.parametrize("name", ["handler_default", "topk6", "multiplex", "edges"])
def test_library_columns_rebuilt_from_slots(ctx, name):
    """The library columns of the fragment tables stay out of the all-gather: fragment_lib_slot and
    the staged library give them back bit for bit (all three feature kernels write the slot)."""
    from alphadia_amd.distributed import LIBRARY_COLUMNS, rebuild_local_columns

    g = H.load_scoring_golden(name)
    got, soa = hip_score(ctx, g, g.config)
    wire = {k: v for k, v in got.items() if k not in LIBRARY_COLUMNS}
    cols = fragment_columns(g.library.fragment_df, "mz_library")
    back = rebuild_local_columns(wire, soa["precursor_idx"], soa["rank"], soa["flags"],
                                 frag_start=soa["frag_start_idx"], fragment_cols=cols)
    assert (got["fragment_lib_slot"] != 0).sum() == (got["fragment_type"] != 0).sum() > 0
    for k in LIBRARY_COLUMNS + ("precursor_idx", "rank", "fragment_precursor_idx", "fragment_rank"):
        assert np.array_equal(back[k], got[k]), k


@pytest.mark.parametrize("name", ["handler_default", "class_default", "topk6", "multiplex", "edges", "manyfrag", "manyfrag_class",
                                  "fitted_quadrupole"])
def test_hip_matches_reference_goldens(ctx, name):
    g = H.load_scoring_golden(name)
    got, _ = hip_score(ctx, g, g.config)
    # north_star tolerance (1e-4 relative).  Shim caveats: the goldens ran under NumPy typing, which
    # moves the ppm errors by up to one float32 ulp of m/z (0.15 ppm; pinned exactly by
    # test_oracle_numpy_typing_pins_every_table) and cancellation-prone correlations by <= 1e-3 absolute
    compare(got, g.expected, PPM_ABS_TOL_GOLDEN, rel_tol=REL_TOL, corr_abs=1e-3)


@pytest.mark.parametrize("name", ["handler_default", "multiplex", "edges", "topk6", "fitted_quadrupole"])
@pytest.mark.parametrize("quant_all", [False, True])
def test_requantification_configs_take_the_fused_kernel(ctx, oracle_lib, monkeypatch, name, quant_all):
    """The scoring config of multiplex / transfer-library requantification is ``CandidateScoringConfig()`` with
    its class defaults - top_k_isotopes = 4, quant_all = False (config.py:70-84,
    multiplexing_requantification_handler.py:121-125) - plus experimental_xic from the search config.  Since
    round 4 the fused kernel scores it (isotope 3 on lane 15; with two observations the more important one is
    quantified and keeps its envelope edit, fragment_features.py:240-250): equal to the oracle, bit for bit
    equal to the two-kernel path, and no candidate of these tables goes through the gather kernel."""
    g = H.load_scoring_golden(name)
    cfg = CandidateScoringConfig()
    cfg.update(dict(score_grouped=bool(g.config.score_grouped), reference_channel=int(g.config.reference_channel),
                    exclude_shared_ions=True, experimental_xic=True, quant_all=quant_all))
    cfg.quadrupole_sigma, cfg.quadrupole_delta_mu = g.config.quadrupole_sigma, g.config.quadrupole_delta_mu
    assert cfg.top_k_isotopes == 4 and cfg.top_k_fragments == 12
    ctx.kernel_time_ms(reset=True)
    got, soa = hip_score(ctx, g, cfg)
    gather_fused, feature_fused, _ = ctx.kernel_time_ms(reset=True)
    exp, _ = H.oracle_score(oracle_lib, g, cfg, soa=soa)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    assert exp["valid"].sum() > (5 if name == "edges" else 20)
    with monkeypatch.context() as mp:
        mp.setenv("ADH_DEBUG_NO_FUSED", "1")
        two_kernel, _ = hip_score(ctx, g, cfg, soa=soa)
        gather_two, _, _ = ctx.kernel_time_ms(reset=True)
    for k in got:
        assert np.array_equal(two_kernel[k], got[k], equal_nan=True), k
    # (no gather launch between the two events of the fused run: microseconds against a kernel; "edges" holds
    # shapes outside the fused kernel's on purpose)
    if name != "edges":
        assert gather_fused < 0.25 * gather_two and feature_fused > 0, (gather_fused, gather_two)


@pytest.mark.parametrize("seed,k_fragments,quant_all,n_iso", [(11, (17, 32), True, 3), (12, (33, 61), True, 3), (13, (14, 61), False, 3),
                                                              (14, (17, 45), True, 4), (15, (20, 50), False, 2), (16, (17, 40), True, 1)])
def test_wide_register_kernels_equal_the_generic_kernel(ctx, oracle_lib, monkeypatch, seed, k_fragments, quant_all, n_iso):
    """Transfer-library requantification scores with ``top_k_fragments = 9999`` against libraries that carry every
    predicted fragment (transfer_library_requantification_handler.py:117-124): candidates keep 17 ... 64 fragments.
    The wide forms of the register kernel (adh_feature_fast_kernel<FM, NO, 32 / 64>, one launch per observation count)
    take them since round 5: equal to the oracle, bit for bit equal to the generic LDS kernel they replace
    (ADH_DEBUG_NO_WIDE), and several times faster."""
    case = syn.make_case(1500, 400, config_id=2, per_precursor=2, threads=8, seed=seed, k_fragments=k_fragments)
    cfg = CandidateScoringConfig()
    # (n_iso: the precursor phase of the register kernels walks 4 x n_iso sequential sums on as many lanes: round 6)
    cfg.update(dict(score_grouped=False, top_k_isotopes=n_iso, reference_channel=-1, precursor_mz_tolerance=10,
                    fragment_mz_tolerance=15, exclude_shared_ions=True, quant_window=3, quant_all=quant_all,
                    experimental_xic=True, top_k_fragments=9999))
    ctx.kernel_time_ms(reset=True)
    got, soa = hip_score(ctx, case, cfg, with_stats=True)
    got = {k: np.array(v, copy=True) for k, v in got.items()}
    _, feature_wide, _ = ctx.kernel_time_ms(reset=True)
    exp, _ = H.oracle_score(oracle_lib, case, cfg, soa=soa, n_threads=4, with_stats=True)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    assert np.array_equal(got["stat_matched_peaks"], exp["stat_matched_peaks"])
    assert exp["valid"].sum() > 1000
    kept = (got["fragment_mz_library"] > 0).sum(axis=1)
    assert kept.max() > 12  # (rows the fused kernel cannot hold)
    with monkeypatch.context() as mp:
        mp.setenv("ADH_DEBUG_NO_WIDE", "1")
        generic, _ = hip_score(ctx, case, cfg, soa=soa, with_stats=True)
        _, feature_generic, _ = ctx.kernel_time_ms(reset=True)
    for k in got:
        assert np.array_equal(generic[k], got[k], equal_nan=True), k
    assert feature_wide < feature_generic, (feature_wide, feature_generic)


@pytest.mark.parametrize("slab_peaks", [20_000, 333_333])
def test_run_staged_in_slabs_scores_identically(ctx, monkeypatch, slab_peaks):
    """Runs of 2^32 peaks and more are sorted into the transposed layout slab by slab (whole groups of cycle
    blocks, < 2^31 peaks each) and the bin table counts from the first entry of every group.  ADH_STAGE_SLAB_PEAKS
    shrinks the slabs so that a small run is staged that way: every table of the fused kernel, of the two-kernel
    path and of candidate selection must come out bit for bit as from one slab."""
    from alphadia_amd.selection import CandidateSelectionConfig, HipCandidateSelection

    g = H.load_scoring_golden("handler_default")
    names = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
                 fragment_mz_column="mz_library")
    scfg = CandidateSelectionConfig()
    scfg.update(dict(rt_tolerance=30.0, candidate_count=2))

    def everything():
        fused, soa = hip_score(ctx, g, g.config, with_stats=True)
        fused = {k: np.array(v, copy=True) for k, v in fused.items()}
        with monkeypatch.context() as mp:
            mp.setenv("ADH_DEBUG_NO_FUSED", "1")
            two, _ = hip_score(ctx, g, g.config, soa=soa, with_stats=True)
            two = {k: np.array(v, copy=True) for k, v in two.items()}
        sel = HipCandidateSelection(g.dia, g.library.precursor_df, g.library.fragment_df, scfg, device=0, **names)()
        return fused, two, sel

    one = everything()
    assert g.dia.mz_values.size > 3 * slab_peaks or slab_peaks > 100_000
    with monkeypatch.context() as mp:
        mp.setenv("ADH_STAGE_SLAB_PEAKS", str(slab_peaks))
        mp.setenv("ADH_BLOCK_CYCLES", "2")  # small blocks: several groups of blocks, so several slabs
        many = everything()
    with monkeypatch.context() as mp:
        mp.setenv("ADH_BLOCK_CYCLES", "2")
        same_blocks = everything()
    for a, b in ((one, many), (same_blocks, many)):
        for k in a[0]:
            assert np.array_equal(a[0][k], b[0][k], equal_nan=True), k
            assert np.array_equal(a[1][k], b[1][k], equal_nan=True), k
        import pandas as pd

        pd.testing.assert_frame_equal(a[2], b[2])
    assert one[0]["valid"].sum() > 100 and len(one[2]) > 100
    ctx.stage_run(g.dia, force=True)  # (leave the handle with the default staging)


def test_fitted_quadrupole_on_every_kernel(ctx, oracle_lib, monkeypatch):
    """A fitted quadrupole calibration (SimpleQuadrupoleJit.sigma / .delta_mu, quadrupole.py:72-113) reaches all
    four places the transfer function is evaluated: the fused kernel, the two-kernel register path
    (ADH_DEBUG_NO_FUSED), the generic kernel (class defaults) and the ion-mobility kernel - each against the
    oracle, which the reference's own run with these parameters pins (golden "fitted_quadrupole"); and the
    features must differ from those of the default calibration."""
    from alphadia_amd.scoring import assemble_candidates

    g = H.load_scoring_golden("fitted_quadrupole")
    assert g.config.quadrupole_sigma == (0.35, 0.12)
    got, soa = hip_score(ctx, g, g.config)
    exp, _ = H.oracle_score(oracle_lib, g, g.config, soa=soa)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    plain = CandidateScoringConfig()
    plain.update({k: getattr(g.config, k) for k in H.CFG_KEYS})
    base, _ = hip_score(ctx, g, plain)
    v = got["valid"].astype(bool) & base["valid"].astype(bool)
    assert v.sum() > 100 and not np.array_equal(got["features"][v], base["features"][v])
    with monkeypatch.context() as mp:
        mp.setenv("ADH_DEBUG_NO_FUSED", "1")
        two_kernel, _ = hip_score(ctx, g, g.config)
    for k in got:
        assert np.array_equal(two_kernel[k], got[k], equal_nan=True), k
    generic = CandidateScoringConfig()  # class defaults: quant_all / experimental_xic off -> adh_feature_kernel
    generic.quadrupole_sigma, generic.quadrupole_delta_mu = g.config.quadrupole_sigma, g.config.quadrupole_delta_mu
    got_g, soa_g = hip_score(ctx, g, generic)
    exp_g, _ = H.oracle_score(oracle_lib, g, generic, soa=soa_g)
    compare(got_g, exp_g, PPM_ABS_TOL_ORACLE)
    # the operator takes the calibration object the reference's operator takes (scoring.py:154,209-212)
    from types import SimpleNamespace

    from alphadia_amd.scoring import HipCandidateScoring

    calibration = SimpleNamespace(jit=SimpleNamespace(sigma=np.array(g.config.quadrupole_sigma),
                                                      delta_mu=np.array(g.config.quadrupole_delta_mu), cycle=g.dia.cycle))
    scorer = HipCandidateScoring(
        dia_data=g.dia, precursors_flat=g.library.precursor_df, fragments_flat=g.library.fragment_df,
        quadrupole_calibration=calibration, rt_column="rt_library", mobility_column="mobility_library",
        precursor_mz_column="mz_library", fragment_mz_column="mz_library", config=plain, device=0,
    )
    # the calibration stays on the scorer: the caller's config object is not written to (it may serve
    # another run, with another or no calibration)
    assert plain.quadrupole_sigma is None and plain.quadrupole_delta_mu is None
    assert scorer._kernel_config().quadrupole_sigma == g.config.quadrupole_sigma
    fdf, frdf = scorer(g.candidates_df, thread_count=4)
    assert np.array_equal(fdf["precursor_idx"].values, g.z["features_df_precursor_idx"])
    assert np.abs(frdf["mz_observed"].values - g.z["fragments_df_mz_observed"]).max() < 1e-4
    # ion mobility
    case = syn.make_timstof_case(n_precursors=300, n_cycles=60, config_id=46, per_precursor=2, n_ms2_frames=6,
                                 windows_per_frame=3, scan_max_index=96)
    soa_t = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    cfg = CandidateScoringConfig()
    cfg.update(dict(quant_all=True, experimental_xic=True, top_k_isotopes=3))
    cfg.quadrupole_sigma, cfg.quadrupole_delta_mu = (0.5, 0.3), (-0.6, 0.8)
    got_t = _hip_score_tims(ctx, case.dia, case.library.fragment_df, soa_t, cfg)
    exp_t = oracle_lib.score_timstof(case.dia, fragment_columns(case.library.fragment_df, "mz_library"),
                                     pack_assembled(soa_t), cfg.to_jitclass(), n_threads=8)
    compare(got_t, exp_t, PPM_ABS_TOL_ORACLE)
    assert got_t["valid"].sum() > 50


def test_config1_full_size_parity(ctx, oracle_lib):
    """BASELINE config 1: 1k-precursor library vs the 10-min (400 cycle) run, C=1."""
    case = syn.make_case(1000, 400, config_id=1, per_precursor=1)
    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15,
                    quant_all=True, experimental_xic=True))
    got, soa = hip_score(ctx, case, cfg, with_stats=True)
    exp, _ = H.oracle_score(oracle_lib, case, cfg, soa=soa, n_threads=8, with_stats=True)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    assert np.array_equal(got["stat_matched_peaks"], exp["stat_matched_peaks"])
    assert got["valid"].sum() > 500


def test_ragged_and_edge_cases(ctx, oracle_lib):
    """even cycle counts, 2-3 fragment precursors, O=2 windows, candidates at the run edges."""
    case = syn.make_case(
        600, 120, config_id=77, per_precursor=3, n_ms2=12, ms1_peaks=800, ms2_peaks=300,
        mz_lo=400, mz_hi=520, frag_mz_lo=200, frag_mz_hi=500, ms1_mz_range=(395, 530),
        ms2_mz_range=(195, 505), few_fragment_fraction=0.1, even_fraction=0.5, threads=2,
    )
    for upd in (dict(quant_all=True, experimental_xic=True, top_k_isotopes=3),
                dict(quant_all=False, experimental_xic=False),
                dict(quant_all=False, experimental_xic=True, top_k_fragments=5, quant_window=1)):
        cfg = CandidateScoringConfig()
        cfg.update(upd)
        got, soa = hip_score(ctx, case, cfg)
        exp, _ = H.oracle_score(oracle_lib, case, cfg, soa=soa, n_threads=8)
        compare(got, exp, PPM_ABS_TOL_ORACLE)


def test_empty_and_skipped(ctx, oracle_lib):
    g = H.load_scoring_golden("handler_default")
    soa = H.soa_for(g, g.config)
    empty = {k: (v[:0] if isinstance(v, np.ndarray) else v) for k, v in soa.items()}
    got, _ = hip_score(ctx, g, g.config, soa=empty)
    assert got["valid"].shape == (0,)
    soa["flags"][::3] = 1
    got, _ = hip_score(ctx, g, g.config, soa=soa)
    exp, _ = H.oracle_score(oracle_lib, g, g.config, soa=soa)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    assert not got["valid"][::3].any() and not got["precursor_idx"][::3].any()


def test_fragcomp_matches_reference_and_oracle(ctx, oracle_lib):
    import pandas as pd

    from alphadia_amd.fragcomp import FragmentCompetition

    z = np.load(H.golden_path("fragcomp.npz"))
    psm_df = pd.DataFrame({k[4:]: z[k] for k in z.files if k.startswith("psm_")})
    frag_df = pd.DataFrame({k[5:]: z[k] for k in z.files if k.startswith("frag_")})
    res = FragmentCompetition(rt_tol_seconds=3, mass_tol_ppm=15, device=0)(psm_df, frag_df, z["cycle"])
    assert np.array_equal(res["precursor_idx"].values, z["surviving_precursor_idx"])
    assert np.array_equal(res["rank"].values, z["surviving_rank"])


def test_fragcomp_reference_kats(ctx):
    import pandas as pd

    from alphadia_amd.fragcomp import FragmentCompetition

    # tests/unit_tests/fragcomp/test_fragcomp.py:38-57
    rt = np.array([10.0, 20.0, 20.0, 10.0, 10.0, 20])
    valid = ctx.fragcomp(np.array([0, 3]), np.array([3, 6]), rt, np.array([0, 10, 20, 30, 40, 50]),
                         np.array([10, 20, 30, 40, 50, 60]), np.tile(np.arange(100, 110), 6), 3, 15)
    assert np.all(valid == np.array([True, True, False, True, False, True]))
    # tests/unit_tests/fragcomp/test_fragcomp.py:60-100
    cycle = np.array([[[[90, 110]], [[190, 210]]]])
    psm_df = pd.DataFrame(
        {
            "precursor_idx": np.arange(6, dtype=np.uint32),
            "rt_observed": np.array([10.0, 20.0, 20.0, 10.0, 10.0, 20]),
            "valid": np.array([True] * 6),
            "mz_observed": np.array([100, 100, 100, 200, 200, 200]),
            "proba": np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6]),
            "rank": np.zeros(6, dtype=np.uint8),
        }
    )
    frag_df = pd.DataFrame(
        {
            "precursor_idx": np.repeat(np.arange(6, dtype=np.uint32), 10),
            "mz_observed": np.tile(np.arange(100, 110), 6),
            "rank": np.zeros(60, dtype=np.uint8),
        }
    )
    out = FragmentCompetition(device=0)(psm_df, frag_df, cycle).reset_index(drop=True)
    assert out["precursor_idx"].tolist() == [0, 1, 3, 5]
    assert out["_candidate_idx"].tolist() == [0, 1, 3, 5] and out["valid"].all()
    assert list(out.columns) == ["precursor_idx", "rt_observed", "valid", "mz_observed", "proba", "rank",
                                 "_candidate_idx"]


def test_fragcomp_large_random_vs_oracle(ctx, oracle_lib):
    rng = np.random.default_rng(5)
    n_win, per = 12, 900
    n = n_win * per
    rt = rng.uniform(0, 60, n).astype(np.float32)
    nfrag = rng.integers(0, 13, n)
    stop = np.cumsum(nfrag)
    start = stop - nfrag
    pool = np.sort(rng.uniform(200, 1800, 400)).astype(np.float32)
    mz = (rng.choice(pool, int(stop[-1])) * (1 + rng.normal(0, 5e-6, int(stop[-1])))).astype(np.float32)
    ws = np.arange(n_win) * per
    got = ctx.fragcomp(ws, ws + per, rt, start, stop, mz, 3, 15)
    exp = oracle_lib.fragcomp(ws, ws + per, rt, start, stop, mz, 3, 15, n_threads=8)
    assert np.array_equal(got, exp) and 0 < exp.sum() < n


def test_operator_and_handler_dataframes(ctx):
    """HipCandidateScoring / HipExtractionHandler: same frames as the reference produced."""
    from types import SimpleNamespace

    from alphadia_amd.extraction_handler import HipExtractionHandler, create_handler
    from alphadia_amd.scoring import DEFAULT_FEATURE_COLUMNS, HipCandidateScoring

    g = H.load_scoring_golden("handler_default")
    scorer = HipCandidateScoring(
        dia_data=g.dia, precursors_flat=g.library.precursor_df, fragments_flat=g.library.fragment_df,
        rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
        fragment_mz_column="mz_library", config=g.config, device=0,
    )
    fdf, frdf = scorer(g.candidates_df, thread_count=4)
    assert list(fdf.columns[:46]) == DEFAULT_FEATURE_COLUMNS
    assert sorted(fdf.columns) == sorted(g.z["features_df_columns"].tolist())
    # the reference appends the merged columns in the order of a Python set (scoring/utils.py:236-240)
    assert list(frdf.columns[:14]) == FRAGMENT_DF_COLUMNS
    assert sorted(frdf.columns) == sorted(g.z["fragments_df_columns"].tolist())
    assert np.array_equal(fdf["precursor_idx"].values, g.z["features_df_precursor_idx"])
    assert np.array_equal(fdf["rank"].values, g.z["features_df_rank"])
    assert len(frdf) == int(g.z["fragments_df_n"])
    assert np.array_equal(frdf["precursor_idx"].values, g.z["fragments_df_precursor_idx"])
    assert np.abs(frdf["mz_observed"].values - g.z["fragments_df_mz_observed"]).max() < 1e-4

    config = {"search": {"extraction_backend": "hip", "exclude_shared_ions": True, "quant_window": 3,
                         "quant_all": True, "experimental_xic": True, "top_k_fragments_scoring": 12,
                         "top_k_fragments_selection": 12},
              "general": {"thread_count": 4}}
    opt = SimpleNamespace(ms1_error=10, ms2_error=15, rt_error=30.0, mobility_error=0.1, num_candidates=2,
                          fwhm_rt=5.0, fwhm_mobility=0.01, score_cutoff=1.0)
    names = SimpleNamespace(get_rt_column=lambda: "rt_library", get_mobility_column=lambda: "mobility_library",
                            get_precursor_mz_column=lambda: "mz_library",
                            get_fragment_mz_column=lambda: "mz_library")
    reporter = SimpleNamespace(log_string=lambda *a, **k: None)
    handler = create_handler(config, opt, None, reporter, names)
    assert isinstance(handler, HipExtractionHandler)
    lib = SimpleNamespace(precursor_df=g.library.precursor_df, fragment_df=g.library.fragment_df)
    f2, fr2 = handler.score_and_quantify_candidates(g.candidates_df, g.dia, lib)
    pd_equal = f2[DEFAULT_FEATURE_COLUMNS].to_numpy()
    assert np.array_equal(pd_equal, fdf[DEFAULT_FEATURE_COLUMNS].to_numpy(), equal_nan=True)
    none, fr3 = handler.quantify_candidates(g.candidates_df, None, g.dia, lib, top_k_fragments=9999)
    assert none is None and len(fr3) >= len(fr2)
    # selection -> scoring through the handler, all on the GPU (a run without ion mobility)
    cands = handler.select_candidates(g.dia, lib, apply_cutoff=True)
    assert {"precursor_idx", "rank", "score", "frame_start", "frame_stop", "elution_group_idx", "decoy"} <= set(cands.columns)
    assert len(cands) > 200 and (cands["score"] > 1.0).all() and cands["rank"].max() <= 1
    f4, _ = handler.score_and_quantify_candidates(cands, g.dia, lib)
    assert len(f4) > 100
    with pytest.raises(ValueError):
        create_handler({"search": {"extraction_backend": "python"}}, opt, None, reporter, names)


@pytest.mark.parametrize("name", ["handler_default", "class_default", "manyfrag", "edges", "multiplex"])
def test_compact_operator_frames_equal_the_padded_ones(ctx, monkeypatch, name):
    """The operator on the compacted copy-out (adh_score_candidates_compact: valid rows and filled slots leave the
    device as columns) returns the frames of the padded path - same columns, order, dtypes and bits - and the frames
    own their memory.  Dtypes of the schema columns as validation/schemas.py:76-120 wants them."""
    from alphadia_amd.scoring import HipCandidateScoring

    g = H.load_scoring_golden(name)
    scorer = HipCandidateScoring(
        dia_data=g.dia, precursors_flat=g.library.precursor_df, fragments_flat=g.library.fragment_df,
        rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
        fragment_mz_column="mz_library", config=g.config, device=0,
    )
    monkeypatch.setenv("ADH_OPERATOR_PADDED", "1")
    f_pad, fr_pad = scorer(g.candidates_df, thread_count=4)
    monkeypatch.delenv("ADH_OPERATOR_PADDED")
    monkeypatch.setenv("ADH_CHUNK", "1024")  # several chunks even on the small goldens
    f_cmp, fr_cmp = scorer(g.candidates_df, thread_count=4)
    assert scorer.last_timings["wire_bytes"] > 0
    for a, b in ((f_pad, f_cmp), (fr_pad, fr_cmp)):
        assert list(a.columns) == list(b.columns)
        assert len(a) == len(b) and len(a) > 0
        for c in a.columns:
            assert a[c].dtype == b[c].dtype, c
            if a[c].dtype == object:
                assert (a[c].values == b[c].values).all(), c
            else:
                assert np.array_equal(a[c].to_numpy(), b[c].to_numpy(), equal_nan=True), c
    for c, dt in (("precursor_idx", np.uint32), ("rank", np.uint8), ("mz_observed", np.float32), ("elution_group_idx", np.uint32)):
        assert f_cmp[c].dtype == dt and fr_cmp[c].dtype == dt, c
    # the frames do not alias pooled buffers: another call leaves them as they are
    keep = f_cmp["mz_observed"].to_numpy().copy(), fr_cmp["height"].to_numpy().copy()
    scorer(g.candidates_df.iloc[::2].reset_index(drop=True), thread_count=4)
    assert np.array_equal(keep[0], f_cmp["mz_observed"].to_numpy(), equal_nan=True)
    assert np.array_equal(keep[1], fr_cmp["height"].to_numpy(), equal_nan=True)


def test_compact_output_capacity_is_checked(ctx):
    """adh_score_candidates_compact never writes beyond the capacities: a call that needs more fails, says what it
    needs, and the repeated call (Context.score_host_compact does that itself) succeeds."""
    import ctypes as C

    from alphadia_amd import _abi, runtime

    g = H.load_scoring_golden("handler_default")
    ctx.stage_run(g.dia, force=True)
    ctx.stage_fragments(*fragment_columns(g.library.fragment_df, "mz_library"), force=True)
    soa = H.soa_for(g, g.config)
    full = ctx.score_host_compact(pack_assembled(soa), g.config.to_jitclass())
    assert len(full["row"]) > 50 and len(full["fragment_row"]) > len(full["row"])
    small = ctx.score_host_compact(pack_assembled(soa), g.config.to_jitclass(), slots_per_row=0.01)  # grows and repeats
    for k in full:
        if isinstance(full[k], np.ndarray):
            assert np.array_equal(full[k], small[k], equal_nan=True), k
    # capacity arrays kept between calls (``buffers``): the same tables, written into the same memory
    keep: dict = {}
    first = ctx.score_host_compact(pack_assembled(soa), g.config.to_jitclass(), buffers=keep)
    snapshot = {k: v.copy() for k, v in first.items() if isinstance(v, np.ndarray)}
    again = ctx.score_host_compact(pack_assembled(soa), g.config.to_jitclass(), buffers=keep)
    assert len(keep) == 1
    for k, v in snapshot.items():
        assert np.array_equal(v, full[k], equal_nan=True) and np.array_equal(again[k], full[k], equal_nan=True), k
        assert np.shares_memory(first[k], again[k]), k
    # the raw call with room for ten rows: error code, needed counts, nothing written past the capacity
    cands = pack_assembled(soa)
    width = _abi.output_width(cands, int(g.config.top_k_fragments))
    out = _abi.CompactOutput()
    out.rows_capacity, out.slots_capacity, out.top_k = 10, 10, width
    guard = {}
    fields = dict(out._fields_)
    for name, dt in _abi.COMPACT_ROW_FIELDS + _abi.COMPACT_SLOT_FIELDS:
        guard[name] = np.full(64, 7, dtype=dt)
        setattr(out, name, guard[name].ctypes.data_as(fields[name]))
    guard["features"] = np.full((_abi.NUM_FEATURES, 10), 7, dtype=np.float32)
    out.features = guard["features"].ctypes.data_as(fields["features"])
    cfg = _abi.pack_config(g.config.to_jitclass())
    rc = runtime.lib.adh_score_candidates_compact(ctx._h, cands.ref(), C.byref(cfg), C.byref(out))
    assert rc != 0 and int(out.n_rows) == len(full["row"]) and int(out.n_slots) == len(full["fragment_row"])
    for name, _ in _abi.COMPACT_ROW_FIELDS + _abi.COMPACT_SLOT_FIELDS:
        assert (guard[name][10:] == 7).all(), name


def test_compact_output_overflow_at_a_later_chunk(ctx, monkeypatch):
    """the capacity runs out at chunk >= 1 while the host team is still unpacking earlier chunks.  The
    chunks that fit land where the full call puts them, nothing is written past the capacity, the counts say what
    is needed."""
    import ctypes as C

    from alphadia_amd import _abi, runtime

    case = syn.make_case(
        2400, 120, config_id=78, per_precursor=3, n_ms2=12, ms1_peaks=800, ms2_peaks=300,
        mz_lo=400, mz_hi=520, frag_mz_lo=200, frag_mz_hi=500, ms1_mz_range=(395, 530),
        ms2_mz_range=(195, 505), threads=2,
    )
    cfg = CandidateScoringConfig()
    cfg.update(dict(quant_all=True, experimental_xic=True, top_k_isotopes=3))
    ctx.stage_run(case.dia, force=True)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"), force=True)
    soa = H.soa_for(case, cfg)
    monkeypatch.setenv("ADH_CHUNK", "1024")  # 7200 candidates -> eight chunks
    full = ctx.score_host_compact(pack_assembled(soa), cfg.to_jitclass())
    nr, ns = len(full["row"]), len(full["fragment_row"])
    assert nr > 3000 and ns > nr
    cands = pack_assembled(soa)
    width = _abi.output_width(cands, int(cfg.top_k_fragments))
    fields = dict(_abi.CompactOutput._fields_)
    for share in (0.3, 0.55, 0.8):
        cap_r, cap_s = int(nr * share), int(ns * share)
        out = _abi.CompactOutput()
        out.rows_capacity, out.slots_capacity, out.top_k = cap_r, cap_s, width
        guard = {}
        for name, dt in _abi.COMPACT_ROW_FIELDS:
            guard[name] = np.full(cap_r + 4096, 7, dtype=dt)
        for name, dt in _abi.COMPACT_SLOT_FIELDS:
            guard[name] = np.full(cap_s + 65536, 7, dtype=dt)
        guard["features"] = np.full((_abi.NUM_FEATURES, cap_r), 7, dtype=np.float32)
        for name, a in guard.items():
            setattr(out, name, a.ctypes.data_as(fields[name]))
        pcfg = _abi.pack_config(cfg.to_jitclass())
        for _ in range(3):  # (a race does not show every time)
            rc = runtime.lib.adh_score_candidates_compact(ctx._h, cands.ref(), C.byref(pcfg), C.byref(out))
            assert rc != 0 and int(out.n_rows) == nr and int(out.n_slots) == ns
            for name, _dt in _abi.COMPACT_ROW_FIELDS:
                assert (guard[name][cap_r:] == 7).all(), name
            for name, _dt in _abi.COMPACT_SLOT_FIELDS:
                assert (guard[name][cap_s:] == 7).all(), name
        # whole chunks that fit are where the full call has them: rows of the first chunk (candidates < 900)
        first = int(np.searchsorted(full["row"], 900))
        assert first > 100 and np.array_equal(guard["row"][:first], full["row"][:first])
        assert np.array_equal(guard["features"][:, :first], full["features"][:, :first], equal_nan=True)


def test_frame_stop_clipped_to_the_last_frame(ctx, oracle_lib):
    """The selection step clips frame_stop to frame_max_index (selection.py:488-491): scoring takes
    the floor of the cycle count like get_dense does."""
    g = H.load_scoring_golden("handler_default")
    cand = g.candidates_df.copy()
    last = g.dia.n_spectra - 1
    L = g.dia.cycle_len
    late = np.argsort(cand["frame_stop"].values)[-40:]
    width = (cand["frame_stop"].values - cand["frame_start"].values)[late]
    cand.loc[cand.index[late], "frame_stop"] = last
    cand.loc[cand.index[late], "frame_start"] = ((last + 1) // L) * L - width
    cand.loc[cand.index[late], "frame_center"] = cand["frame_start"].values[late] + (width // (2 * L)) * L
    case = type("Case", (), {})()
    case.dia, case.library, case.candidates_df = g.dia, g.library, cand
    got, soa = hip_score(ctx, case, g.config)
    exp, _ = H.oracle_score(oracle_lib, case, g.config, soa=soa, n_threads=4)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    assert got["valid"].sum() > 100


def test_multiplex_requantification_handler(ctx):
    """HipMultiplexingRequantificationHandler.requantify: reference-channel PSMs in, the feature
    table handed to the FDR manager equals what the reference computed for the same candidates
    (multiplex golden; groups that kept their reference channel)."""
    from types import SimpleNamespace

    from alphadia_amd.multiplexing import HipMultiplexingRequantificationHandler
    from alphadia_amd.scoring import DEFAULT_FEATURE_COLUMNS

    g = H.load_scoring_golden("multiplex")
    pdf = g.library.precursor_df
    cand = g.candidates_df.merge(pdf[["precursor_idx", "channel", "decoy"]], on="precursor_idx")
    # one reference-channel PSM per elution group (boxes are shared by the channels of a group)
    ref = cand.sort_values("channel").groupby("elution_group_idx").first().reset_index()
    ref["precursor_idx"] = ((ref["precursor_idx"] // 4) * 4).astype(np.uint32)
    ref["channel"] = np.uint32(0)
    ref["proba"] = np.linspace(0.0, 0.5, len(ref)).astype(np.float32)
    seen = {}

    def fit_predict(features, **kw):
        seen["features"], seen["kw"] = features, kw
        return features

    cfg = {"multiplexing": {"reference_channel": 0, "target_channels": "4,8,12", "decoy_channel": -1,
                            "competitive_scoring": True},
           "search": {"experimental_xic": True}}
    names = SimpleNamespace(get_rt_column=lambda: "rt_library", get_mobility_column=lambda: "mobility_library",
                            get_precursor_mz_column=lambda: "mz_library",
                            get_fragment_mz_column=lambda: "mz_library")
    lib = SimpleNamespace(precursor_df_unfiltered=pdf, fragment_df=g.library.fragment_df,
                          _fragment_df=g.library.fragment_df)
    handler = HipMultiplexingRequantificationHandler(
        cfg, None, SimpleNamespace(fit_predict=fit_predict), SimpleNamespace(log_string=lambda *a, **k: None),
        names, lib, device=0)
    handler._dia = None
    out = handler.requantify(g.dia, ref)
    assert out is seen["features"]
    assert seen["kw"] == dict(decoy_strategy="channel", competitive=True, decoy_channel=-1)
    feats = seen["features"].set_index("precursor_idx")
    exp_valid = g.expected["valid"].astype(bool)
    exp_pidx = g.expected["precursor_idx"][exp_valid]
    exp_feat = g.expected["features"][exp_valid]
    assert set(exp_pidx) <= set(feats.index), "a candidate the reference scored is missing"
    got = feats.loc[exp_pidx, DEFAULT_FEATURE_COLUMNS].to_numpy()
    keep = [j for j in range(46) if j not in (8, 9, 10, 41, 42, 45)]  # ppm columns: see PPM_ABS_TOL_GOLDEN
    err = H.rel_err(got[:, keep], exp_feat[:, keep])
    assert np.quantile(err, 0.999) < 2e-3
    # groups that lost their reference channel in the golden are scored here (their PSM exists)
    assert len(feats) > exp_valid.sum()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("ADH_FUZZ_SEEDS", "10")))))
def test_randomized_shapes_and_settings(ctx, oracle_lib, seed):
    """Differential test: random run geometry, library shape and scoring settings, HIP vs oracle."""
    rng = np.random.default_rng(1000 + seed)
    n_ms2 = int(rng.integers(3, 14))
    mz_lo = 400.0
    mz_hi = mz_lo + 10.0 * n_ms2
    case = syn.make_case(
        int(rng.integers(80, 260)), int(rng.integers(40, 110)), config_id=300 + seed,
        per_precursor=int(rng.integers(1, 4)), n_ms2=n_ms2, ms1_peaks=int(rng.integers(100, 900)),
        ms2_peaks=int(rng.integers(40, 400)), mz_lo=mz_lo, mz_hi=mz_hi, frag_mz_lo=200,
        frag_mz_hi=float(rng.choice([320.0, 500.0])), ms1_mz_range=(395, mz_hi + 25),
        ms2_mz_range=(195, 520), few_fragment_fraction=float(rng.choice([0.0, 0.15])),
        even_fraction=float(rng.choice([0.0, 0.5])), planted_fraction=float(rng.uniform(0.2, 0.9)), threads=1,
        # library shape: 12 fragments (predicted library), or ragged 5..60 (empirical / transfer libraries)
        k_fragments=12 if rng.random() < 0.5 else (int(rng.integers(4, 14)), int(rng.integers(14, 61))),
    )
    if rng.random() < 0.4:  # a run that ends inside a cycle
        k = int(rng.integers(1, case.dia.cycle_len))
        d = case.dia
        d.rt_values = d.rt_values[:-k]
        d.peak_start_idx_list = d.peak_start_idx_list[:-k]
        d.peak_stop_idx_list = d.peak_stop_idx_list[:-k]
        keep = case.candidates_df["frame_stop"].values <= d.rt_values.shape[0]
        case.candidates_df = case.candidates_df[keep].reset_index(drop=True)
    if rng.random() < 0.5:  # shared fragments
        card = case.library.fragment_df["cardinality"].values.copy()
        card[rng.random(card.size) < 0.2] = 2
        case.library.fragment_df["cardinality"] = card
    upd = dict(
        top_k_fragments=int(rng.choice([4, 6, 12, 16, 20, 33, 9999])), top_k_isotopes=int(rng.integers(1, 5)),
        precursor_mz_tolerance=float(rng.choice([5, 10, 40, 150])),
        fragment_mz_tolerance=float(rng.choice([7, 15, 60, 200])),
        exclude_shared_ions=bool(rng.integers(0, 2)), quant_window=int(rng.integers(1, 6)),
        quant_all=bool(rng.integers(0, 2)), experimental_xic=bool(rng.integers(0, 2)),
    )
    cfg = CandidateScoringConfig()
    cfg.update(upd)
    got, soa = hip_score(ctx, case, cfg, with_stats=True)
    exp, _ = H.oracle_score(oracle_lib, case, cfg, soa=soa, n_threads=4, with_stats=True)
    # experimental_xic = False: the K x K contraction runs on MFMA, its summation order is not the
    # oracle's (nor is the reference's BLAS order defined): correlations near zero get an absolute floor
    compare(got, exp, PPM_ABS_TOL_ORACLE, corr_abs=0.0 if upd["experimental_xic"] else 2e-6)
    assert np.array_equal(got["stat_matched_peaks"], exp["stat_matched_peaks"]), upd


def test_invalid_inputs_fail_loudly(ctx):
    from alphadia_amd.runtime import HipBackendError

    g = H.load_scoring_golden("handler_default")
    soa = H.soa_for(g, g.config)
    ctx.stage_run(g.dia, force=True)
    ctx.stage_fragments(*fragment_columns(g.library.fragment_df, "mz_library"), force=True)
    bad = dict(soa)
    bad["frame_stop"] = soa["frame_stop"].copy()
    bad["frame_stop"][0] = g.dia.n_spectra + g.dia.cycle_len * 5
    with pytest.raises(HipBackendError, match="frame limits"):
        ctx.score_host(pack_assembled(bad), g.config.to_jitclass())
    bad = dict(soa)
    bad["frag_stop_idx"] = soa["frag_stop_idx"].copy()
    bad["frag_stop_idx"][0] = 10**9
    with pytest.raises(HipBackendError, match="fragment slice"):
        ctx.score_host(pack_assembled(bad), g.config.to_jitclass())


def test_config2_scale_properties(ctx, oracle_lib):
    """BASELINE config 2 shape at reduced run length (same 61-spectrum cycle, same peak
    densities): order independence, bounded sample vs the oracle, matched-peak checksum."""
    case = syn.make_case(20000, 1200, config_id=2, per_precursor=3, threads=16)
    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15,
                    quant_all=True, experimental_xic=True))
    got, soa = hip_score(ctx, case, cfg, with_stats=True)
    n = len(soa["precursor_idx"])
    # (1) a permuted candidate table gives the same rows
    perm = np.random.default_rng(0).permutation(n)
    soa_p = {k: (v[perm] if isinstance(v, np.ndarray) and v.shape[:1] == (n,) else v) for k, v in soa.items()}
    got_p, _ = hip_score(ctx, case, cfg, soa=soa_p, with_stats=True)
    for k in got:
        assert np.array_equal(got[k][perm], got_p[k], equal_nan=True), k
    # (2) the first 6000 rows equal the oracle's
    from alphadia_amd.distributed import slice_soa

    sub = slice_soa(soa, 0, 6000)
    exp, _ = H.oracle_score(oracle_lib, case, cfg, soa=sub, n_threads=16, with_stats=True)
    compare({k: v[:6000] for k, v in got.items()}, exp, PPM_ABS_TOL_ORACLE)
    assert np.array_equal(got["stat_matched_peaks"][:6000], exp["stat_matched_peaks"])
    # (3) planted precursors at rank 0 are found with near-complete fragment evidence
    planted = case.apex_cycle[soa["precursor_idx"]] >= 0
    r0 = planted & (soa["rank"] == 0)
    assert got["valid"][r0].mean() > 0.99
    assert np.nanmean(got["features"][r0][:, 20]) > 0.9 > np.nanmean(got["features"][~planted][:, 20])


# ---- ion-mobility (timsTOF) layout ---------------------------------------------------

def _tims_case_from_golden():
    import pandas as pd

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


def _hip_score_tims(ctx, dia, fragment_df, soa, cfg, with_stats=False):
    ctx.stage_run(dia, force=True)
    ctx.stage_fragments(*fragment_columns(fragment_df, "mz_library"), force=True)
    return ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=with_stats)


def test_timstof_golden_inputs(ctx, oracle_lib):
    from alphadia_amd.scoring import assemble_candidates

    z, dia, fragment_df, precursor_df, cand, cfg = _tims_case_from_golden()
    soa = assemble_candidates(cand, precursor_df, "mz_library")
    got = _hip_score_tims(ctx, dia, fragment_df, soa, cfg, with_stats=True)
    exp = oracle_lib.score_timstof(dia, fragment_columns(fragment_df, "mz_library"), pack_assembled(soa),
                                   cfg.to_jitclass(), with_stats=True)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
    assert np.array_equal(got["stat_matched_peaks"], exp["stat_matched_peaks"])
    golden = {n: z["out_" + n] for n in H.OUT_NAMES}
    compare(got, golden, PPM_ABS_TOL_GOLDEN, rel_tol=REL_TOL, corr_abs=1e-3)  # the north-star bar, as for the AlphaRaw layout
    v = got["valid"].astype(bool)
    assert v.sum() > 100 and (got["features"][v][:, 29] != 0).sum() > 50


def test_timstof_larger_case_all_configs(ctx, oracle_lib):
    from alphadia_amd.scoring import assemble_candidates

    case = syn.make_timstof_case(n_precursors=400, n_cycles=60, config_id=44, per_precursor=3,
                                 n_ms2_frames=6, windows_per_frame=3, scan_max_index=96)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    for upd in (dict(quant_all=True, experimental_xic=True, top_k_isotopes=3, precursor_mz_tolerance=10,
                     fragment_mz_tolerance=15),
                dict(quant_all=False, experimental_xic=False),
                dict(quant_all=True, experimental_xic=True, top_k_fragments=6, fragment_mz_tolerance=40)):
        cfg = CandidateScoringConfig()
        cfg.update(upd)
        got = _hip_score_tims(ctx, case.dia, case.library.fragment_df, soa, cfg)
        exp = oracle_lib.score_timstof(case.dia, fragment_columns(case.library.fragment_df, "mz_library"),
                                       pack_assembled(soa), cfg.to_jitclass(), n_threads=8)
        compare(got, exp, PPM_ABS_TOL_ORACLE)
        assert got["valid"].sum() > 100


@pytest.mark.parametrize("quant_all", [True, False])
@pytest.mark.parametrize("shape", ["small_tiles", "larger_tiles", "dense_tiles"])
def test_timstof_split_feature_path_equals_the_one_kernel_path(ctx, oracle_lib, monkeypatch, shape, quant_all):
    """Round 4 splits the ion-mobility feature kernel after its passes over the tiles; the profile phase runs
    four candidates per wavefront from a few KB of profiles per candidate (adh_features_im2.hip).  Same numbers,
    bit for bit, as the one-kernel path (ADH_DEBUG_IM_NO_SPLIT=1) - for the small-tile class, the common class
    and candidates whose tiles were materialised - and equal to the oracle."""
    from alphadia_amd.scoring import assemble_candidates

    kw = dict(small_tiles=dict(scan_max_index=64, n_cycles=70), larger_tiles=dict(scan_max_index=96, n_cycles=90, h_range=(8, 14), hs_range=(14, 19)),
              dense_tiles=dict(scan_max_index=64, n_cycles=70))[shape]
    case = syn.make_timstof_case(n_precursors=260, config_id=47, per_precursor=2, n_ms2_frames=5, windows_per_frame=2,
                                 planted_fraction=0.6, **kw)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, quant_all=quant_all, experimental_xic=True))
    with monkeypatch.context() as mp:
        if shape == "dense_tiles":
            mp.setenv("ADH_DEBUG_IM", "8")  # (the gather kernel materialises every tile)
        split = _hip_score_tims(ctx, case.dia, case.library.fragment_df, soa, cfg, with_stats=True)
        split = {k: np.array(v, copy=True) for k, v in split.items()}
        mp.setenv("ADH_DEBUG_IM_NO_SPLIT", "1")
        one = _hip_score_tims(ctx, case.dia, case.library.fragment_df, soa, cfg, with_stats=True)
    for name, ref in one.items():
        assert np.array_equal(np.asarray(split[name]), np.asarray(ref), equal_nan=np.asarray(ref).dtype.kind == "f"), name
    exp = oracle_lib.score_timstof(case.dia, fragment_columns(case.library.fragment_df, "mz_library"),
                                   pack_assembled(soa), cfg.to_jitclass(), n_threads=8, with_stats=True)
    compare(split, exp, PPM_ABS_TOL_ORACLE)
    v = split["valid"].astype(bool)
    assert v.sum() > 80 and (split["features"][v][:, 29] != 0).sum() > 30
    # two-observation candidates (precursor isotopes across two isolation windows) take the split path too
    assert (split["features"][v][:, 17] == 2).sum() > 5


def test_timstof_search_indices_and_tile_modes_agree(ctx, monkeypatch):
    """The staged search indices (m/z lookup table, (TOF bin, cycle) table - one column per cycle or per
    block of cycles), the two forms of the tiles (sparse entry lists, dense tiles), the two forms of the
    feature assembly (lane-parallel sums, one-lane loops) and the feature kernel's instantiations (capacities
    fixed at compile time, capacities of the launch) are different ways to the same numbers: every output table
    must come out bit for bit the same."""
    from alphadia_amd.scoring import assemble_candidates

    case = syn.make_timstof_case(n_precursors=300, n_cycles=70, config_id=45, per_precursor=2, n_ms2_frames=5,
                                 windows_per_frame=2, scan_max_index=64, planted_fraction=0.6)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, quant_all=True, experimental_xic=True))
    base = _hip_score_tims(ctx, case.dia, case.library.fragment_df, soa, cfg, with_stats=True)
    base = {k: np.array(v, copy=True) for k, v in base.items()}
    assert base["valid"].sum() > 100
    # ADH_DEBUG_IM_NO_SPLIT: the one-kernel feature path against the split one of round 4 (tile passes, then the
    # profile phase with four candidates per wavefront, adh_features_im2.hip), which is the default `base` ran
    for env in (dict(ADH_IM_INDEX="0"), dict(ADH_IM_INDEX_MB="1"), dict(ADH_DEBUG_IM="8"),
                dict(ADH_DEBUG_IM="21"), dict(ADH_DEBUG_IM_DYNAMIC_LAYOUT="1"), dict(ADH_DEBUG_IM_NO_SPLIT="1"),
                # the tile-ordered copy of the events the gather reads by default (DevTims::tile_ev): not built, built
                # but not used, tiles smaller than any candidate, oblong tiles, one tile for the whole run
                # (ADH_DEBUG_IM=14: the gather's windows in batches that are certain to fit, never all at once)
                dict(ADH_DEBUG_IM="14"), dict(ADH_DEBUG_IM="14", ADH_IM_TILED="0"),
                dict(ADH_IM_TILED="0"), dict(ADH_DEBUG_IM_NO_TILES="1"), dict(ADH_IM_TILE_SHIFTS="2,3"),
                dict(ADH_IM_TILE_SHIFTS="5,4"), dict(ADH_IM_TILE_SHIFTS="1,6"), dict(ADH_IM_TILE_SHIFTS="12,12"),
                # (round 6) the tiles keyed by the frame of the cycle (the default) against frames sharing a tile, in
                # several shapes; the tile phase four candidates per wavefront against the one-candidate kernel; tile
                # and profile phase in one kernel against two
                dict(ADH_IM_TILE_FRAMES="0"), dict(ADH_IM_TILE_FRAMES="0", ADH_IM_TILE_SHIFTS="2,3"),
                dict(ADH_IM_TILE_FRAMES="0", ADH_DEBUG_IM="14"), dict(ADH_DEBUG_IM_TILE1="1"), dict(ADH_DEBUG_IM_NO_FUSE4="1"),
                dict(ADH_DEBUG_IM_TILE1_TWO="1"), dict(ADH_DEBUG_IM_NO_FUSE4="1", ADH_DEBUG_IM="8")):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            got = _hip_score_tims(ctx, case.dia, case.library.fragment_df, soa, cfg, with_stats=True)
            for name, ref in base.items():
                assert np.array_equal(np.asarray(got[name]), ref, equal_nan=ref.dtype.kind == "f"), (env, name)
    ctx.stage_run(case.dia, force=True)  # (leave the handle with the default indices)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("ADH_FUZZ_SEEDS_IM", "6")))))
def test_timstof_randomized(ctx, oracle_lib, seed):
    """Differential test on the ion-mobility layout: random geometry and settings, HIP vs oracle."""
    from alphadia_amd.scoring import assemble_candidates

    rng = np.random.default_rng(2000 + seed)
    case = syn.make_timstof_case(
        n_precursors=int(rng.integers(60, 200)), n_cycles=int(rng.integers(25, 60)), config_id=500 + seed,
        per_precursor=int(rng.integers(1, 4)), n_ms2_frames=int(rng.integers(2, 7)),
        windows_per_frame=int(rng.integers(1, 4)), scan_max_index=int(rng.choice([48, 64, 96])),
        events_per_push=float(rng.choice([15.0, 40.0, 80.0])), planted_fraction=float(rng.uniform(0.2, 0.8)),
    )
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    upd = dict(
        top_k_fragments=int(rng.choice([5, 12, 16])), top_k_isotopes=int(rng.integers(1, 5)),
        precursor_mz_tolerance=float(rng.choice([10, 40])), fragment_mz_tolerance=float(rng.choice([15, 60])),
        quant_window=int(rng.integers(1, 5)), quant_all=bool(rng.integers(0, 2)),
        experimental_xic=bool(rng.integers(0, 2)),
    )
    cfg = CandidateScoringConfig()
    cfg.update(upd)
    got = _hip_score_tims(ctx, case.dia, case.library.fragment_df, soa, cfg, with_stats=True)
    exp = oracle_lib.score_timstof(case.dia, fragment_columns(case.library.fragment_df, "mz_library"),
                                   pack_assembled(soa), cfg.to_jitclass(), n_threads=8, with_stats=True)
    # experimental_xic = False: the K x K contraction runs on MFMA, its summation order is not the
    # oracle's (nor is the reference's BLAS order defined): correlations near zero get an absolute floor
    compare(got, exp, PPM_ABS_TOL_ORACLE, corr_abs=0.0 if upd["experimental_xic"] else 2e-6)
    assert np.array_equal(got["stat_matched_peaks"], exp["stat_matched_peaks"]), upd


def test_switching_run_layouts_on_one_handle(ctx, oracle_lib):
    g = H.load_scoring_golden("handler_default")
    got, soa = hip_score(ctx, g, g.config)
    exp, _ = H.oracle_score(oracle_lib, g, g.config, soa=soa)
    compare(got, exp, PPM_ABS_TOL_ORACLE)
