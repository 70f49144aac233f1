"""f-3 as written: scoring -> classifier -> q-values -> fragment competition -> best row per group
with the feature / fragment tables staying in HBM.  The device-resident stage must give what the
host-interface stage (perform_fdr on the DataFrames) gives, and must not pull the tables back."""

import numpy as np
import pandas as pd
import pytest

import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("competitive, with_fragcomp", [(True, True), (False, False), (True, False)])
def test_resident_fdr_equals_host_fdr(competitive, with_fragcomp):
    from alphadia_amd import fdr, runtime
    from alphadia_amd.scoring import (DEFAULT_FEATURE_COLUMNS, CandidateScoringConfig, HipCandidateScoring,
                                      assemble_candidates)

    case = syn.make_case(5000, 260, config_id=78, per_precursor=2, planted_fraction=0.5, threads=4)
    dia, pdf, fdf = case.dia, case.library.precursor_df, case.library.fragment_df
    names = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
                 fragment_mz_column="mz_library")
    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                    experimental_xic=True))
    scorer = HipCandidateScoring(dia_data=dia, precursors_flat=pdf, fragments_flat=fdf, config=cfg, device=0, **names)
    features_df, fragments_df = scorer(case.candidates_df, thread_count=4)
    ctx = runtime.get_context(0)
    # classifier columns: the 46 kernel features plus the two host-derived ones of the reference's list
    cols = [c for c in DEFAULT_FEATURE_COLUMNS if c not in ("mobility_observed", "base_width_mobility")] + ["delta_rt", "mz_library"]
    kw = dict(test_size=0.2, batch_size=500, learning_rate=0.001, epochs=4, random_state=11)

    host = fdr.perform_fdr(fdr.HipBinaryClassifier(**kw), cols, features_df[features_df["decoy"] == 0].copy(),
                           features_df[features_df["decoy"] == 1].copy(), competitive=competitive, group_channels=True,
                           df_fragments=fragments_df if with_fragcomp else None,
                           dia_cycle=dia.cycle if with_fragcomp else None, random_state=5, device=0)

    # the same stage on the tables the scoring call left in HBM: one metadata row per table row
    soa = assemble_candidates(case.candidates_df, scorer.precursors_flat_df, "mz_library")
    lib = scorer.precursors_flat_df.iloc[soa["prec_row"]]
    meta = pd.DataFrame({"precursor_idx": soa["precursor_idx"], "rank": soa["rank"], "decoy": soa["decoy"],
                         "elution_group_idx": soa["elution_group_idx"], "channel": soa["channel"],
                         "rt_library": lib["rt_library"].to_numpy(), "mz_library": lib["mz_library"].to_numpy()})
    assert len(meta) == int(ctx.device_tables().n)
    ctx.d2h_bytes(reset=True)
    res = fdr.perform_fdr_resident(fdr.HipBinaryClassifier(**kw), cols, meta, competitive=competitive,
                                   group_channels=True, dia_cycle=dia.cycle if with_fragcomp else None,
                                   random_state=5, device=0)
    moved = ctx.d2h_bytes()
    n_table = len(meta)
    # the 184-byte feature rows (and 600 bytes of fragment tables) per candidate stayed on the GPU:
    # what came back is row maps, metric probes of the training loop and the final triples
    assert moved < 60 * n_table, (moved, n_table)

    assert len(res) == len(host) > 500
    assert np.array_equal(res["precursor_idx"].to_numpy(), host["precursor_idx"].to_numpy())
    assert np.array_equal(res["rank"].to_numpy(), host["rank"].to_numpy())
    assert np.allclose(res["proba"].to_numpy(), host["proba"].to_numpy(), rtol=0, atol=1e-6)
    assert np.allclose(res["qval"].to_numpy(), host["qval"].to_numpy(), rtol=1e-12, atol=0)
    if competitive:
        assert res.groupby(["elution_group_idx", "channel"]).size().max() == 1
    ids = res[(res["qval"] <= 0.01) & (res["decoy"] == 0)]
    assert len(ids) > 0.6 * (case.apex_cycle >= 0).sum()


def test_resident_fdr_needs_the_tables_it_was_staged_from():
    from alphadia_amd import fdr, runtime
    from alphadia_amd.runtime import HipBackendError

    ctx = runtime.get_context(0)
    mlp = runtime.DeviceMlp(ctx, 4, [8], 2)
    try:
        with pytest.raises(HipBackendError):
            mlp.predict_resident()
        with pytest.raises(HipBackendError):  # row count does not match the tables in HBM
            mlp.stage_rows_device([0, 1, 2, 3], np.zeros(int(ctx.device_tables().n) + 7, np.uint8))
        with pytest.raises(HipBackendError):
            ctx.fdr_resident(mlp, np.zeros(3, np.int64))
    finally:
        mlp.close()
    assert fdr.perform_fdr_resident is not None


def test_resident_fdr_with_too_few_psms_answers_like_the_host_stage():
    """fdr.py:125-137: when the train / test split would be empty, perform_fdr returns every usable PSM with
    qval = proba = 1 instead of failing; so does the device-resident stage, also for an empty table."""
    from alphadia_amd import fdr, runtime
    from alphadia_amd.scoring import (DEFAULT_FEATURE_COLUMNS, CandidateScoringConfig, HipCandidateScoring,
                                      assemble_candidates)

    case = syn.make_case(40, 80, config_id=79, per_precursor=1, n_ms2=8, ms1_peaks=400, ms2_peaks=150, mz_lo=400, mz_hi=480,
                         frag_mz_lo=200, frag_mz_hi=350, ms1_mz_range=(395, 500), ms2_mz_range=(195, 355),
                         planted_fraction=1.0, threads=1)
    pdf, fdf = case.library.precursor_df, case.library.fragment_df
    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                    experimental_xic=True))
    scorer = HipCandidateScoring(dia_data=case.dia, precursors_flat=pdf, fragments_flat=fdf, config=cfg, device=0,
                                 rt_column="rt_library", mobility_column="mobility_library",
                                 precursor_mz_column="mz_library", fragment_mz_column="mz_library")
    cols = [c for c in DEFAULT_FEATURE_COLUMNS if c not in ("mobility_observed", "base_width_mobility")]
    for keep in (1, 0):  # one candidate: its train split is empty; none at all
        cands = case.candidates_df.iloc[:keep]
        features_df, _ = scorer(cands, thread_count=1)
        soa = assemble_candidates(cands, scorer.precursors_flat_df, "mz_library")
        meta = pd.DataFrame({"precursor_idx": soa["precursor_idx"], "rank": soa["rank"], "decoy": soa["decoy"],
                             "elution_group_idx": soa["elution_group_idx"], "channel": soa["channel"]})
        res = fdr.perform_fdr_resident(fdr.HipBinaryClassifier(epochs=1, random_state=1), cols, meta, random_state=5, device=0)
        host = fdr.perform_fdr(fdr.HipBinaryClassifier(epochs=1, random_state=1), cols,
                               features_df[features_df["decoy"] == 0].copy(), features_df[features_df["decoy"] == 1].copy(),
                               random_state=5, device=0)
        assert len(res) == len(host) == len(features_df) <= keep
        assert (res["qval"] == 1.0).all() and (res["proba"] == 1.0).all()
        assert (host["qval"] == 1.0).all() and (host["proba"] == 1.0).all()
        assert np.array_equal(res["precursor_idx"].to_numpy(), host["precursor_idx"].to_numpy())
