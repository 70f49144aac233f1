"""Selection -> scoring -> FDR on one synthetic run, every stage on the GPU: the stages must work
together the way the reference's workflow chains them (peptidecentric.py:190-229), and planted
peptides must come out the other end."""

from __future__ import annotations

import numpy as np
import pytest

import synthetic as syn


@pytest.mark.gpu
def test_select_score_fdr_end_to_end():
    from alphadia_amd import fdr
    from alphadia_amd.scoring import DEFAULT_FEATURE_COLUMNS, CandidateScoringConfig, HipCandidateScoring
    from alphadia_amd.selection import CandidateSelectionConfig, HipCandidateSelection

    case = syn.make_case(6000, 300, config_id=77, per_precursor=1, planted_fraction=0.5, threads=4)
    dia, pdf, fdf = case.dia, case.library.precursor_df.copy(), case.library.fragment_df
    planted = case.apex_cycle >= 0
    L = dia.cycle.shape[1]
    rng = np.random.default_rng(1)
    # the library knows roughly where a peptide elutes (the planted apex is random in the generator)
    rt_apex = dia.rt_values[np.clip(case.apex_cycle, 0, None) * L]
    pdf.loc[planted, "rt_library"] = (rt_apex[planted] + rng.normal(0, 4, planted.sum())).astype(np.float32)
    names = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
                 fragment_mz_column="mz_library")

    scfg = CandidateSelectionConfig()
    scfg.update(dict(rt_tolerance=30.0, candidate_count=2, precursor_mz_tolerance=10, fragment_mz_tolerance=15))
    cands = HipCandidateSelection(dia, pdf, fdf, scfg, fwhm_rt=scfg.peak_len_rt, fwhm_mobility=scfg.peak_len_mobility,
                                  **names)()
    assert len(cands) > len(pdf) and set(cands["precursor_idx"]) <= set(pdf["precursor_idx"])

    cfg = CandidateScoringConfig()
    cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                    experimental_xic=True))
    scorer = HipCandidateScoring(dia_data=dia, precursors_flat=pdf, fragments_flat=fdf, config=cfg, device=0, **names)
    features_df, fragments_df = scorer(cands, thread_count=4)
    assert len(features_df) > 0.5 * len(cands)

    cols = [c for c in DEFAULT_FEATURE_COLUMNS if c in features_df.columns]
    clf = fdr.HipBinaryClassifier(test_size=0.001, batch_size=5000, learning_rate=0.001, epochs=10,
                                  experimental_hyperparameter_tuning=True, random_state=3)
    res = fdr.perform_fdr(clf, cols, features_df[features_df["decoy"] == 0].copy(),
                          features_df[features_df["decoy"] == 1].copy(), competitive=True, group_channels=True,
                          df_fragments=fragments_df, dia_cycle=dia.cycle, random_state=4)
    hits = res[(res["qval"] <= 0.01)]
    n_decoy = int((hits["decoy"] == 1).sum())
    ids = hits[hits["decoy"] == 0]
    found = planted[ids["precursor_idx"].to_numpy()]
    n_planted = int(planted.sum())
    # most planted peptides are identified, the rest of the accepted list is at the 1 % level
    assert found.sum() >= 0.8 * n_planted, (int(found.sum()), n_planted)
    assert (~found).sum() <= 0.03 * len(ids) + 2, (int((~found).sum()), len(ids))
    assert n_decoy <= 0.02 * len(ids) + 2
    # one row per elution group and channel after the competition
    assert res.groupby(["elution_group_idx", "channel"]).size().max() == 1
    # apex found where it was planted
    apex_rt = dia.rt_values[case.apex_cycle[ids["precursor_idx"].to_numpy()[found]] * L]
    assert np.median(np.abs(ids["rt_observed"].to_numpy()[found] - apex_rt)) < 3.0
