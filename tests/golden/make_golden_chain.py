"""End-to-end golden of the reference chain (north_star: "identical top-1 candidate ranking"), produced by RUNNING
THE REFERENCE in the build container:

    python tests/golden/make_golden_chain.py

TEST INFRASTRUCTURE, same rules as make_golden.py.  On a 1 000-precursor library (500 target / decoy pairs) against a
synthetic Thermo-style run the reference's own stages are chained the way its workflow chains them
(alphadia/workflow/peptidecentric/peptidecentric.py:190-229):

    CandidateSelection.__call__                  search/selection/selection.py
    -> CandidateScoring (collect_candidates / collect_fragments)   search/scoring/scoring.py:394-580
    -> a classifier with FROZEN weights (``predict_proba`` only; the weights are stored in the fixture)
    -> perform_fdr(competitive=True, df_fragments=..., dia_cycle=...)   fdr/fdr.py:25-188
       (get_q_values, FragmentCompetition, keep_best per (elution group, channel), get_q_values)

The frozen classifier is the reference's ``BinaryClassifierLegacyNewBatching`` trained once (dropout 0) on the
reference's feature table; its ``state_dict`` is data of the fixture, so the chain under test does no training and its
result does not depend on a random stream.  Stored: inputs (run, library), every intermediate (candidates, feature
table, probabilities) and the surviving rows with their q-values.
"""

from __future__ import annotations

import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()
ref_shim.install_selection_glue()

import make_golden as mg  # noqa: E402  (DuckDia, case_to_dict, small_case: shared with the stage goldens)
from alphadia.fdr import fdr as ref_fdr  # noqa: E402
from alphadia.fdr.classifiers import BinaryClassifierLegacyNewBatching  # noqa: E402
from alphadia.search.scoring import scoring as ref_scoring  # noqa: E402
from alphadia.search.scoring.config import CandidateScoringConfig  # noqa: E402
from alphadia.search.scoring.output import OutputPsmDF  # noqa: E402
from alphadia.search.selection import selection as ref_sel  # noqa: E402
from alphadia.search.selection.config_df import CandidateSelectionConfig  # noqa: E402

import synthetic as syn  # noqa: E402

# the feature columns the classifier sees: the workflow's list (peptidecentric/utils.py feature_columns, what
# FDRManager is built with at peptidecentric.py:108-116) restricted to what the scoring stage provides
from alphadia.workflow.peptidecentric.utils import feature_columns as WORKFLOW_FEATURE_COLUMNS  # noqa: E402

SELECTION = dict(rt_tolerance=30.0, candidate_count=2, min_size_rt=3, precursor_mz_tolerance=10.0, fragment_mz_tolerance=15.0)
SCORING = dict(mg.SCORING_CONFIGS["handler_default"])
CLASSIFIER = dict(test_size=0.2, batch_size=128, epochs=20, learning_rate=0.001, weight_decay=0.00001,
                  layers=[100, 50, 20, 5], dropout=0.0, metric_interval=1000, random_state=21)


class FrozenClassifier:
    """``fit`` does nothing, ``predict_proba`` is the trained network: the chain has no training step."""

    def __init__(self, trained):
        self.trained = trained

    def fit(self, x, y):
        pass

    def predict_proba(self, x):
        return self.trained.predict_proba(x)


def chain_case() -> syn.SyntheticCase:
    case = mg.small_case(131, n_precursors=1000, n_cycles=400, per_precursor=1, planted_fraction=0.6,
                         few_fragment_fraction=0.0, even_fraction=0.0)
    # an empirical library knows roughly where its peptides elute (the generator plants apexes at random)
    pdf = case.library.precursor_df
    L = case.dia.cycle.shape[1]
    planted = case.apex_cycle >= 0
    rng = np.random.default_rng(5)
    rt_apex = case.dia.rt_values[np.clip(case.apex_cycle, 0, None) * L]
    pdf.loc[planted, "rt_library"] = (rt_apex[planted] + rng.normal(0, 4, planted.sum())).astype(np.float32)
    return case


def main():
    t0 = time.time()
    case = chain_case()
    d = mg.case_to_dict(case)
    d["apex_cycle"] = case.apex_cycle
    dia = mg.DuckDia(case.dia)
    names = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
                 fragment_mz_column="mz_library")

    # ---- stage 1: candidate selection
    scfg = CandidateSelectionConfig()
    scfg.update(SELECTION)
    sel = ref_sel.CandidateSelection(dia, case.library.precursor_df.copy(), case.library.fragment_df.copy(), scfg,
                                     fwhm_rt=scfg.peak_len_rt, fwhm_mobility=scfg.peak_len_mobility, **names)
    cands = sel(thread_count=1)
    d["sel_kernel"] = np.asarray(sel.kernel, dtype=np.float32)
    for k, v in SELECTION.items():
        d["sel_upd_" + k] = np.asarray(v)
    cj = sel.config_jit  # every field the kernels read, as golden_selection stores them
    for k in ("rt_tolerance precursor_mz_tolerance fragment_mz_tolerance candidate_count "
              "top_k_precursors exclude_shared_ions kernel_size f_mobility f_rt center_fraction "
              "min_size_mobility min_size_rt max_size_mobility max_size_rt use_weighted_score "
              "join_close_candidates join_close_candidates_scan_threshold "
              "join_close_candidates_cycle_threshold").split():
        d["sel_cfg_" + k] = np.asarray(getattr(cj, k))
    for k in ("feature_mean", "feature_std", "feature_weight"):
        d["sel_cfg_" + k] = np.asarray(getattr(cj, k), dtype=np.float64)
    for c in ("precursor_idx rank score scan_center scan_start scan_stop frame_center frame_start frame_stop "
              "elution_group_idx decoy").split():
        d["sel_" + c] = cands[c].values
    print(f"selection: {len(cands)} candidates for {cands['precursor_idx'].nunique()} precursors ({time.time() - t0:.0f} s)")

    # ---- stage 2: candidate scoring (the internals run_scoring of make_golden.py drives, on the selected candidates)
    cfg = CandidateScoringConfig()
    cfg.update(SCORING)
    cs = ref_scoring.CandidateScoring(dia_data=dia, precursors_flat=case.library.precursor_df.copy(),
                                      fragments_flat=case.library.fragment_df.copy(), config=cfg, **names)
    fragment_container = cs.assemble_fragments()
    sgc = cs.assemble_score_group_container(cands)
    out = OutputPsmDF(sgc.get_candidate_count(), cs.config.top_k_fragments)
    ref_scoring._process_score_groups(range(len(sgc)), sgc, out, fragment_container, dia.to_jitclass(),
                                      cs.config.to_jitclass(), cs.quadrupole_calibration.jit, False)
    features_df = cs.collect_candidates(cands, out)
    fragments_df = cs.collect_fragments(cands, out)
    for k, v in SCORING.items():
        d["score_cfg_" + k] = np.asarray(v)
    cols = [c for c in WORKFLOW_FEATURE_COLUMNS if c in features_df.columns]
    d["feature_columns"] = np.asarray(cols)
    d["feat_matrix"] = features_df[cols].to_numpy()
    for c in ("precursor_idx", "rank", "elution_group_idx", "decoy", "channel", "rt_observed", "mz_observed"):
        if c in features_df.columns:
            d["feat_" + c] = features_df[c].to_numpy()
    for c in ("precursor_idx", "rank", "mz_observed", "mz_library", "intensity", "correlation", "mass_error", "height"):
        if c in fragments_df.columns:
            d["fragdf_" + c] = fragments_df[c].to_numpy()
    print(f"scoring: {len(features_df)} valid candidates, {len(fragments_df)} fragment rows ({time.time() - t0:.0f} s)")

    # ---- the classifier: trained ONCE here, then frozen (weights are fixture data)
    x = features_df[cols].to_numpy()
    y = features_df["decoy"].to_numpy().astype(np.float64)
    ok = ~np.isnan(x).any(axis=1)  # (perform_fdr drops rows with missing features before anything else, fdr.py:84-85)
    clf = BinaryClassifierLegacyNewBatching(**CLASSIFIER)
    clf.fit(x[ok], y[ok])
    for k, v in clf.network.state_dict().items():
        d["clf/" + k] = v.detach().numpy().copy()
    for k, v in CLASSIFIER.items():
        d["clf_hp_" + k] = np.asarray(v)
    d["clf_input_dim"] = np.asarray(x.shape[1])
    proba_all = np.full(len(x), np.nan, dtype=np.float64)
    proba_all[ok] = clf.predict_proba(x[ok])[:, 1]
    d["clf_proba_all"] = proba_all
    print(f"classifier: trained on {int(ok.sum())} of {len(x)} rows ({int((~ok).sum())} with missing features)")

    # ---- stage 3: FDR with fragment competition and target / decoy competition, classifier frozen
    res = ref_fdr.perform_fdr(FrozenClassifier(clf), cols, features_df[features_df["decoy"] == 0].copy(),
                              features_df[features_df["decoy"] == 1].copy(), competitive=True, group_channels=True,
                              df_fragments=fragments_df.copy(), dia_cycle=case.dia.cycle, random_state=4)
    for c in ("precursor_idx", "rank", "proba", "qval", "decoy", "elution_group_idx"):
        d["fdr_" + c] = res[c].to_numpy()
    hits = res[(res["qval"] <= 0.01) & (res["decoy"] == 0)]
    planted = case.apex_cycle >= 0
    print(f"fdr: {len(res)} rows survive competition, {len(hits)} targets at 1 % FDR, "
          f"{int(planted[hits['precursor_idx'].to_numpy()].sum())} of them planted ({int(planted.sum())} planted)")
    d["caveat"] = np.asarray(mg.CAVEAT + "; selection through the np.fft stand-in of ref_shim.install_selection_glue")
    path = os.path.join(OUT_DIR, "chain.npz")
    np.savez_compressed(path, **d)
    print(path, f"{os.path.getsize(path) / 1e6:.2f} MB in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
