"""Shared helpers of the test-suite: golden fixture loading and comparisons."""

from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np
import pandas as pd

import synthetic as syn
from alphadia_amd.scoring import (
    CandidateScoringConfig,
    assemble_candidates,
    fragment_columns,
    pack_assembled,
)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FRAG_COLS = ["mz_library", "intensity", "cardinality", "type", "loss_type", "charge", "number", "position"]
PREC_COLS = [
    "elution_group_idx", "precursor_idx", "channel", "decoy", "flat_frag_start_idx",
    "flat_frag_stop_idx", "charge", "rt_library", "mobility_library", "mz_library",
    "i_0", "i_1", "i_2", "i_3",
]
CAND_COLS = [
    "elution_group_idx", "precursor_idx", "rank", "scan_start", "scan_stop", "scan_center",
    "frame_start", "frame_stop", "frame_center", "score",
]
CFG_KEYS = (
    "collect_fragments score_grouped exclude_shared_ions top_k_fragments top_k_isotopes "
    "reference_channel quant_window quant_all precursor_mz_tolerance fragment_mz_tolerance "
    "experimental_xic"
).split()
OUT_NAMES = [
    "valid", "precursor_idx", "rank", "features", "fragment_precursor_idx", "fragment_rank",
    "fragment_mz_library", "fragment_mz", "fragment_mz_observed", "fragment_height",
    "fragment_intensity", "fragment_mass_error", "fragment_correlation", "fragment_position",
    "fragment_number", "fragment_type", "fragment_charge", "fragment_loss_type",
]


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name)


def dia_from_npz(z) -> syn.AlphaRawArrays:
    return syn.AlphaRawArrays(
        cycle=z["dia_cycle"],
        rt_values=z["dia_rt_values"],
        peak_start_idx_list=z["dia_peak_start"],
        peak_stop_idx_list=z["dia_peak_stop"],
        mz_values=z["dia_mz"],
        intensity_values=z["dia_intensity"],
        mobility_values=z["dia_mobility_values"],
    )


def load_scoring_golden(name: str):
    z = np.load(golden_path(f"scoring_{name}.npz"))
    dia = dia_from_npz(z)
    fragment_df = pd.DataFrame({c: z["frag_" + c] for c in FRAG_COLS})
    precursor_df = pd.DataFrame({c: z["prec_" + c] for c in PREC_COLS})
    n = len(precursor_df)
    for c, v in (("proteins", "P"), ("genes", "G"), ("sequence", "PEPTIDEK"), ("mods", ""), ("mod_sites", "")):
        precursor_df[c] = np.full(n, v, dtype=object)
    candidates_df = pd.DataFrame({c: z["cand_" + c] for c in CAND_COLS})
    cfg = CandidateScoringConfig()
    upd = {}
    for k in CFG_KEYS:
        v = z["cfg_" + k].item()
        upd[k] = v
    cfg.update(upd)
    if "cfg_quadrupole_sigma" in z.files:  # a fitted quadrupole calibration (golden "fitted_quadrupole")
        cfg.quadrupole_sigma = tuple(float(x) for x in z["cfg_quadrupole_sigma"])
        cfg.quadrupole_delta_mu = tuple(float(x) for x in z["cfg_quadrupole_delta_mu"])
    expected = {n_: z["out_" + n_] for n_ in OUT_NAMES}
    return SimpleNamespace(
        dia=dia,
        library=syn.SyntheticLibrary(precursor_df, fragment_df),
        candidates_df=candidates_df,
        config=cfg,
        expected=expected,
        z=z,
    )


def soa_for(case_like, cfg: CandidateScoringConfig, precursor_mz_column="mz_library"):
    return assemble_candidates(
        case_like.candidates_df,
        case_like.library.precursor_df,
        precursor_mz_column,
        score_grouped=cfg.score_grouped,
        reference_channel=cfg.reference_channel,
    )


def oracle_score(oracle, case_like, cfg: CandidateScoringConfig, n_threads=1, soa=None, with_stats=False):
    soa = soa if soa is not None else soa_for(case_like, cfg)
    cols = fragment_columns(case_like.library.fragment_df, "mz_library")
    return oracle.score(
        case_like.dia, cols, pack_assembled(soa), cfg.to_jitclass(), n_threads=n_threads,
        with_stats=with_stats,
    ), soa


# correlation-type features (differences of nearly equal sums: a relative bound means nothing near zero):
# isotope correlations 15/16, fragment-vs-library correlations 18/19, scan correlations 29/30, the
# profile correlations 31-34 and 36.  Only these get an absolute floor next to the relative tolerance.
CORR_FEATURES = (15, 16, 18, 19, 29, 30, 31, 32, 33, 34, 36)


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    err = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)
    err = np.where(both_nan, 0.0, err)
    err = np.where(np.isnan(err), np.inf, err)
    return err
