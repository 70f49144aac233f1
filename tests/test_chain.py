"""End to end: "identical top-1 candidate ranking" (north_star) against the reference's own chain.

``tests/golden/chain.npz`` (made by ``tests/golden/make_golden_chain.py``, which RUNS the reference) holds a
1 000-precursor library, a synthetic run and what the reference's stages produce when chained the way its workflow
chains them (peptidecentric.py:190-229): CandidateSelection -> CandidateScoring -> a classifier with frozen weights
(stored in the fixture) -> perform_fdr(competitive=True) with fragment competition -> keep_best -> q-values.

* CPU: the oracle's stages chained the same way reproduce the reference's survivors (pins the oracle chain).
* GPU: the product chain (HipCandidateSelection -> HipCandidateScoring -> HipBinaryClassifier.from_state_dict ->
  alphadia_amd.fdr.perform_fdr, everything through the C ABI) reproduces the surviving (precursor_idx, rank) set -
  every precursor's top-1 candidate - exactly and the q-values to 1e-4.
"""

from __future__ import annotations

import types

import numpy as np
import pandas as pd
import pytest

import helpers as H
from alphadia_amd import _abi
from alphadia_amd.scoring import (
    CandidateScoringConfig,
    OutputPsmDF,
    assemble_candidates,
    collect_candidates,
    collect_fragments,
    fragment_columns,
    pack_assembled,
)
from alphadia_amd.selection import CANDIDATE_COLUMNS

NAMES = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
             fragment_mz_column="mz_library")
PROBA_ATOL = 1e-2
BOX = ["scan_center", "scan_start", "scan_stop", "frame_center", "frame_start", "frame_stop"]
SCORE_KEYS = ("score_grouped top_k_isotopes reference_channel precursor_mz_tolerance fragment_mz_tolerance "
              "exclude_shared_ions quant_window quant_all experimental_xic top_k_fragments").split()


@pytest.fixture(scope="module")
def chain():
    z = np.load(H.golden_path("chain.npz"))
    dia = H.dia_from_npz(z)
    fdf = pd.DataFrame({c: z["frag_" + c] for c in H.FRAG_COLS})
    pdf = pd.DataFrame({c: z["prec_" + c] for c in H.PREC_COLS})
    for c, v in (("proteins", "P"), ("genes", "G"), ("sequence", "PEPTIDEK"), ("mods", ""), ("mod_sites", "")):
        pdf[c] = np.full(len(pdf), v, dtype=object)
    cfg = CandidateScoringConfig()
    cfg.update({k: z["score_cfg_" + k].item() for k in SCORE_KEYS})
    sel = pd.DataFrame({c: z["sel_" + c] for c in CANDIDATE_COLUMNS + ["elution_group_idx", "decoy"]})
    return types.SimpleNamespace(z=z, dia=dia, fdf=fdf, pdf=pdf, cfg=cfg, sel=sel,
                                 cols=[str(c) for c in z["feature_columns"]])


def _sel_cfg(z):
    pre = "sel_cfg_"
    return types.SimpleNamespace(**{k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})


def _pack(pdf):
    pdf = pdf.sort_values("precursor_idx").reset_index(drop=True)
    iso = pdf[[c for c in pdf.columns if c.startswith("i_")]].values
    return _abi.pack_precursors(pdf.precursor_idx.values, pdf.flat_frag_start_idx.values, pdf.flat_frag_stop_idx.values,
                                pdf.charge.values, pdf.rt_library.values, pdf.mobility_library.values,
                                pdf.mz_library.values, iso)


def _state_dict(z) -> dict:
    net = {k[len("clf/"):]: np.asarray(z[k]) for k in z.files if k.startswith("clf/")}
    hp = {k[len("clf_hp_"):]: z[k] for k in z.files if k.startswith("clf_hp_")}
    return dict(_fitted=True, input_dim=int(z["clf_input_dim"]), output_dim=2, test_size=float(hp["test_size"]),
                batch_size=int(hp["batch_size"]), epochs=int(hp["epochs"]), learning_rate=float(hp["learning_rate"]),
                weight_decay=float(hp["weight_decay"]), layers=[int(v) for v in hp["layers"]], dropout=float(hp["dropout"]),
                metric_interval=int(hp["metric_interval"]), metrics={}, network_state_dict=net)


def _check_selection(got: pd.DataFrame, ref: pd.DataFrame):
    """Top-1 boxes identical; the few low-score rows the reference's float32 FFT adds or moves are tolerated
    (tests/test_selection.py says why) and counted."""
    m = got.merge(ref, on=["precursor_idx", "rank"], how="outer", suffixes=("_g", "_e"), indicator=True)
    top = m[m["rank"] == 0]
    assert (top["_merge"] == "both").all(), "a precursor's best candidate exists on one side only"
    for c in BOX:
        assert (top[c + "_g"] == top[c + "_e"]).all(), c
    both = m[m["_merge"] == "both"]
    same = np.ones(len(both), dtype=bool)
    for c in BOX:
        same &= (both[c + "_g"] == both[c + "_e"]).values
    assert same.mean() >= 0.99
    return int((m["_merge"] != "both").sum()), int((~same).sum())


def _check_survivors(res: pd.DataFrame, z, what: str):
    exp = pd.DataFrame({c: z["fdr_" + c] for c in ("precursor_idx", "rank", "proba", "qval", "decoy")})
    got = res[["precursor_idx", "rank", "proba", "qval"]].copy()
    key_e = set(zip(exp["precursor_idx"].tolist(), exp["rank"].tolist(), strict=True))
    key_g = set(zip(got["precursor_idx"].tolist(), got["rank"].tolist(), strict=True))
    assert key_g == key_e, f"{what}: survivors differ (+{sorted(key_g - key_e)[:5]} -{sorted(key_e - key_g)[:5]})"
    m = got.merge(exp, on=["precursor_idx", "rank"], suffixes=("_g", "_e"))
    assert len(m) == len(exp)
    # every precursor's top-1: one surviving row per precursor, the same rank on both sides (the key match above) and
    # the same order of the survivors by probability
    assert m["precursor_idx"].is_unique
    # (the probabilities see the documented float32-typing differences of a golden made under NumPy - mass errors up
    # to 0.15 ppm, tests/test_oracle_golden.py - through the network; the q-values depend on the ORDER only)
    np.testing.assert_allclose(m["proba_g"].to_numpy(), m["proba_e"].to_numpy(), rtol=0, atol=PROBA_ATOL)
    np.testing.assert_allclose(m["qval_g"].to_numpy(), m["qval_e"].to_numpy(), rtol=0, atol=1e-4)
    accepted_e = set(exp.loc[(exp["qval"] <= 0.01) & (exp["decoy"] == 0), "precursor_idx"].tolist())
    res_t = res[(res["qval"] <= 0.01) & (res["decoy"] == 0)]
    assert set(res_t["precursor_idx"].tolist()) == accepted_e, f"{what}: the 1 % identifications differ"
    return len(m), len(accepted_e)


class _Frozen:
    """fit() is a no-op: the chain under test uses the fixture's weights."""

    def __init__(self, predict_proba):
        self.predict_proba = predict_proba

    def fit(self, x, y):
        pass


# --------------------------------------------------------------------------------------------------------------
# CPU: the oracle's stages, chained like the reference's
# --------------------------------------------------------------------------------------------------------------
def _oracle_fdr(oracle, feats: pd.DataFrame, frags: pd.DataFrame, proba: np.ndarray, cycle, fdr_heuristic=0.1):
    """fdr.py:134-178 with the oracle's parts: q-values -> fragment competition below the heuristic FDR ->
    best row per (elution group, channel) -> q-values."""
    from alphadia_amd.fdr import _int_key
    from alphadia_amd.fragcomp import competition_plan
    from oracle import fdr_oracle

    df = feats.copy()
    df["proba"] = proba
    order, q = fdr_oracle.q_values(df["proba"].to_numpy(), df["_decoy"].to_numpy(), _int_key(df, ["precursor_idx"]))
    df = df.iloc[order].copy()
    df["qval"] = q
    start = int(df["qval"].searchsorted(fdr_heuristic, side="left")) or len(df)
    df = df.iloc[:start].copy()
    plan = competition_plan(df["precursor_idx"].values, df["rank"].values, df["mz_observed"].values, df["proba"].values,
                            frags["precursor_idx"].values, frags["rank"].values, cycle)
    valid = oracle.fragcomp(plan.window_start, plan.window_stop, df["rt_observed"].values[plan.rows], plan.frag_start,
                            plan.frag_stop, frags["mz_observed"].values, 3, 15, n_threads=2)
    df = df.iloc[plan.rows[valid]].reset_index(drop=True)
    keep = fdr_oracle.keep_best(df["proba"].to_numpy(), df["elution_group_idx"].to_numpy().astype(np.int64),
                                df["channel"].to_numpy().astype(np.int64))
    df = df[keep].reset_index(drop=True)
    order, q = fdr_oracle.q_values(df["proba"].to_numpy(), df["_decoy"].to_numpy(), _int_key(df, ["precursor_idx"]))
    df = df.iloc[order].copy()
    df["qval"] = q
    return df


def test_oracle_chain_reproduces_the_reference_chain(chain, oracle_lib):
    from oracle import fdr_oracle

    z = chain.z
    cols_lib = fragment_columns(chain.fdf, "mz_library")
    # ---- selection
    arrays = oracle_lib.select(chain.dia, cols_lib, _pack(chain.pdf), _sel_cfg(z), z["sel_kernel"], n_threads=4)
    keep = arrays["score"] > 0
    cands = pd.DataFrame({c: arrays[c][keep] for c in CANDIDATE_COLUMNS})
    cands = cands.merge(chain.pdf[["precursor_idx", "elution_group_idx", "decoy"]], on="precursor_idx", how="left")
    extra, moved = _check_selection(cands, chain.sel)
    # ---- scoring of the oracle's own candidates
    soa = assemble_candidates(cands, chain.pdf, "mz_library", score_grouped=chain.cfg.score_grouped,
                              reference_channel=chain.cfg.reference_channel)
    out = oracle_lib.score(chain.dia, cols_lib, pack_assembled(soa), chain.cfg.to_jitclass(), n_threads=4)
    proto = OutputPsmDF({k: v for k, v in out.items()})
    feats = collect_candidates(cands, proto, chain.pdf, "rt_library", "mobility_library", "mz_library")
    frags = collect_fragments(proto, chain.pdf)
    # the rows both sides scored carry the reference's features (float32 tolerance of the oracle's Numba typing)
    ref_key = pd.DataFrame({"precursor_idx": z["feat_precursor_idx"], "rank": z["feat_rank"], "row": np.arange(len(z["feat_rank"]))})
    m = feats[["precursor_idx", "rank"]].assign(at=np.arange(len(feats))).merge(ref_key, on=["precursor_idx", "rank"])
    assert len(m) >= 0.98 * len(ref_key)
    from alphadia_amd.scoring import DEFAULT_FEATURE_COLUMNS

    a = feats[chain.cols].to_numpy()[m["at"].to_numpy()].astype(np.float64)
    b = z["feat_matrix"][m["row"].to_numpy()].astype(np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    # tolerances of tests/test_oracle_golden.py (Numba typing vs a golden made under NumPy typing): mass errors in ppm
    # absolute, correlations with an absolute floor, 1e-4 relative otherwise
    for j, name in enumerate(chain.cols):
        f = DEFAULT_FEATURE_COLUMNS.index(name) if name in DEFAULT_FEATURE_COLUMNS else -1
        d = np.abs(a[:, j] - b[:, j])
        d = d[~np.isnan(d)]
        if f in (8, 9, 41, 42, 45):
            assert d.max() <= 0.15, (name, d.max())
            continue
        rel = H.rel_err(a[:, j], b[:, j])
        rel = np.where(np.isnan(rel), 0.0, rel)
        if f in H.CORR_FEATURES or f == 27:  # (27 = difference of two log intensities: same reasoning)
            rel = np.where(np.abs(a[:, j] - b[:, j]) <= 1e-3, 0.0, rel)
        assert rel.max() <= 1e-4, (name, float(rel.max()))
    # ---- frozen classifier, then the FDR stage
    feats = feats.dropna(subset=chain.cols)
    feats = pd.concat([feats[feats["decoy"] == 0], feats[feats["decoy"] == 1]])
    feats["_decoy"] = feats["decoy"].to_numpy().astype(np.float64)
    sd = _state_dict(z)
    dims = [sd["input_dim"], *sd["layers"], 2]
    from alphadia_amd.fdr import HipBinaryClassifier

    clf = HipBinaryClassifier.__new__(HipBinaryClassifier)
    clf.layers, clf.dropout = sd["layers"], 0.0
    clf.from_state_dict(sd)
    params, rm, rv, _ = clf._state
    proba = fdr_oracle.mlp_predict(dims, params, rm, rv, feats[chain.cols].to_numpy().astype(np.float32))[:, 1]
    res = _oracle_fdr(oracle_lib, feats, frags, proba, chain.dia.cycle)
    n, acc = _check_survivors(res, z, "oracle chain")
    print(f"oracle chain: {n} survivors identical, {acc} identifications at 1 %; selection rows not in both: {extra}, "
          f"boxes moved: {moved}")


# --------------------------------------------------------------------------------------------------------------
# GPU: the product chain through the C ABI
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("start", ["hip_selection", "reference_candidates"])
def test_hip_chain_reproduces_the_reference_survivors(chain, start):
    from alphadia_amd import fdr
    from alphadia_amd.scoring import HipCandidateScoring
    from alphadia_amd.selection import CandidateSelectionConfig, HipCandidateSelection

    z = chain.z
    if start == "hip_selection":
        scfg = CandidateSelectionConfig()
        scfg.update({k[len("sel_upd_"):]: z[k].item() for k in z.files if k.startswith("sel_upd_")})
        sel = HipCandidateSelection(chain.dia, chain.pdf.copy(), chain.fdf.copy(), scfg, fwhm_rt=scfg.peak_len_rt,
                                    fwhm_mobility=scfg.peak_len_mobility, **NAMES)
        assert np.array_equal(sel.kernel, z["sel_kernel"])
        cands = sel()
        extra, moved = _check_selection(cands[CANDIDATE_COLUMNS + ["elution_group_idx", "decoy"]], chain.sel)
    else:
        cands, extra, moved = chain.sel.copy(), 0, 0
    scorer = HipCandidateScoring(dia_data=chain.dia, precursors_flat=chain.pdf.copy(), fragments_flat=chain.fdf.copy(),
                                 config=chain.cfg, device=0, **NAMES)
    features_df, fragments_df = scorer(cands, thread_count=4)
    clf = fdr.HipBinaryClassifier()
    clf.from_state_dict(_state_dict(z), load_hyperparameters=True)
    res = fdr.perform_fdr(_Frozen(clf.predict_proba), chain.cols, features_df[features_df["decoy"] == 0].copy(),
                          features_df[features_df["decoy"] == 1].copy(), competitive=True, group_channels=True,
                          df_fragments=fragments_df, dia_cycle=chain.dia.cycle, random_state=4)
    n, acc = _check_survivors(res, z, f"HIP chain from {start}")
    print(f"HIP chain ({start}): {n} survivors identical, {acc} identifications at 1 %; selection rows not in both: "
          f"{extra}, boxes moved: {moved}")
