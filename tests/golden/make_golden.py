"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE.  Run in the build container only (``/root/reference`` is
mounted there):

    python tests/golden/make_golden.py

It imports the reference's own modules through ``ref_shim`` (which stubs the
missing third-party packages numba / alphatims / alpharaw, nothing else), feeds
them synthetic inputs from ``tests/synthetic.py`` and stores inputs AND
outputs as small ``.npz`` fixtures, so that the tests need neither the
reference nor bit-reproducible random streams on the GPU box.

Shim caveats (NumPy-2 promotion, pairwise sums, argsort ties) are listed in
``ref_shim.py`` and repeated in every fixture's ``caveat`` field.
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
# fixtures go to tests/golden/ unless --out DIR is given (tests/test_golden_regenerate.py writes to a temp dir)
OUT_DIR = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))  # tests/: the synthetic generators
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()

from alphadia.fragcomp.fragcomp import FragmentCompetition  # noqa: E402
from alphadia.search.jitclasses.alpharaw_jit import AlphaRawJIT  # noqa: E402
from alphadia.search.scoring import scoring as ref_scoring  # noqa: E402
from alphadia.search.scoring.config import CandidateScoringConfig  # noqa: E402
from alphadia.search.scoring.output import OutputPsmDF  # noqa: E402

import synthetic as syn  # noqa: E402

CAVEAT = (
    "reference executed as pure Python under a numba stub with NumPy "
    + np.__version__
    + ": float32-with-python-float stays float32 (Numba: float64), np.sum is "
    "pairwise (Numba: sequential), argsort tie order may differ"
)


def to_jit(dia: syn.AlphaRawArrays) -> AlphaRawJIT:
    return AlphaRawJIT(
        dia.cycle,
        dia.rt_values,
        dia.mobility_values,
        dia.zeroth_frame,
        dia.max_mz_value,
        dia.min_mz_value,
        np.float32(dia.cycle[0, 1:, :, 1].max()),
        np.float32(dia.cycle[0, 1:, :, 0].min()),
        dia.precursor_cycle_max_index,
        dia.peak_start_idx_list,
        dia.peak_stop_idx_list,
        dia.mz_values,
        dia.intensity_values,
        dia.scan_max_index,
        dia.frame_max_index,
    )


class DuckDia:
    def __init__(self, dia):
        self._dia = dia
        self.cycle = dia.cycle
        self._jit = to_jit(dia)

    def to_jitclass(self):
        return self._jit


def dia_to_dict(dia: syn.AlphaRawArrays) -> dict:
    return {
        "dia_cycle": dia.cycle,
        "dia_rt_values": dia.rt_values,
        "dia_mobility_values": dia.mobility_values,
        "dia_peak_start": dia.peak_start_idx_list,
        "dia_peak_stop": dia.peak_stop_idx_list,
        "dia_mz": dia.mz_values,
        "dia_intensity": dia.intensity_values,
    }


FRAG_COLS = [
    "mz_library",
    "intensity",
    "cardinality",
    "type",
    "loss_type",
    "charge",
    "number",
    "position",
]
PREC_NUM_COLS = [
    "elution_group_idx",
    "precursor_idx",
    "channel",
    "decoy",
    "flat_frag_start_idx",
    "flat_frag_stop_idx",
    "charge",
    "rt_library",
    "mobility_library",
    "mz_library",
    "i_0",
    "i_1",
    "i_2",
    "i_3",
]
CAND_COLS = [
    "elution_group_idx",
    "precursor_idx",
    "rank",
    "scan_start",
    "scan_stop",
    "scan_center",
    "frame_start",
    "frame_stop",
    "frame_center",
    "score",
]


# parameters of the fitted quadrupole calibration of the "fitted_quadrupole" golden (what SimpleQuadrupole.fit
# leaves in jit.sigma / jit.delta_mu, quadrupole.py:195-204)
FITTED_QUADRUPOLE = dict(sigma=(0.35, 0.12), delta_mu=(0.4, -0.25))


def run_scoring(case: syn.SyntheticCase, cfg_updates: dict, quadrupole: dict | None = None):
    cfg = CandidateScoringConfig()
    cfg.update(cfg_updates)
    dia = DuckDia(case.dia)
    cs = ref_scoring.CandidateScoring(
        dia_data=dia,
        precursors_flat=case.library.precursor_df.copy(),
        fragments_flat=case.library.fragment_df.copy(),
        rt_column="rt_library",
        mobility_column="mobility_library",
        precursor_mz_column="mz_library",
        fragment_mz_column="mz_library",
        config=cfg,
    )
    cands = case.candidates_df.copy()
    if quadrupole is not None:  # a fitted calibration: sigma / delta_mu as SimpleQuadrupole.fit sets them
        cs.quadrupole_calibration.jit.sigma = np.array(quadrupole["sigma"], dtype=np.float64)
        cs.quadrupole_calibration.jit.delta_mu = np.array(quadrupole["delta_mu"], dtype=np.float64)
    fragment_container = cs.assemble_fragments()
    sgc = cs.assemble_score_group_container(cands)
    n = sgc.get_candidate_count()
    out = OutputPsmDF(n, cs.config.top_k_fragments)
    ref_scoring._process_score_groups(
        range(len(sgc)),
        sgc,
        out,
        fragment_container,
        dia.to_jitclass(),
        cs.config.to_jitclass(),
        cs.quadrupole_calibration.jit,
        False,
    )
    features_df = cs.collect_candidates(cands, out)
    fragments_df = cs.collect_fragments(cands, out)
    # order of rows in `out` = order of candidates inside the score group container
    order_pidx = np.array(
        [c.precursor_idx for sg in sgc.score_groups for c in sg.candidates], dtype=np.uint32
    )
    order_rank = np.array(
        [c.rank for sg in sgc.score_groups for c in sg.candidates], dtype=np.uint8
    )
    return out, features_df, fragments_df, order_pidx, order_rank, cs.config


def out_to_dict(out: OutputPsmDF) -> dict:
    names = [
        "valid",
        "precursor_idx",
        "rank",
        "features",
        "fragment_precursor_idx",
        "fragment_rank",
        "fragment_mz_library",
        "fragment_mz",
        "fragment_mz_observed",
        "fragment_height",
        "fragment_intensity",
        "fragment_mass_error",
        "fragment_correlation",
        "fragment_position",
        "fragment_number",
        "fragment_type",
        "fragment_charge",
        "fragment_loss_type",
    ]
    return {"out_" + n: np.asarray(getattr(out, n)) for n in names}


def case_to_dict(case: syn.SyntheticCase) -> dict:
    d = dia_to_dict(case.dia)
    for c in FRAG_COLS:
        d["frag_" + c] = case.library.fragment_df[c].values
    for c in PREC_NUM_COLS:
        d["prec_" + c] = case.library.precursor_df[c].values
    for c in CAND_COLS:
        d["cand_" + c] = case.candidates_df[c].values
    return d


def small_case(config_id: int, **kw) -> syn.SyntheticCase:
    args = dict(
        n_precursors=300,
        n_cycles=60,
        config_id=config_id,
        per_precursor=2,
        n_ms2=8,
        ms1_peaks=400,
        ms2_peaks=150,
        mz_lo=400,
        mz_hi=480,
        frag_mz_lo=200,
        frag_mz_hi=350,
        ms1_mz_range=(395, 500),
        ms2_mz_range=(195, 355),
        few_fragment_fraction=0.05,
        even_fraction=0.3,
        planted_fraction=0.5,
        threads=1,
    )
    args.update(kw)
    return syn.make_case(**args)


SCORING_CONFIGS = {
    # what ClassicExtractionHandler passes (extraction_handler.py:370-376,400-409,460-468
    # with default.yaml:158-199 and target tolerances)
    "handler_default": dict(
        score_grouped=False,
        top_k_isotopes=3,
        reference_channel=-1,
        precursor_mz_tolerance=10,
        fragment_mz_tolerance=15,
        exclude_shared_ions=True,
        quant_window=3,
        quant_all=True,
        experimental_xic=True,
        top_k_fragments=12,
    ),
    # the handler's settings with a FITTED quadrupole calibration (FITTED_QUADRUPOLE above)
    "fitted_quadrupole": dict(
        score_grouped=False,
        top_k_isotopes=3,
        reference_channel=-1,
        precursor_mz_tolerance=10,
        fragment_mz_tolerance=15,
        exclude_shared_ions=True,
        quant_window=3,
        quant_all=True,
        experimental_xic=True,
        top_k_fragments=12,
    ),
    # CandidateScoringConfig() defaults (config.py:73-85); used by multiplex requant
    "class_default": dict(),
    # top-k filtering active + narrow tolerances
    "topk6": dict(
        top_k_isotopes=2,
        precursor_mz_tolerance=5,
        fragment_mz_tolerance=7,
        quant_window=2,
        quant_all=True,
        experimental_xic=True,
        top_k_fragments=6,
    ),
}


# libraries with 5..40 fragments per precursor (more than the 12 of a predicted library and than
# the 16 lanes of the register kernels): transfer-library requantification scores every fragment
# (transfer_library_requantification_handler.py:117-124, top_k_fragments = 9999), and a plain top-k
# of 24 with the class defaults exercises the K > 16 branch of the K x K contraction
MANYFRAG_CONFIGS = {
    "manyfrag": dict(SCORING_CONFIGS["handler_default"], top_k_fragments=9999),
    "manyfrag_class": dict(top_k_fragments=24),
}


def golden_scoring(which=None):
    configs = dict(SCORING_CONFIGS, **MANYFRAG_CONFIGS)
    for name, upd in configs.items():
        if which is not None and name not in which:
            continue
        if name in MANYFRAG_CONFIGS:
            case = small_case(131, n_precursors=160, k_fragments=(5, 40), frag_mz_hi=500, ms2_mz_range=(195, 505))
        else:
            case = small_case(101)
        if name == "topk6":
            # give some fragments cardinality 2 so exclude_shared_ions drops them
            rng = np.random.default_rng(7)
            card = case.library.fragment_df["cardinality"].values.copy()
            card[rng.random(card.size) < 0.15] = 2
            case.library.fragment_df["cardinality"] = card
        out, fdf, frdf, opidx, orank, cfg = run_scoring(case, upd, FITTED_QUADRUPOLE if name == "fitted_quadrupole" else None)
        d = case_to_dict(case)
        od = out_to_dict(out)
        if name == "fitted_quadrupole":
            d["cfg_quadrupole_sigma"] = np.asarray(FITTED_QUADRUPOLE["sigma"], dtype=np.float64)
            d["cfg_quadrupole_delta_mu"] = np.asarray(FITTED_QUADRUPOLE["delta_mu"], dtype=np.float64)
        if name in MANYFRAG_CONFIGS:
            # the reference allocates top_k_fragments (9999) columns; keep those a library slice can fill
            lens = (case.library.precursor_df["flat_frag_stop_idx"].values.astype(np.int64)
                    - case.library.precursor_df["flat_frag_start_idx"].values.astype(np.int64))
            width = int(min(int(cfg.top_k_fragments), lens.max()))
            for k, v in od.items():
                if v.ndim == 2 and k != "out_features":
                    assert not v[:, width:].any(), k
                    od[k] = np.ascontiguousarray(v[:, :width])
        d.update(od)
        d["order_precursor_idx"] = opidx
        d["order_rank"] = orank
        cfgj = cfg.to_jitclass()
        for k in (
            "collect_fragments score_grouped exclude_shared_ions top_k_fragments top_k_isotopes "
            "reference_channel quant_window quant_all precursor_mz_tolerance "
            "fragment_mz_tolerance experimental_xic"
        ).split():
            d["cfg_" + k] = np.asarray(getattr(cfgj, k))
        d["features_df_columns"] = np.array(list(fdf.columns), dtype="U")
        d["fragments_df_columns"] = np.array(list(frdf.columns), dtype="U")
        d["features_df_precursor_idx"] = fdf["precursor_idx"].values
        d["features_df_rank"] = fdf["rank"].values
        d["features_df_delta_rt"] = fdf["delta_rt"].values
        d["fragments_df_precursor_idx"] = frdf["precursor_idx"].values
        d["fragments_df_mz_observed"] = frdf["mz_observed"].values
        d["fragments_df_n"] = np.asarray(len(frdf))
        d["caveat"] = np.asarray(CAVEAT)
        path = os.path.join(OUT_DIR, f"scoring_{name}.npz")
        np.savez_compressed(path, **d)
        v = np.asarray(out.valid)
        print(
            f"{path}: {v.sum()}/{len(v)} valid, nan rows "
            f"{np.isnan(out.features[v]).any(axis=1).sum()}, {os.path.getsize(path)/1e6:.2f} MB"
        )


def multiplex_case(n_base: int = 120, n_cycles: int = 60):
    """Config 5 in small: 4 label channels per elution group, y-ions shifted, b-ions shared."""
    seed = syn.BASE_SEED + 105
    cycle = syn.make_cycle(n_ms2=8, mz_lo=400, mz_hi=480)
    base = syn.make_library(n_base, seed, mz_lo=400, mz_hi=465, rt_max=n_cycles * 1.5,
                            frag_mz_lo=200, frag_mz_hi=340)
    lib = syn.multiplex_library(base, channels=(0, 4, 8, 12), seed=seed)
    planted = syn.plant_peptides(lib, cycle, n_cycles, seed, fraction=0.6)
    dia = syn.make_thermo_run(n_cycles, seed, cycle=cycle, ms1_peaks=400, ms2_peaks=150,
                              planted=planted, threads=1, ms1_mz_range=(395, 500),
                              ms2_mz_range=(195, 355))
    # reference-channel candidates (one per target precursor of channel 0) ...
    pdf = lib.precursor_df
    ref = pdf[(pdf["channel"] == 0) & (pdf["decoy"] == 0)]
    ref_lib = syn.SyntheticLibrary(ref.reset_index(drop=True), lib.fragment_df)
    apex = planted.apex_cycle[ref.index.values]
    cands = syn.make_candidates(ref_lib, n_cycles, cycle.shape[1], seed, per_precursor=1,
                                apex_cycle=apex, even_fraction=0.3)
    cands["proba"] = np.linspace(0.0, 0.5, len(cands)).astype(np.float32)
    cands["decoy"] = np.uint8(0)
    # ... expanded to all channels by the reference (scoring/utils.py:114-200), as the
    # multiplexing handler does (multiplexing_requantification_handler.py:94-131)
    from alphadia.search.scoring.utils import multiplex_candidates

    mult = multiplex_candidates(cands, pdf, channels=[0, 4, 8, 12])
    mult["rank"] = np.uint8(0)
    # a few elution groups lose their reference channel: ScoreGroup.process skips them
    # (score_group.py:50-65)
    egs = np.unique(mult["elution_group_idx"].values)
    lost = egs[::7]
    mult = mult[~((mult["channel"] == 0) & mult["elution_group_idx"].isin(lost))].reset_index(drop=True)
    for c in ("scan_start", "scan_stop", "scan_center", "frame_start", "frame_stop", "frame_center"):
        mult[c] = mult[c].astype(np.int64)
    mult["elution_group_idx"] = mult["elution_group_idx"].astype(np.uint32)
    mult["precursor_idx"] = mult["precursor_idx"].astype(np.uint32)
    return syn.SyntheticCase(dia, lib, mult[CAND_COLS].copy(), planted.apex_cycle)


def golden_host_helpers():
    """Inputs and outputs of the reference's DataFrame helpers that the host layer restates in
    numpy: ``multiplex_candidates`` (scoring/utils.py:114-200, three settings) and
    ``calculate_score_groups`` (scoring/utils.py:269-410, grouped and not)."""
    from alphadia.search.scoring.utils import calculate_score_groups, multiplex_candidates

    rng = np.random.default_rng(20260928)
    n_eg, channels_all = 150, np.array([0, 4, 8, 12])
    rows = []
    for eg in range(n_eg):
        for decoy in (0, 1):
            for ch in channels_all[: rng.integers(2, 5)]:
                rows.append((eg, decoy, int(ch)))
    lib = pd.DataFrame(rows, columns=["elution_group_idx", "decoy", "channel"])
    lib = lib.sample(frac=1.0, random_state=3).reset_index(drop=True)  # library order is arbitrary
    lib["precursor_idx"] = rng.permutation(len(lib)).astype(np.uint32)
    lib["elution_group_idx"] = lib["elution_group_idx"].astype(np.uint32)
    lib["decoy"] = lib["decoy"].astype(np.uint8)
    lib["channel"] = lib["channel"].astype(np.uint32)
    # the columns precursors_flat_schema asks for (validation/schemas.py:11-32)
    for c in ("flat_frag_start_idx", "flat_frag_stop_idx"):
        lib[c] = np.zeros(len(lib), dtype=np.uint32)
    lib["charge"] = np.full(len(lib), 2, dtype=np.uint8)
    for c in ("rt_library", "mobility_library", "mz_library"):
        lib[c] = rng.random(len(lib)).astype(np.float32)
    lib["proteins"] = np.full(len(lib), "P", dtype=object)
    lib["genes"] = np.full(len(lib), "G", dtype=object)
    # candidates: several per elution group (different precursors, ranks), ties in proba, some groups absent
    pick = lib[lib["elution_group_idx"] % 5 != 0].sample(frac=0.7, random_state=9)
    cand = pick[["elution_group_idx", "precursor_idx", "decoy", "channel"]].copy().reset_index(drop=True)
    cand = pd.concat([cand, cand.iloc[::3]], ignore_index=True)
    cand["channel"] = cand["channel"].astype(np.uint8)  # candidates_schema: uint8
    cand["rank"] = rng.integers(0, 3, len(cand)).astype(np.uint8)
    cand["proba"] = np.round(rng.random(len(cand)), 1).astype(np.float32)  # many ties
    for c in ("scan_start", "scan_stop", "scan_center", "frame_start", "frame_stop", "frame_center"):
        cand[c] = rng.integers(0, 5000, len(cand)).astype(np.int64)
    cand["score"] = rng.random(len(cand)).astype(np.float32)
    d = {"lib_" + c: lib[c].values for c in lib.columns if lib[c].dtype != object}
    d.update({"cand_" + c: cand[c].values for c in cand.columns})
    settings = {"default": dict(), "with_decoys": dict(remove_decoys=False), "two_channels": dict(channels=[0, 8])}
    for name, kw in settings.items():
        out = multiplex_candidates(cand.copy(), lib.copy(), **kw)
        d[f"mult_{name}_columns"] = np.array(list(out.columns), dtype="U")
        for c in out.columns:
            d[f"mult_{name}_{c}"] = out[c].values
        print(f"multiplex_candidates[{name}]: {len(out)} rows")
    for name, grouped in (("plain", False), ("grouped", True)):
        out = calculate_score_groups(cand.copy(), group_channels=grouped)
        for c in ("precursor_idx", "rank", "score_group_idx"):
            d[f"groups_{name}_{c}"] = out[c].values
    path = os.path.join(OUT_DIR, "host_helpers.npz")
    np.savez_compressed(path, **d)
    print(f"{path}: {os.path.getsize(path)/1e3:.0f} kB")


def _spectrum_tables(rng, n_cycles, n_windows, prefix, ms1_every_cycle=True, peaks=(3, 9)):
    """An AlphaRaw-style spectrum_df / peak_df pair: ``prefix`` non-DIA spectra, then cycles of one MS1
    + ``n_windows`` MS2 spectra (rt in minutes, as alpharaw delivers it)."""
    edges = np.linspace(400.0, 400.0 + 10.0 * n_windows, n_windows + 1)
    lower, upper, level, prec = [], [], [], []
    for _ in range(prefix):  # calibration / settling scans before the method starts cycling
        lo = float(rng.uniform(300, 900))
        lower.append(lo)
        upper.append(lo + float(rng.uniform(1, 30)))
        level.append(2)
        prec.append(lo + 1)
    for c in range(n_cycles):
        if ms1_every_cycle or c % 3 != 1:
            lower.append(-1.0)
            upper.append(-1.0)
            level.append(1)
            prec.append(-1.0)
        for w in range(n_windows):
            lower.append(edges[w])
            upper.append(edges[w + 1])
            level.append(2)
            prec.append(0.5 * (edges[w] + edges[w + 1]))
    n = len(lower)
    counts = rng.integers(peaks[0], peaks[1], n)
    stop = np.cumsum(counts)
    spectrum_df = pd.DataFrame({
        "spec_idx": np.arange(n, dtype=np.int64), "rt": np.cumsum(rng.uniform(0.0008, 0.0012, n)),
        "ms_level": np.array(level, dtype=np.int64), "precursor_mz": np.array(prec, dtype=np.float64),
        "isolation_lower_mz": np.array(lower, dtype=np.float64), "isolation_upper_mz": np.array(upper, dtype=np.float64),
        "peak_start_idx": (stop - counts).astype(np.int64), "peak_stop_idx": stop.astype(np.int64),
    })
    mz = np.concatenate([np.sort(rng.uniform(150, 1500, c)) for c in counts])
    peak_df = pd.DataFrame({"mz": mz, "intensity": rng.lognormal(5, 1, mz.size)})
    return spectrum_df, peak_df


def golden_staging():
    """f-4: what the reference derives from a spectrum table before anything is scored -
    ``determine_dia_cycle`` (raw_data/dia_cycle.py:18-82) and ``AlphaRaw._preprocess_raw_data``
    (raw_data/alpharaw_wrapper.py:72-117) - on tables with a non-DIA prefix and with an MS1 that
    does not follow the cycle."""
    import types as _types

    from alphadia.raw_data.alpharaw_wrapper import AlphaRaw
    from alphadia.raw_data.dia_cycle import determine_dia_cycle

    rng = np.random.default_rng(77)
    cases = {
        "plain": dict(n_cycles=40, n_windows=12, prefix=0),
        "prefix": dict(n_cycles=35, n_windows=9, prefix=7),
        "irregular_ms1": dict(n_cycles=45, n_windows=6, prefix=0, ms1_every_cycle=False),
    }
    d = {}
    for name, kw in cases.items():
        sdf, pdf = _spectrum_tables(rng, **kw)
        for c in sdf.columns:
            d[f"{name}_spec_{c}"] = sdf[c].values
        for c in pdf.columns:
            d[f"{name}_peak_{c}"] = pdf[c].values
        obj = _types.SimpleNamespace(spectrum_df=sdf.copy(), peak_df=pdf.copy(), has_ms1=True)
        obj._is_ms1_dia = _types.MethodType(AlphaRaw._is_ms1_dia, obj)
        AlphaRaw._preprocess_raw_data(obj)
        cycle, start, length = determine_dia_cycle(obj.spectrum_df if not obj.has_ms1 else sdf)
        d[f"{name}_cycle_direct"] = cycle
        for attr in ("cycle", "rt_values", "_peak_start_idx_list", "_peak_stop_idx_list", "_mz_values", "_intensity_values"):
            d[f"{name}_{attr.lstrip('_')}"] = np.asarray(getattr(obj, attr))
        for attr in ("_cycle_start", "_cycle_length", "_precursor_cycle_max_index", "_max_mz_value", "_min_mz_value",
                     "_quad_max_mz_value", "_quad_min_mz_value", "frame_max_index", "has_ms1"):
            d[f"{name}_{attr.lstrip('_')}"] = np.asarray(getattr(obj, attr))
        print(name, "cycle length", obj._cycle_length, "start", obj._cycle_start, "has_ms1", obj.has_ms1,
              "spectra", len(obj.rt_values))
    path = os.path.join(OUT_DIR, "staging.npz")
    np.savez_compressed(path, **d)
    print(f"{path}: {os.path.getsize(path)/1e3:.0f} kB")


def golden_multiplex():
    case = multiplex_case()
    upd = dict(score_grouped=True, exclude_shared_ions=True, reference_channel=0, experimental_xic=True)
    out, fdf, frdf, opidx, orank, cfg = run_scoring(case, upd)
    d = case_to_dict(case)
    d.update(out_to_dict(out))
    d["order_precursor_idx"] = opidx
    d["order_rank"] = orank
    cfgj = cfg.to_jitclass()
    for k in (
        "collect_fragments score_grouped exclude_shared_ions top_k_fragments top_k_isotopes "
        "reference_channel quant_window quant_all precursor_mz_tolerance "
        "fragment_mz_tolerance experimental_xic"
    ).split():
        d["cfg_" + k] = np.asarray(getattr(cfgj, k))
    d["features_df_columns"] = np.array(list(fdf.columns), dtype="U")
    d["fragments_df_columns"] = np.array(list(frdf.columns), dtype="U")
    d["features_df_precursor_idx"] = fdf["precursor_idx"].values
    d["features_df_rank"] = fdf["rank"].values
    d["features_df_delta_rt"] = fdf["delta_rt"].values
    d["fragments_df_precursor_idx"] = frdf["precursor_idx"].values
    d["fragments_df_mz_observed"] = frdf["mz_observed"].values
    d["fragments_df_n"] = np.asarray(len(frdf))
    d["caveat"] = np.asarray(CAVEAT)
    path = os.path.join(OUT_DIR, "scoring_multiplex.npz")
    np.savez_compressed(path, **d)
    v = np.asarray(out.valid)
    print(f"{path}: {v.sum()}/{len(v)} valid, {os.path.getsize(path)/1e6:.2f} MB")


def edges_case():
    """Hand-made edge cases on top of a small run: precursors at the edge of the isolation scheme,
    windows of 1-4 cycles, boxes at the first / last cycle, libraries of 0-4 fragments,
    fragments outside the m/z range of the run, precursors on a window boundary."""
    case = small_case(103, n_precursors=60, per_precursor=1, few_fragment_fraction=0.0,
                      even_fraction=0.0, planted_fraction=0.8)
    pdf = case.library.precursor_df.copy()
    fdf = case.library.fragment_df.copy()
    L = case.dia.cycle_len
    n_cycles = case.dia.n_cycles
    cands = case.candidates_df.copy().reset_index(drop=True)
    n = len(pdf)
    mz = pdf["mz_library"].values.copy()
    # (a precursor outside EVERY window makes the reference raise inside
    # quadrupole_transfer_function_single; that case is implementation-defined)
    mz[0] = 400.2   # isotope range starts below the first window
    mz[1] = 479.4   # ... and ends above the last one
    mz[2] = 410.0   # exactly on a window boundary (windows are 10 Th wide from 400)
    mz[3] = 409.6   # isotopes straddle the boundary -> two observations
    pdf["mz_library"] = mz.astype(np.float32)
    # fragment counts 0..4 for precursors 4..8
    start = pdf["flat_frag_start_idx"].values.astype(np.int64).copy()
    stop = pdf["flat_frag_stop_idx"].values.astype(np.int64).copy()
    for j, cnt in zip(range(4, 9), (0, 1, 2, 3, 4)):
        stop[j] = start[j] + cnt
    pdf["flat_frag_stop_idx"] = stop.astype(np.uint32)
    # precursor 9: all fragments outside the acquired m/z range
    f = fdf["mz_library"].values.copy()
    f[start[9]:stop[9]] = np.linspace(50.0, 60.0, stop[9] - start[9], dtype=np.float32)
    # precursor 10: half of them above it
    f[start[10]:start[10] + 6] = np.linspace(2000.0, 2100.0, 6, dtype=np.float32)
    fdf["mz_library"] = f.astype(np.float32)

    def box(i, c, h_lo, h_hi):
        cands.loc[i, "frame_start"] = (c - h_lo) * L
        cands.loc[i, "frame_stop"] = (c + h_hi + 1) * L
        cands.loc[i, "frame_center"] = c * L

    box(11, 20, 0, 0)   # one cycle
    box(12, 20, 0, 1)   # two cycles
    box(13, 20, 1, 1)   # three
    box(14, 20, 1, 2)   # four
    box(15, 3, 3, 3)    # touches the first cycle
    box(16, n_cycles - 4, 3, 3)  # touches the last cycle
    box(17, 30, 14, 14)  # 29 cycles
    box(18, 30, 16, 17)  # 34 cycles: beyond the register kernels
    case.library.precursor_df = pdf
    case.library.fragment_df = fdf
    case.candidates_df = cands
    return case


def golden_edges():
    case = edges_case()
    out, fdf, frdf, opidx, orank, cfg = run_scoring(case, SCORING_CONFIGS["handler_default"])
    d = case_to_dict(case)
    d.update(out_to_dict(out))
    d["order_precursor_idx"] = opidx
    d["order_rank"] = orank
    cfgj = cfg.to_jitclass()
    for k in (
        "collect_fragments score_grouped exclude_shared_ions top_k_fragments top_k_isotopes "
        "reference_channel quant_window quant_all precursor_mz_tolerance "
        "fragment_mz_tolerance experimental_xic"
    ).split():
        d["cfg_" + k] = np.asarray(getattr(cfgj, k))
    d["features_df_columns"] = np.array(list(fdf.columns), dtype="U")
    d["fragments_df_columns"] = np.array(list(frdf.columns), dtype="U")
    d["features_df_precursor_idx"] = fdf["precursor_idx"].values
    d["features_df_rank"] = fdf["rank"].values
    d["features_df_delta_rt"] = fdf["delta_rt"].values
    d["fragments_df_precursor_idx"] = frdf["precursor_idx"].values
    d["fragments_df_mz_observed"] = frdf["mz_observed"].values
    d["fragments_df_n"] = np.asarray(len(frdf))
    d["caveat"] = np.asarray(CAVEAT)
    path = os.path.join(OUT_DIR, "scoring_edges.npz")
    np.savez_compressed(path, **d)
    v = np.asarray(out.valid)
    print(f"{path}: {v.sum()}/{len(v)} valid; first 20 valid flags {v[:20].astype(int)}; "
          f"n_obs of 0..3: {np.asarray(out.features)[:4, 17]}")


def golden_selection():
    """Next-row golden: CandidateSelection.__call__ on a small AlphaRaw run.

    Produced with the FFT stand-in of ref_shim.install_selection_glue (np.fft instead of
    rocket_fft/pocketfft; float32 results): scores agree with any other exact convolution to
    ~1e-5 relative, so a candidate whose score ties another one within that margin may swap or
    move by a cycle.  The golden stores the smoothed score row of every precursor as well."""
    import ref_shim

    ref_shim.install_selection_glue()
    from alphadia.search.selection import selection as ref_sel
    from alphadia.search.selection.config_df import CandidateSelectionConfig

    case = small_case(104, n_precursors=240, n_cycles=120, per_precursor=1, planted_fraction=0.7)
    cfgs = {
        "default": dict(rt_tolerance=30.0, candidate_count=3, min_size_rt=3),
        "wide": dict(rt_tolerance=70.0, candidate_count=5, min_size_rt=2, exclude_shared_ions=False,
                     join_close_candidates=False, precursor_mz_tolerance=20.0, fragment_mz_tolerance=30.0),
    }
    d = case_to_dict(case)
    d["caveat"] = np.asarray(CAVEAT)
    for name, upd in cfgs.items():
        cfg = CandidateSelectionConfig()
        cfg.update(upd)
        dia = DuckDia(case.dia)
        cs = ref_sel.CandidateSelection(
            dia, case.library.precursor_df.copy(), case.library.fragment_df.copy(), cfg,
            rt_column="rt_library", mobility_column="mobility_library",
            precursor_mz_column="mz_library", fragment_mz_column="mz_library",
            fwhm_rt=cfg.peak_len_rt, fwhm_mobility=cfg.peak_len_mobility,
        )
        df = cs(thread_count=1)
        d[f"{name}_kernel"] = np.asarray(cs.kernel, dtype=np.float32)
        cj = cs.config_jit
        for k in ("rt_tolerance precursor_mz_tolerance fragment_mz_tolerance candidate_count "
                  "top_k_precursors exclude_shared_ions kernel_size f_mobility f_rt center_fraction "
                  "min_size_mobility min_size_rt max_size_mobility max_size_rt use_weighted_score "
                  "join_close_candidates join_close_candidates_scan_threshold "
                  "join_close_candidates_cycle_threshold").split():
            d[f"{name}_cfg_{k}"] = np.asarray(getattr(cj, k))
        for k in ("feature_mean", "feature_std", "feature_weight"):
            d[f"{name}_cfg_{k}"] = np.asarray(getattr(cj, k), dtype=np.float64)
        for c in ("precursor_idx rank score scan_center scan_start scan_stop frame_center frame_start "
                  "frame_stop elution_group_idx decoy").split():
            d[f"{name}_out_{c}"] = df[c].values
        print(name, len(df), "candidates for", df["precursor_idx"].nunique(), "precursors;",
              "ranks", np.bincount(df["rank"].values))
    path = os.path.join(OUT_DIR, "selection.npz")
    np.savez_compressed(path, **d)
    print(path, f"{os.path.getsize(path)/1e6:.2f} MB")


def golden_transpose():
    """Staging-format golden: the reference's `_transpose`
    (alphadia/raw_data/bruker.py:201-280) on a small frame-major event list."""
    from alphadia.raw_data.bruker import _transpose

    rng = np.random.default_rng(17)
    n_push, n_tof = 3000, 700
    counts = rng.poisson(12, n_push)
    counts[rng.random(n_push) < 0.1] = 0  # empty pushes
    push_indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(push_indptr[-1])
    # inside a push the tof indices are ascending and may repeat across pushes
    tof = np.concatenate([np.sort(rng.integers(0, n_tof, c)) for c in counts]).astype(np.uint32)
    values = rng.integers(1, 4000, n).astype(np.uint16)
    push_indices, tof_indptr, new_values = _transpose(tof, push_indptr, n_tof, values)
    path = os.path.join(OUT_DIR, "transpose.npz")
    np.savez_compressed(path, tof_indices=tof, push_indptr=push_indptr, n_tof=np.asarray(n_tof), values=values,
                        out_push_indices=push_indices, out_tof_indptr=tof_indptr, out_values=new_values)
    print(path, n, "events", f"{os.path.getsize(path)/1e6:.2f} MB")


def golden_selection_kats():
    """Known answers of the small selection functions, produced by the reference functions
    (selection/utils.py:49-77,218-280): random inputs -> outputs."""
    import ref_shim

    ref_shim.install_selection_glue()
    from alphadia.search.selection.utils import _symetric_limits_1d, find_peaks_1d

    rng = np.random.default_rng(23)
    lim_in, lim_out = [], []
    for _ in range(400):
        n = int(rng.integers(0, 40))
        x = rng.random(n)
        if n and rng.random() < 0.5:  # a real peak shape
            c0 = rng.integers(0, n)
            x = np.exp(-0.5 * ((np.arange(n) - c0) / rng.uniform(1, 6)) ** 2) + 0.05 * rng.random(n)
        center = int(rng.integers(-2, 42))
        f, cf = float(rng.random() * 1.5), float(rng.random())
        mn, mx = int(rng.integers(0, 12)), int(rng.integers(0, 25))
        out = _symetric_limits_1d(x, center, f=f, center_fraction=cf, min_size=mn, max_size=mx)
        pad = np.full(40, np.nan)
        pad[:n] = x
        lim_in.append(np.concatenate([pad, [n, center, f, cf, mn, mx]]))
        lim_out.append(np.asarray(out, dtype=np.int32))
    pk_in, pk_cyc, pk_val, pk_n = [], [], [], []
    for _ in range(200):
        n = int(rng.integers(3, 60))
        row = np.cumsum(rng.normal(size=n))  # random walk: plenty of strict 5-point maxima
        if rng.random() < 0.3:
            row = np.round(row, 1)  # ties
        top_n = int(rng.integers(1, 7))
        scan, cyc, val = find_peaks_1d(np.stack([row, row]), top_n=top_n)
        pad = np.full(60, np.nan)
        pad[:n] = row
        pk_in.append(np.concatenate([pad, [n, top_n]]))
        c = np.full(6, -1, dtype=np.int32)
        v = np.full(6, np.nan)
        c[: len(cyc)] = cyc
        v[: len(val)] = val
        pk_cyc.append(c)
        pk_val.append(v)
        pk_n.append(len(cyc))
    path = os.path.join(OUT_DIR, "selection_kats.npz")
    np.savez_compressed(path, limits_in=np.stack(lim_in), limits_out=np.stack(lim_out), peaks_in=np.stack(pk_in),
                        peaks_cycle=np.stack(pk_cyc), peaks_score=np.stack(pk_val), peaks_n=np.asarray(pk_n))
    print(path, f"{os.path.getsize(path)/1e6:.2f} MB")


def golden_get_dense():
    """G1: AlphaRawJIT.get_dense on hand-picked query lists (incl. overlapping windows)."""
    case = small_case(102, n_precursors=40)
    jit = to_jit(case.dia)
    rng = np.random.default_rng(11)
    L = case.dia.cycle_len
    d = dia_to_dict(case.dia)
    n_cases = 32
    for i in range(n_cases):
        c0 = int(rng.integers(0, 40))
        nc = int(rng.integers(1, 18))
        frame_limits = np.array([[c0 * L, (c0 + nc) * L, 1]], dtype=np.uint64)
        k = int(rng.integers(2, 13))
        ms1 = i % 3 == 0
        if ms1:
            mzq = np.sort(rng.uniform(400, 480, k)).astype(np.float32)
            quad = np.array([[-1.0, -1.0]])
            tol = np.float32(10)
        else:
            mzq = np.sort(rng.uniform(200, 350, k)).astype(np.float32)
            lo = rng.uniform(400, 478)
            quad = np.array([[lo, lo + rng.uniform(0.5, 12.0)]], dtype=np.float32)
            tol = np.float32(15)
        if i % 4 == 1:
            # overlapping windows: duplicate / near-duplicate queries, huge tolerance
            mzq[1] = mzq[0] * np.float32(1 + 5e-6)
            tol = np.float32(200)
            mzq = np.sort(mzq)
        if i % 5 == 4 and nc > 0:
            # use the real m/z of peaks so windows are hit for sure
            s = int(rng.integers(0, case.dia.n_spectra))
            a, b = case.dia.peak_start_idx_list[s], case.dia.peak_stop_idx_list[s]
            mzq = np.sort(rng.choice(case.dia.mz_values[a:b], size=k, replace=False)).astype(
                np.float32
            )
        absolute = i % 2 == 0
        dense, pidx = jit.get_dense(
            frame_limits,
            np.array([[0, 1, 1]], dtype=np.uint64),
            mzq,
            tol,
            quad,
            absolute_masses=absolute,
        )
        d[f"q{i}_frame_limits"] = frame_limits
        d[f"q{i}_mz"] = mzq
        d[f"q{i}_tol"] = np.asarray(tol)
        d[f"q{i}_quad"] = np.asarray(quad, dtype=np.float64)
        d[f"q{i}_absolute"] = np.asarray(absolute)
        d[f"q{i}_dense"] = dense
        d[f"q{i}_pidx"] = np.asarray(pidx, dtype=np.int64)
    d["n_cases"] = np.asarray(n_cases)
    d["caveat"] = np.asarray(CAVEAT)
    path = os.path.join(OUT_DIR, "get_dense_alpharaw.npz")
    np.savez_compressed(path, **d)
    print(path, f"{os.path.getsize(path)/1e6:.2f} MB")


def golden_fragcomp():
    """G6: FragmentCompetition.__call__ on a synthetic PSM / fragment table."""
    rng = np.random.default_rng(13)
    n_psm = 5000
    cycle = syn.make_cycle(n_ms2=8, mz_lo=400, mz_hi=480)
    precursor_idx = rng.permutation(n_psm).astype(np.uint32)
    rank = rng.integers(0, 2, n_psm).astype(np.uint8)
    mz_observed = rng.uniform(400, 480, n_psm).astype(np.float32)
    rt_observed = rng.uniform(0, 120, n_psm).astype(np.float32)
    proba = rng.random(n_psm).astype(np.float32)
    nfrag = rng.integers(3, 13, n_psm)
    # a pool of shared fragment m/z values so that overlaps >= 3 really happen
    pool = np.sort(rng.uniform(200, 350, 160)).astype(np.float32)
    f_pidx, f_rank, f_mz = [], [], []
    for i in range(n_psm):
        mz = rng.choice(pool, nfrag[i], replace=False) * (
            1 + rng.normal(0, 4e-6, nfrag[i])
        ).astype(np.float32)
        f_pidx.append(np.full(nfrag[i], precursor_idx[i], np.uint32))
        f_rank.append(np.full(nfrag[i], rank[i], np.uint8))
        f_mz.append(np.sort(mz).astype(np.float32))
    psm_df = pd.DataFrame(
        {
            "precursor_idx": precursor_idx,
            "rank": rank,
            "mz_observed": mz_observed,
            "rt_observed": rt_observed,
            "proba": proba,
        }
    )
    frag_df = pd.DataFrame(
        {
            "precursor_idx": np.concatenate(f_pidx),
            "rank": np.concatenate(f_rank),
            "mz_observed": np.concatenate(f_mz),
        }
    )
    d = {
        "psm_precursor_idx": precursor_idx,
        "psm_rank": rank,
        "psm_mz_observed": mz_observed,
        "psm_rt_observed": rt_observed,
        "psm_proba": proba,
        "frag_precursor_idx": frag_df["precursor_idx"].values,
        "frag_rank": frag_df["rank"].values,
        "frag_mz_observed": frag_df["mz_observed"].values,
        "cycle": cycle,
    }
    fc = FragmentCompetition(rt_tol_seconds=3, mass_tol_ppm=15, thread_count=1)
    res = fc(psm_df.copy(), frag_df.copy(), cycle)
    d["surviving_precursor_idx"] = res["precursor_idx"].values
    d["surviving_rank"] = res["rank"].values
    d["caveat"] = np.asarray(CAVEAT)
    path = os.path.join(OUT_DIR, "fragcomp.npz")
    np.savez_compressed(path, **d)
    print(path, f"{len(res)}/{n_psm} survive, {os.path.getsize(path)/1e6:.2f} MB")


def timstof_to_jit(dia):
    from alphadia.search.jitclasses.bruker_jit import TimsTOFTransposeJIT

    n_frames = dia.rt_values.shape[0]
    return TimsTOFTransposeJIT(
        np.ones(n_frames),                      # accumulation_times
        dia.cycle,
        dia.dia_mz_cycle,
        dia.dia_precursor_cycle,
        dia.frame_max_index,
        np.ones(dia.mz_values.shape[0]),        # intensity_corrections
        60000, 0,                               # intensity max / min
        dia.intensity_values,
        1.0,                                    # max_accumulation_time
        float(dia.mobility_values.max()), float(dia.mobility_values.min()),
        dia.mobility_values,
        dia.mz_values,
        np.zeros(1, np.int64), 0,               # precursor_indices, precursor_max_index
        np.zeros(1, np.int64),                  # quad_indptr
        float(dia.cycle.max()), float(dia.cycle[dia.cycle > 0].min()),
        np.zeros((1, 2)),                       # quad_mz_values
        np.zeros(1, np.int64),                  # raw_quad_indptr
        dia.rt_values,
        dia.scan_max_index,
        dia.mz_values.shape[0],                 # tof_max_index
        0,                                      # use_calibrated_mz_values_as_default
        dia.zeroth_frame,
        dia.push_indices,
        dia.tof_indptr,
    )


class DuckTims:
    def __init__(self, dia):
        self.cycle = dia.cycle
        self._jit = timstof_to_jit(dia)

    def to_jitclass(self):
        return self._jit


TIMS_COLS = ["cycle", "dia_precursor_cycle", "rt_values", "mobility_values", "mz_values",
             "tof_indptr", "push_indices", "intensity_values"]


def golden_selection_timstof():
    """CandidateSelection.__call__ on a small ion-mobility run (2-D tiles, find_peaks_2d); same FFT
    stand-in and caveats as golden_selection."""
    import ref_shim

    ref_shim.install_selection_glue()
    from alphadia.search.selection import selection as ref_sel
    from alphadia.search.selection.config_df import CandidateSelectionConfig

    case = syn.make_timstof_case(n_precursors=90, n_cycles=70, config_id=46, per_precursor=1,
                                 scan_max_index=96, planted_fraction=0.7, n_tof=24000, events_per_push=20.0)
    d = {"tims_" + c: getattr(case.dia, c) for c in TIMS_COLS}
    d["tims_scan_max_index"] = np.asarray(case.dia.scan_max_index)
    d["tims_zeroth_frame"] = np.asarray(case.dia.zeroth_frame)
    for c in FRAG_COLS:
        d["frag_" + c] = case.library.fragment_df[c].values
    for c in PREC_NUM_COLS:
        d["prec_" + c] = case.library.precursor_df[c].values
    d["caveat"] = np.asarray(CAVEAT)
    cfg = CandidateSelectionConfig()
    cfg.update(dict(rt_tolerance=2.0, mobility_tolerance=0.22, candidate_count=3, min_size_rt=2,
                    peak_len_rt=1.5, sigma_scale_rt=0.5, peak_len_mobility=0.06, sigma_scale_mobility=1.0,
                    max_size_mobility=20))
    dia = DuckTims(case.dia)
    dia.has_mobility = True
    cs = ref_sel.CandidateSelection(
        dia, case.library.precursor_df.copy(), case.library.fragment_df.copy(), cfg,
        rt_column="rt_library", mobility_column="mobility_library",
        precursor_mz_column="mz_library", fragment_mz_column="mz_library",
        fwhm_rt=cfg.peak_len_rt, fwhm_mobility=cfg.peak_len_mobility,
    )
    df = cs(thread_count=1)
    d["kernel"] = np.asarray(cs.kernel, dtype=np.float32)
    cj = cs.config_jit
    for k in ("rt_tolerance mobility_tolerance precursor_mz_tolerance fragment_mz_tolerance candidate_count "
              "top_k_precursors exclude_shared_ions kernel_size f_mobility f_rt center_fraction "
              "min_size_mobility min_size_rt max_size_mobility max_size_rt use_weighted_score "
              "join_close_candidates join_close_candidates_scan_threshold "
              "join_close_candidates_cycle_threshold sigma_scale_rt sigma_scale_mobility peak_len_rt "
              "peak_len_mobility").split():
        d[f"cfg_{k}"] = np.asarray(getattr(cj, k))
    for k in ("feature_mean", "feature_std", "feature_weight"):
        d[f"cfg_{k}"] = np.asarray(getattr(cj, k), dtype=np.float64)
    for c in ("precursor_idx rank score scan_center scan_start scan_stop frame_center frame_start "
              "frame_stop elution_group_idx decoy").split():
        d[f"out_{c}"] = df[c].values
    print(len(df), "candidates for", df["precursor_idx"].nunique(), "precursors; ranks", np.bincount(df["rank"].values),
          "kernel", cs.kernel.shape, "scan width", np.unique(df["scan_stop"] - df["scan_start"]))
    path = os.path.join(OUT_DIR, "selection_timstof.npz")
    np.savez_compressed(path, **d)
    print(path, f"{os.path.getsize(path)/1e6:.2f} MB")


def golden_timstof():
    """G2 + G4 for ion-mobility data: TimsTOFTransposeJIT.get_dense and full scoring."""
    case = syn.make_timstof_case(n_precursors=160, n_cycles=36)
    d = {"tims_" + c: getattr(case.dia, c) for c in TIMS_COLS}
    d["tims_scan_max_index"] = np.asarray(case.dia.scan_max_index)
    d["tims_zeroth_frame"] = np.asarray(case.dia.zeroth_frame)
    for c in FRAG_COLS:
        d["frag_" + c] = case.library.fragment_df[c].values
    for c in PREC_NUM_COLS:
        d["prec_" + c] = case.library.precursor_df[c].values
    for c in CAND_COLS:
        d["cand_" + c] = case.candidates_df[c].values

    # direct get_dense cases
    jit = timstof_to_jit(case.dia)
    rng = np.random.default_rng(17)
    L, S = case.dia.cycle_len, case.dia.scan_max_index
    n_cases = 16
    for i in range(n_cases):
        c0 = int(rng.integers(0, 20)); nc = int(rng.integers(3, 12))
        s0 = int(rng.integers(0, S - 24)); ns = int(rng.integers(4, 24))
        frame_limits = np.array([[c0 * L + 1, (c0 + nc) * L + 1, 1]], dtype=np.uint64)
        scan_limits = np.array([[s0, s0 + ns, 1]], dtype=np.uint64)
        k = int(rng.integers(2, 10))
        if i % 3 == 0:
            mzq = np.sort(rng.uniform(400, 480, k)).astype(np.float32)
            quad = np.array([[-1.0, -1.0]])
        else:
            mzq = np.sort(rng.uniform(200, 350, k)).astype(np.float32)
            lo = rng.uniform(400, 470)
            quad = np.array([[lo, lo + rng.uniform(1.0, 25.0)]], dtype=np.float32)
        tol = np.float32(30 if i % 2 else 80)
        dense, pidx = jit.get_dense(frame_limits, scan_limits, mzq, tol, quad, absolute_masses=True)
        d[f"q{i}_frame_limits"] = frame_limits
        d[f"q{i}_scan_limits"] = scan_limits
        d[f"q{i}_mz"] = mzq
        d[f"q{i}_tol"] = np.asarray(tol)
        d[f"q{i}_quad"] = np.asarray(quad, dtype=np.float64)
        d[f"q{i}_dense"] = dense
        d[f"q{i}_pidx"] = np.asarray(pidx, dtype=np.int64)
    d["n_cases"] = np.asarray(n_cases)

    # full scoring (handler defaults)
    cfg = CandidateScoringConfig()
    cfg.update(SCORING_CONFIGS["handler_default"])
    dia = DuckTims(case.dia)
    cs = ref_scoring.CandidateScoring(
        dia_data=dia,
        precursors_flat=case.library.precursor_df.copy(),
        fragments_flat=case.library.fragment_df.copy(),
        rt_column="rt_library", mobility_column="mobility_library",
        precursor_mz_column="mz_library", fragment_mz_column="mz_library", config=cfg,
    )
    cands = case.candidates_df.copy()
    fragment_container = cs.assemble_fragments()
    sgc = cs.assemble_score_group_container(cands)
    out = OutputPsmDF(sgc.get_candidate_count(), cs.config.top_k_fragments)
    ref_scoring._process_score_groups(range(len(sgc)), sgc, out, fragment_container, dia.to_jitclass(),
                                      cs.config.to_jitclass(), cs.quadrupole_calibration.jit, False)
    d.update(out_to_dict(out))
    cfgj = cfg.to_jitclass()
    for kk in ("collect_fragments score_grouped exclude_shared_ions top_k_fragments top_k_isotopes "
               "reference_channel quant_window quant_all precursor_mz_tolerance "
               "fragment_mz_tolerance experimental_xic").split():
        d["cfg_" + kk] = np.asarray(getattr(cfgj, kk))
    d["caveat"] = np.asarray(CAVEAT)
    path = os.path.join(OUT_DIR, "scoring_timstof.npz")
    np.savez_compressed(path, **d)
    v = np.asarray(out.valid)
    print(f"{path}: {v.sum()}/{len(v)} valid, {os.path.getsize(path)/1e6:.2f} MB, "
          f"mean f29 {np.nanmean(out.features[v][:, 29]):.3f}")


if __name__ == "__main__":
    if "--timstof-only" in sys.argv:
        golden_timstof()
        sys.exit(0)
    if "--selection-timstof-only" in sys.argv:
        golden_selection_timstof()
        sys.exit(0)
    if "--selection-kats-only" in sys.argv:
        golden_selection_kats()
        sys.exit(0)
    if "--transpose-only" in sys.argv:
        golden_transpose()
        sys.exit(0)
    if "--selection-only" in sys.argv:
        golden_selection()
        sys.exit(0)
    if "--small-only" in sys.argv:
        golden_get_dense()
        golden_fragcomp()
        golden_timstof()
        sys.exit(0)
    if "--edges-only" in sys.argv:
        golden_edges()
        sys.exit(0)
    if "--multiplex-only" in sys.argv:
        golden_multiplex()
        sys.exit(0)
    if "--staging-only" in sys.argv:
        golden_staging()
        sys.exit(0)
    if "--host-helpers-only" in sys.argv:
        golden_host_helpers()
        sys.exit(0)
    if "--manyfrag-only" in sys.argv:
        golden_scoring(which=MANYFRAG_CONFIGS)
        sys.exit(0)
    golden_get_dense()
    golden_fragcomp()
    golden_scoring()
    golden_multiplex()
    golden_host_helpers()
    golden_staging()
    golden_edges()
    golden_selection()
    golden_selection_kats()
    golden_selection_timstof()
    golden_transpose()
    golden_timstof()
