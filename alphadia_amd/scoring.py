"""Host side of the MI355X candidate-scoring path.

Mirrors the reference operator interface so that callers can switch classes
without touching their code:

* ``CandidateScoringConfig``  <- alphadia/search/scoring/config.py:63-222
* ``OutputPsmDF``             <- alphadia/search/scoring/output.py:17-97
* ``HipCandidateScoring``     <- ``CandidateScoring`` alphadia/search/scoring/scoring.py:140-661
  (same constructor keywords, same ``__call__`` signature, same returned
  ``(features_df, fragments_df)`` column contract)

All per-candidate arithmetic runs in hand-written HIP kernels behind the C ABI
of ``include/alphadia_hip.h``.  There is no CPU fallback: if the shared
library is missing, construction fails.
"""

from __future__ import annotations

import logging
import os
from types import SimpleNamespace

import numpy as np
import pandas as pd

from alphadia_amd import _abi

logger = logging.getLogger(__name__)

# scoring.py:34-81
DEFAULT_FEATURE_COLUMNS = [
    "base_width_mobility",
    "base_width_rt",
    "rt_observed",
    "mobility_observed",
    "mono_ms1_intensity",
    "top_ms1_intensity",
    "sum_ms1_intensity",
    "weighted_ms1_intensity",
    "weighted_mass_deviation",
    "weighted_mass_error",
    "mz_observed",
    "mono_ms1_height",
    "top_ms1_height",
    "sum_ms1_height",
    "weighted_ms1_height",
    "isotope_intensity_correlation",
    "isotope_height_correlation",
    "n_observations",
    "intensity_correlation",
    "height_correlation",
    "intensity_fraction",
    "height_fraction",
    "intensity_fraction_weighted",
    "height_fraction_weighted",
    "mean_observation_score",
    "sum_b_ion_intensity",
    "sum_y_ion_intensity",
    "diff_b_y_ion_intensity",
    "f_masked",
    "fragment_scan_correlation",
    "template_scan_correlation",
    "fragment_frame_correlation",
    "top3_frame_correlation",
    "template_frame_correlation",
    "top3_b_ion_correlation",
    "n_b_ions",
    "top3_y_ion_correlation",
    "n_y_ions",
    "cycle_fwhm",
    "mobility_fwhm",
    "delta_frame_peak",
    "top_3_ms2_mass_error",
    "mean_ms2_mass_error",
    "n_overlapping",
    "mean_overlapping_intensity",
    "mean_overlapping_mass_error",
]
assert len(DEFAULT_FEATURE_COLUMNS) == _abi.NUM_FEATURES

# scoring.py:83-107
DEFAULT_CANDIDATE_COLUMNS = [
    "elution_group_idx",
    "scan_center",
    "scan_start",
    "scan_stop",
    "frame_center",
    "frame_start",
    "frame_stop",
]
DEFAULT_PRECURSOR_COLUMNS = [
    "rt_library",
    "mobility_library",
    "mz_library",
    "charge",
    "decoy",
    "channel",
    "flat_frag_start_idx",
    "flat_frag_stop_idx",
    "proteins",
    "genes",
    "sequence",
    "mods",
    "mod_sites",
]
# scoring.py:542-557
FRAGMENT_DF_COLUMNS = [
    "precursor_idx",
    "rank",
    "mz_library",
    "mz",
    "mz_observed",
    "height",
    "intensity",
    "mass_error",
    "correlation",
    "position",
    "number",
    "type",
    "charge",
    "loss_type",
]
MAX_FRAGMENT_MZ_TOLERANCE = 200  # constants/settings.py:6


class CandidateScoringConfig:
    """Hyper-parameters of candidate scoring (config.py:63-222)."""

    _FIELDS = (
        "collect_fragments",
        "score_grouped",
        "exclude_shared_ions",
        "top_k_fragments",
        "top_k_isotopes",
        "reference_channel",
        "quant_window",
        "quant_all",
        "precursor_mz_tolerance",
        "fragment_mz_tolerance",
        "experimental_xic",
    )

    def __init__(self):
        self.collect_fragments = True
        self.score_grouped = False
        self.exclude_shared_ions = True
        self.top_k_fragments = 12
        self.top_k_isotopes = 4
        self.reference_channel = -1
        self.quant_window = 3
        self.quant_all = False
        self.precursor_mz_tolerance = 15
        self.fragment_mz_tolerance = 15
        self.experimental_xic = False
        # not fields of the reference's config: the parameters of a FITTED quadrupole calibration
        # (SimpleQuadrupoleJit.sigma / .delta_mu, quadrupole.py:72-76); None = the class defaults (0.2, 0.0).
        # HipCandidateScoring takes them from its ``quadrupole_calibration`` argument (kept on the scorer, this
        # object is not written to); setting them here by hand works too.
        self.quadrupole_sigma = None
        self.quadrupole_delta_mu = None

    def update(self, input_dict: dict) -> None:
        """Type-checked update (jit_config.py:69-138)."""
        for key, value in input_dict.items():
            if key not in self._FIELDS:
                raise ValueError(f"Parameter {key} does not exist in CandidateScoringConfig")
            current = getattr(self, key)
            if not isinstance(value, type(current)):
                try:
                    value = type(current)(value)
                except Exception as e:
                    raise ValueError(f"Parameter {key} has wrong type {type(value)}") from e
            setattr(self, key, value)

    def validate(self) -> None:
        """config.py:197-219"""
        assert isinstance(self.score_grouped, bool), "score_grouped must be a boolean"
        assert self.top_k_fragments > 0, "top_k_fragments must be greater than 0"
        assert self.top_k_isotopes > 0, "top_k_isotopes must be greater than 0"
        assert self.reference_channel >= -1, (
            "reference_channel must be greater than or equal to -1"
        )
        assert self.precursor_mz_tolerance >= 0, (
            "precursor_mz_tolerance must be greater than or equal to 0"
        )
        assert self.precursor_mz_tolerance < 200, "precursor_mz_tolerance must be less than 200"
        assert self.fragment_mz_tolerance >= 0, (
            "fragment_mz_tolerance must be greater than or equal to 0"
        )
        assert self.fragment_mz_tolerance <= MAX_FRAGMENT_MZ_TOLERANCE, (
            f"fragment_mz_tolerance must be less than or equal {MAX_FRAGMENT_MZ_TOLERANCE}"
        )

    def to_jitclass(self):
        """Plain value object with the CandidateScoringConfigJIT field names/dtypes."""
        self.validate()
        return SimpleNamespace(
            collect_fragments=bool(self.collect_fragments),
            score_grouped=bool(self.score_grouped),
            exclude_shared_ions=bool(self.exclude_shared_ions),
            top_k_fragments=np.uint32(self.top_k_fragments),
            top_k_isotopes=np.uint32(self.top_k_isotopes),
            reference_channel=np.int16(self.reference_channel),
            quant_window=np.uint32(self.quant_window),
            quant_all=bool(self.quant_all),
            precursor_mz_tolerance=np.float32(self.precursor_mz_tolerance),
            fragment_mz_tolerance=np.float32(self.fragment_mz_tolerance),
            experimental_xic=bool(self.experimental_xic),
            quadrupole_sigma=None if self.quadrupole_sigma is None else tuple(float(x) for x in self.quadrupole_sigma),
            quadrupole_delta_mu=None if self.quadrupole_delta_mu is None
            else tuple(float(x) for x in self.quadrupole_delta_mu),
        )

    def __repr__(self) -> str:
        return "<CandidateScoringConfig " + ", ".join(
            f"{k}={getattr(self, k)}" for k in self._FIELDS
        ) + ">"


class OutputPsmDF:
    """SoA result buffers (output.py:17-97) filled by the HIP kernels."""

    def __init__(self, arrays: dict):
        self.__dict__.update(arrays)
        self.valid = self.valid.view(np.bool_)

    def fragment_rows(self):
        """Flat positions (candidate row * top_k + slot) of the filled fragment rows."""
        return np.flatnonzero(self.fragment_mz_library.reshape(-1) > 0)

    def to_fragment_df(self, idx=None):
        idx = self.fragment_rows() if idx is None else idx
        cols = (
            "fragment_precursor_idx fragment_rank fragment_mz_library fragment_mz "
            "fragment_mz_observed fragment_height fragment_intensity fragment_mass_error "
            "fragment_correlation fragment_position fragment_number fragment_type "
            "fragment_charge fragment_loss_type"
        ).split()
        return tuple(getattr(self, c).reshape(-1).take(idx) for c in cols)

    def to_precursor_df(self):
        v = self.valid
        return self.precursor_idx[v], self.rank[v], self.features[v]


def get_isotope_column_names(colnames) -> list[str]:
    """alphadia/utils.py:55-71 + scoring.py:110-111"""
    iso = []
    for col in colnames:
        if col[:2] == "i_":
            try:
                iso.append(int(col[2:]))
            except ValueError:
                logger.warning(f"Column {col} does not seem to be a valid isotope column")
    return [f"i_{i}" for i in sorted(iso)]


def _row_lookup(right_df: pd.DataFrame, left_df: pd.DataFrame, on: list[str]):
    """Position in ``right_df`` of the row whose key columns equal those of every ``left_df`` row
    (-1 where there is none), or None when the right keys are not unique / not integer-like."""
    keys_r = [right_df[c].values for c in on]
    keys_l = [left_df[c].values for c in on]
    if not all(np.issubdtype(k.dtype, np.integer) for k in keys_r + keys_l):
        return None
    # composite key: mixed-radix over the value ranges of the right table
    code_r = np.zeros(len(right_df), dtype=np.int64)
    code_l = np.zeros(len(left_df), dtype=np.int64)
    ok_l = np.ones(len(left_df), dtype=bool)
    radix = 1
    for kr, kl in zip(keys_r, keys_l, strict=True):
        lo = int(kr.min()) if len(kr) else 0
        span = (int(kr.max()) - lo + 1) if len(kr) else 1
        if radix * span >= 2**62:
            return None
        kl64 = kl.astype(np.int64)
        ok_l &= (kl64 >= lo) & (kl64 < lo + span)
        code_r += (kr.astype(np.int64) - lo) * radix
        code_l += (np.clip(kl64, lo, lo + span - 1) - lo) * radix
        radix *= span
    order = np.argsort(code_r, kind="stable")
    sorted_codes = code_r[order]
    if len(sorted_codes) > 1 and (sorted_codes[1:] == sorted_codes[:-1]).any():
        return None
    pos = np.searchsorted(sorted_codes, code_l)
    pos_c = np.minimum(pos, max(len(sorted_codes) - 1, 0))
    hit = ok_l & (sorted_codes[pos_c] == code_l) if len(sorted_codes) else np.zeros(len(left_df), dtype=bool)
    return np.where(hit, order[pos_c] if len(order) else 0, -1)


def merge_missing_columns(left_df, right_df, right_columns, on=None, how="left"):
    """Bring the columns of ``right_columns`` that ``left_df`` lacks over from ``right_df``.

    Interface, checks and result of the reference helper (scoring/utils.py:203-266).  The common
    case (``how="left"``, integer keys, unique on the right) is a sort + searchsorted lookup and a
    gather per column; anything else is handed to ``DataFrame.merge``."""
    on = [on] if isinstance(on, str) else on
    wanted = [right_columns] if isinstance(right_columns, str) else list(right_columns)
    to_add = [c for c in dict.fromkeys(wanted) if c not in left_df.columns]
    if not to_add:
        return left_df
    unknown = [c for c in to_add if c not in right_df.columns]
    if unknown:
        raise ValueError(f"Columns {unknown} must be present in right_df")
    if on is None:
        raise ValueError("Parameter on must be specified")
    for name, frame in (("left_df", left_df), ("right_df", right_df)):
        if any(c not in frame.columns for c in on):
            raise ValueError(f"Columns {on} must be present in {name}")
    if how not in ("left", "right", "inner", "outer"):
        raise ValueError("Parameter how must be one of left, right, inner, outer")
    rows = _row_lookup(right_df, left_df, on) if how == "left" else None
    if rows is None or (rows < 0).any():  # duplicate / non-integer keys, or rows without a partner (NaN fill)
        return left_df.merge(right_df[on + to_add], on=on, how=how)
    out = left_df.reset_index(drop=True)  # (a fresh RangeIndex, as DataFrame.merge returns)
    for c in to_add:
        out[c] = right_df[c].values[rows]
    return out


_CAND_REQUIRED = {
    "elution_group_idx": np.uint32,
    "precursor_idx": np.uint32,
    "rank": np.uint8,
    "scan_start": np.int64,
    "scan_stop": np.int64,
    "scan_center": np.int64,
    "frame_start": np.int64,
    "frame_stop": np.int64,
    "frame_center": np.int64,
}


def assemble_candidates(
    candidates_df: pd.DataFrame,
    precursors_flat_df: pd.DataFrame,
    precursor_mz_column: str,
    score_grouped: bool = False,
    reference_channel: int = -1,
    pool=None,
) -> dict:
    """Candidate table -> struct of arrays in score-group order.

    ``pool`` (a ``runtime.PinnedPool``) places the columns that are uploaded to the GPU in
    page-locked memory, written directly by the final gather (no extra copy): they are then valid
    until the next call with the same pool.

    Does the work of ``assemble_score_group_container`` (scoring.py:273-353),
    ``calculate_score_groups`` (scoring/utils.py:269-410) and
    ``ScoreGroupContainer.build_from_df`` (score_group.py:145-229) without
    creating one object per candidate: precursor columns are looked up with a
    searchsorted on ``precursor_idx`` and the order is one lexsort.
    """
    n = len(candidates_df)
    for col in _CAND_REQUIRED:
        if col not in candidates_df.columns and col != "rank":
            raise KeyError(f"candidates_df is missing required column '{col}'")
    cols = {
        c: np.ascontiguousarray(candidates_df[c].values, dtype=dt)
        for c, dt in _CAND_REQUIRED.items()
        if c in candidates_df.columns
    }
    if "rank" not in cols:
        cols["rank"] = np.zeros(n, dtype=np.uint8)

    lib_pidx = precursors_flat_df["precursor_idx"].values
    lib_rows = None  # row in the caller's table of every row of the sorted one
    if len(lib_pidx) > 1 and not np.all(lib_pidx[1:] > lib_pidx[:-1]):
        lib_rows = np.argsort(lib_pidx, kind="stable")
        precursors_flat_df = precursors_flat_df.iloc[lib_rows]
        lib_pidx = precursors_flat_df["precursor_idx"].values
    # (strictly ascending is only certain where the check above passed: after the argsort branch the keys are merely
    # non-decreasing - a library with a duplicated precursor_idx, e.g. [0, 1, 1, 3], would pass the end-point test and
    # map the absent index 2 to a wrong row)
    if lib_rows is None and len(lib_pidx) and int(lib_pidx[0]) == 0 and int(lib_pidx[-1]) == len(lib_pidx) - 1:
        # strictly ascending from 0 to len - 1: the index IS the row (a library as alphabase writes it)
        pos = cols["precursor_idx"].astype(np.intp)
        if n and int(pos.max()) >= len(lib_pidx):
            raise ValueError("candidates_df contains precursor_idx values missing from precursors_flat")
        pos_c = pos
    else:
        pos = np.searchsorted(lib_pidx, cols["precursor_idx"])
        pos_c = np.minimum(pos, max(len(lib_pidx) - 1, 0))
        if n and (len(lib_pidx) == 0 or not np.array_equal(lib_pidx[pos_c], cols["precursor_idx"])):
            raise ValueError("candidates_df contains precursor_idx values missing from precursors_flat")

    def from_lib(name, dtype, default=None):
        if name in candidates_df.columns:
            return np.ascontiguousarray(candidates_df[name].values, dtype=dtype)
        if name in precursors_flat_df.columns:
            return np.ascontiguousarray(precursors_flat_df[name].values[pos_c], dtype=dtype)
        if default is None:
            raise ValueError(f"Columns ['{name}'] must be present in right_df")
        return np.full(n, default, dtype=dtype)

    iso_names = get_isotope_column_names(
        list(dict.fromkeys(list(candidates_df.columns) + list(precursors_flat_df.columns)))
    )
    # library lookups: one gather per column, side by side on the host pool
    lookups = [("channel", np.uint8, 0), ("decoy", np.uint8, None), ("flat_frag_start_idx", np.uint32, None),
               ("flat_frag_stop_idx", np.uint32, None), ("charge", np.uint8, None), (precursor_mz_column, np.float32, None)]
    lookups += [(c, np.float32, None) for c in iso_names]
    looked = _parallel([(lambda a=a: from_lib(*a)) for a in lookups])
    channel, decoy, frag_start, frag_stop, charge, prec_mz = looked[:6]
    if iso_names:
        iso = np.stack(looked[6:], axis=1)
    else:
        iso = np.ones((n, 1), dtype=np.float32)  # scoring.py:322-323

    # score-group order (scoring/utils.py:388-410): elution group, decoy, rank, precursor.  A table that is in
    # that order already - what candidate selection emits for a library sorted by precursor - skips the sort
    eg0, rk0, pi0 = cols["elution_group_idx"], cols["rank"], cols["precursor_idx"]
    presorted = n < 2
    if not presorted:
        a_eg, b_eg = eg0[:-1], eg0[1:]
        lt = a_eg < b_eg
        tie = a_eg == b_eg
        for a_k, b_k in ((decoy[:-1], decoy[1:]), (rk0[:-1], rk0[1:]), (pi0[:-1], pi0[1:])):
            lt |= tie & (a_k < b_k)
            tie &= a_k == b_k
        presorted = bool((lt | tie).all())
    order = np.arange(n, dtype=np.intp) if presorted else np.lexsort((pi0, rk0, decoy, eg0))
    order_arg = None if presorted else order

    def ordered(name, col):
        """``col[order]``, in page-locked memory when a pool is given"""
        if pool is not None:
            return pool.take("cand:" + name, col, order_arg)
        return col[order] if order_arg is not None else np.array(col, copy=True)

    eg = eg0 if presorted else eg0[order]
    dc = decoy if presorted else decoy[order]
    rk, pi = ordered("rank", rk0), ordered("precursor_idx", pi0)
    if score_grouped and n:
        change = np.ones(n, dtype=bool)
        change[1:] = (eg[1:] != eg[:-1]) | (dc[1:] != dc[:-1]) | (rk[1:] != rk[:-1])
        score_group_idx = (np.cumsum(change) - 1).astype(np.uint32)
    else:
        score_group_idx = np.arange(n, dtype=np.uint32)
    if n > 1:
        dup = (pi[1:] == pi[:-1]) & (score_group_idx[1:] == score_group_idx[:-1])
        if dup.any():
            raise ValueError("precursor_idx must be unique within a score group")

    ch = channel if presorted else channel[order]
    flags = np.zeros(n, dtype=np.uint8) if pool is None else pool.empty("cand:flags", (n,), np.uint8)
    flags[...] = 0
    if reference_channel >= 0 and n:
        has_ref = np.zeros(int(score_group_idx[-1]) + 1, dtype=bool)
        has_ref[score_group_idx[ch == reference_channel]] = True
        flags[~has_ref[score_group_idx]] = _abi.FLAG_SKIP

    prec_row = pos_c if lib_rows is None else lib_rows[pos_c]
    out = {
        "order": order,
        # row of every candidate in the caller's precursor table
        "prec_row": prec_row if presorted else prec_row[order],
        "score_group_idx": score_group_idx,
        "elution_group_idx": eg,
        "decoy": dc,
        "channel": ch,
        "precursor_idx": pi,
        "rank": rk,
        "flags": flags,
    }
    # the uploaded columns, in processing order (page-locked when a pool is given): side by side as well
    moves = [("frag_start_idx", frag_start), ("frag_stop_idx", frag_stop), ("charge", charge), ("precursor_mz", prec_mz),
             ("isotope_intensity", np.ascontiguousarray(iso))]
    moves += [(c, cols[c]) for c in ("scan_start", "scan_stop", "scan_center", "frame_start", "frame_stop", "frame_center")]
    for (name, _), arr in zip(moves, _parallel([(lambda m=m: ordered(*m)) for m in moves])):
        out[name] = arr
    return out


def pack_assembled(soa: dict) -> _abi.Marshalled:
    return _abi.pack_candidates(
        soa["precursor_idx"],
        soa["rank"],
        soa["frag_start_idx"],
        soa["frag_stop_idx"],
        soa["scan_start"],
        soa["scan_stop"],
        soa["scan_center"],
        soa["frame_start"],
        soa["frame_stop"],
        soa["frame_center"],
        soa["charge"],
        soa["precursor_mz"],
        soa["isotope_intensity"],
        flags=soa["flags"],
    )


def fragment_columns(fragments_flat: pd.DataFrame, fragment_mz_column: str) -> tuple:
    """assemble_fragments (scoring.py:355-392): the nine FragmentContainer arrays."""
    if "cardinality" not in fragments_flat.columns:
        logger.warning(
            "Fragment cardinality column not found in fragment dataframe. Setting cardinality to 1."
        )
        fragments_flat["cardinality"] = np.ones(len(fragments_flat), dtype=np.uint8)
    return (
        fragments_flat["mz_library"].values,
        fragments_flat[fragment_mz_column].values,
        fragments_flat["intensity"].values,
        fragments_flat["type"].values,
        fragments_flat["loss_type"].values,
        fragments_flat["charge"].values,
        fragments_flat["number"].values,
        fragments_flat["position"].values,
        fragments_flat["cardinality"].values,
    )


_HOST_POOL = None


def _host_pool():
    """Threads for the column gathers of ``collect_candidates`` / ``collect_fragments``: numpy releases the
    GIL inside ``take`` / ``flatnonzero`` on plain dtypes, so the ~90 columns of the result frames are built
    side by side.  Sized to the CPU quota of the container (at most 16), like the library's own host team."""
    global _HOST_POOL
    if _HOST_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        # the library's own figure (adh_host_threads): cgroup quota, affinity mask and hardware threads, this rank's
        # share of them (LOCAL_WORLD_SIZE ranks build their frames side by side), at most 16; ADH_HOST_THREADS overrides
        from . import runtime as _rt

        n = _rt.host_threads(1 << 40)[0]
        _HOST_POOL = (ThreadPoolExecutor(max_workers=max(n, 1), thread_name_prefix="adh_collect"), max(n, 1))
    return _HOST_POOL


def _parallel(tasks):
    """Run callables on the host pool; results in task order."""
    pool, n = _host_pool()
    if n == 1 or len(tasks) <= 1:
        return [t() for t in tasks]
    return [f.result() for f in [pool.submit(t) for t in tasks]]


def _take_chunked(src: np.ndarray, idx: np.ndarray, parts: int) -> np.ndarray:
    """``src[idx]`` along axis 0, as ``parts`` slices of idx gathered into one result."""
    out = np.empty((len(idx),) + src.shape[1:], dtype=src.dtype)
    if src.dtype == object or parts <= 1 or len(idx) < 65536:
        np.take(src, idx, axis=0, out=out)
        return out
    cuts = np.linspace(0, len(idx), parts + 1).astype(np.int64)
    _parallel([(lambda a=a, b=b: np.take(src, idx[a:b], axis=0, out=out[a:b])) for a, b in zip(cuts[:-1], cuts[1:])])
    return out


def _take_rows_transposed(src: np.ndarray, idx: np.ndarray, parts: int, block: int = 4096) -> np.ndarray:
    """``src[idx].T`` of a row-major table as a C-contiguous [column][row] array (every column of the result is
    contiguous): row blocks small enough for the cache are gathered and written transposed, slices of idx in
    parallel."""
    m, ncol = len(idx), src.shape[1]
    out = np.empty((ncol, m), dtype=src.dtype)

    def work(a, b):
        for lo in range(a, b, block):
            hi = min(lo + block, b)
            out[:, lo:hi] = src[idx[lo:hi]].T

    if parts <= 1 or m < 65536:
        work(0, m)
        return out
    cuts = np.linspace(0, m, parts + 1).astype(np.int64)
    _parallel([(lambda a=a, b=b: work(int(a), int(b))) for a, b in zip(cuts[:-1], cuts[1:])])
    return out


def _flatnonzero_chunked(mask_source: np.ndarray, parts: int) -> np.ndarray:
    """``np.flatnonzero(mask_source > 0)`` of a flat array, chunk by chunk on the pool."""
    n = len(mask_source)
    if parts <= 1 or n < 1 << 20:
        return np.flatnonzero(mask_source > 0)
    cuts = np.linspace(0, n, parts + 1).astype(np.int64)
    found = _parallel([(lambda a=a, b=b: np.flatnonzero(mask_source[a:b] > 0) + a) for a, b in zip(cuts[:-1], cuts[1:])])
    return np.concatenate(found)


def _take_missing_columns(left_df, right_df, right_columns, rows):
    """``merge_missing_columns`` when the matching right row of every left row is known already:
    same columns, order and dtypes as the left merge on a unique key, without the hash join."""
    missing = [c for c in dict.fromkeys(right_columns) if c not in left_df.columns]
    absent = [c for c in missing if c not in right_df.columns]
    if absent:
        raise ValueError(f"Columns {absent} must be present in right_df")
    for c in missing:
        left_df[c] = right_df[c].values[rows]
    return left_df


def collect_candidates(
    candidates_df: pd.DataFrame,
    psm_proto_df: OutputPsmDF,
    precursors_flat_df: pd.DataFrame,
    rt_column: str,
    mobility_column: str,
    precursor_mz_column: str,
    row_maps=None,
    sequence_counts=None,
    compact: dict | None = None,
) -> pd.DataFrame:
    """scoring.py:394-467 (column names, order and merges).

    ``row_maps = (candidate_row, precursor_row)`` of every row of ``psm_proto_df`` (known from
    ``assemble_candidates``) replaces the two hash joins by gathers; ``sequence_counts`` are the
    per-precursor K / R / P counts (constant per library)."""
    candidate_columns = DEFAULT_CANDIDATE_COLUMNS.copy()
    candidate_columns += ["score"] if "score" in candidates_df.columns else []
    if "rank" not in candidates_df.columns:
        candidates_df = candidates_df.assign(rank=np.zeros(len(candidates_df), dtype=np.uint8))
    precursor_df_columns = DEFAULT_PRECURSOR_COLUMNS + get_isotope_column_names(
        precursors_flat_df.columns
    )
    for col in [rt_column, mobility_column, precursor_mz_column]:
        if col not in precursor_df_columns:
            precursor_df_columns.append(col)
    if row_maps is not None:
        return _collect_candidates_by_rows(candidates_df, psm_proto_df, precursors_flat_df, rt_column, candidate_columns,
                                           precursor_df_columns, row_maps, sequence_counts, compact=compact)
    precursor_idx, rank, features = psm_proto_df.to_precursor_df()
    df = pd.DataFrame(features, columns=DEFAULT_FEATURE_COLUMNS)
    df["precursor_idx"] = precursor_idx
    df["rank"] = rank
    df = merge_missing_columns(
        df, candidates_df, candidate_columns, on=["precursor_idx", "rank"], how="left"
    )
    df = merge_missing_columns(
        df, precursors_flat_df, precursor_df_columns, on=["precursor_idx"], how="left"
    )
    df["delta_rt"] = df["rt_observed"] - df[rt_column]
    df["n_K"] = df["sequence"].str.count("K")
    df["n_R"] = df["sequence"].str.count("R")
    df["n_P"] = df["sequence"].str.count("P")
    return df


def _collect_candidates_by_rows(candidates_df, psm_proto_df, precursors_flat_df, rt_column, candidate_columns,
                                precursor_df_columns, row_maps, sequence_counts, compact: dict | None = None) -> pd.DataFrame:
    """``collect_candidates`` when the candidate row and the precursor row of every table row are known: the same
    columns, order and dtypes as the two left merges on unique keys give, as gathers - one task per column on
    the host pool, the 46-column feature block in row slices - and a frame assembled around those arrays
    without another copy (pandas would otherwise re-stack the columns by dtype)."""
    _, threads = _host_pool()
    # ``compact`` (Context.score_host_compact): the device already dropped the invalid rows and transposed the feature
    # table - row ids, ids and the [feature][row] block arrive as they go into the frame
    rows = compact["row"] if compact is not None else np.flatnonzero(np.asarray(psm_proto_df.valid, dtype=bool))
    cand_rows = np.asarray(row_maps[0]).take(rows)
    prec_rows = np.asarray(row_maps[1]).take(rows)
    names: list[str] = list(DEFAULT_FEATURE_COLUMNS)
    jobs: list[tuple[str, np.ndarray, np.ndarray]] = [] if compact is not None else [
        ("precursor_idx", psm_proto_df.precursor_idx, rows), ("rank", psm_proto_df.rank, rows)]
    have = set(names) | {"precursor_idx", "rank"}
    for frame, wanted, idx in ((candidates_df, candidate_columns, cand_rows),
                               (precursors_flat_df, precursor_df_columns, prec_rows)):
        missing = [c for c in dict.fromkeys(wanted) if c not in have]
        absent = [c for c in missing if c not in frame.columns]
        if absent:
            raise ValueError(f"Columns {absent} must be present in right_df")
        for c in missing:
            jobs.append((c, frame[c].values, idx))
            have.add(c)
    counts = None
    if sequence_counts is not None:
        counts = [(name, np.asarray(cnt), prec_rows) for name, cnt in zip(("n_K", "n_R", "n_P"), sequence_counts, strict=True)]
    if compact is not None:
        features = compact["features"]
    else:
        features = _take_rows_transposed(psm_proto_df.features, rows, threads)  # [feature][row]: contiguous columns
    # object columns (proteins ... mod_sites) are reference-counted pointers: NumPy gathers them one column after the
    # other under the GIL (~10 ms per 2.7 M rows); runtime.take_objects gathers them on several threads
    obj_jobs = [(name, src, idx) for name, src, idx in jobs if src.dtype == object]
    num_jobs = [(name, src, idx) for name, src, idx in jobs if src.dtype != object] + (counts or [])
    from . import runtime as _rt

    # the numeric gathers run on the pool (NumPy releases the GIL inside take) while this thread gathers the object
    # columns (which keeps the GIL, but only Python-level code needs it)
    pool, n_pool = _host_pool()
    tasks = [(lambda src=src, idx=idx: src[idx]) for _, src, idx in num_jobs]
    futures = [pool.submit(t) for t in tasks] if n_pool > 1 else None
    cols = {}
    for name, src, idx in obj_jobs:
        cols[name] = _rt.take_objects(src, idx, threads)
    gathered = [f.result() for f in futures] if futures is not None else [t() for t in tasks]
    cols.update({name: arr for (name, _, _), arr in zip(num_jobs, gathered)})
    frame = {name: features[j] for j, name in enumerate(DEFAULT_FEATURE_COLUMNS)}
    if compact is not None:
        frame["precursor_idx"] = compact["precursor_idx"]
        frame["rank"] = compact["rank"]
    frame.update({name: cols[name] for name, _, _ in jobs})
    frame["delta_rt"] = frame["rt_observed"] - frame[rt_column]
    if counts is not None:
        for name, _, _ in counts:
            frame[name] = cols[name]
    # copy=False: every column stays the array built above (one block per column); the default would re-stack
    # them by dtype - 0.5 s per million rows for nothing
    df = pd.DataFrame(frame, copy=False)
    if counts is None:
        df["n_K"] = df["sequence"].str.count("K")
        df["n_R"] = df["sequence"].str.count("R")
        df["n_P"] = df["sequence"].str.count("P")
    return df


def collect_fragments_compact(compact: dict, precursors_flat_df: pd.DataFrame, prec_rows) -> pd.DataFrame:
    """scoring.py:520-580 from the compacted columns of ``Context.score_host_compact``: the filled slots arrive as
    columns; only ``elution_group_idx`` and ``decoy`` are looked up (slot -> candidate row -> precursor row)."""
    missing = [c for c in ("elution_group_idx", "decoy") if c not in FRAGMENT_DF_COLUMNS]
    absent = [c for c in missing if c not in precursors_flat_df.columns]
    if absent:
        raise ValueError(f"Columns {absent} must be present in right_df")
    _, threads = _host_pool()
    frow = compact["fragment_row"]
    prec_rows = np.asarray(prec_rows)
    lib_rows = np.empty(len(frow), dtype=prec_rows.dtype)
    cuts = np.linspace(0, len(frow), (threads if len(frow) > 1 << 20 else 1) + 1).astype(np.int64)
    _parallel([(lambda a=int(a), b=int(b): np.take(prec_rows, frow[a:b], out=lib_rows[a:b])) for a, b in zip(cuts[:-1], cuts[1:])])
    looked_up = [_take_chunked(precursors_flat_df[c].values, lib_rows, threads) for c in missing]
    frame = {c: compact["fragment_" + c] for c in FRAGMENT_DF_COLUMNS}
    frame.update(dict(zip(missing, looked_up, strict=True)))
    return pd.DataFrame(frame, copy=False)


def collect_fragments(psm_proto_df: OutputPsmDF, precursors_flat_df: pd.DataFrame, prec_rows=None) -> pd.DataFrame:
    """scoring.py:520-580; ``prec_rows`` (precursor row of every row of ``psm_proto_df``) replaces the
    hash join by a gather."""
    if prec_rows is None:
        idx = psm_proto_df.fragment_rows()
        df = pd.DataFrame(dict(zip(FRAGMENT_DF_COLUMNS, psm_proto_df.to_fragment_df(idx), strict=True)))
        return merge_missing_columns(
            df, precursors_flat_df, ["elution_group_idx", "decoy"], on=["precursor_idx"], how="left"
        )
    # filled slots (output.py:89-97: fragment_mz_library > 0), then one gather per column on the host pool
    _, threads = _host_pool()
    top_k = psm_proto_df.fragment_mz_library.shape[1]
    idx = _flatnonzero_chunked(psm_proto_df.fragment_mz_library.reshape(-1), threads)
    sources = [getattr(psm_proto_df, "fragment_" + c).reshape(-1) for c in FRAGMENT_DF_COLUMNS]
    missing = [c for c in ("elution_group_idx", "decoy") if c not in FRAGMENT_DF_COLUMNS]
    absent = [c for c in missing if c not in precursors_flat_df.columns]
    if absent:
        raise ValueError(f"Columns {absent} must be present in right_df")
    prec_rows = np.asarray(prec_rows)
    lib_cols = [precursors_flat_df[c].values for c in missing]
    # precursor row of every filled slot: slot -> candidate (a shift when top_k is a power of two) -> precursor
    lib_rows = np.empty(len(idx), dtype=prec_rows.dtype)
    cuts = np.linspace(0, len(idx), (threads if len(idx) > 1 << 20 else 1) + 1).astype(np.int64)

    def slot_rows(a, b):
        np.take(prec_rows, idx[a:b] // top_k, out=lib_rows[a:b])

    gathered = _parallel([(lambda src=src: src.take(idx)) for src in sources] +
                         [(lambda a=a, b=b: slot_rows(int(a), int(b))) for a, b in zip(cuts[:-1], cuts[1:])])[:len(sources)]
    gathered += _parallel([(lambda col=col: col.take(lib_rows)) for col in lib_cols])
    return pd.DataFrame(dict(zip(list(FRAGMENT_DF_COLUMNS) + missing, gathered, strict=True)), copy=False)


def _jit_view(dia_data):
    return dia_data.to_jitclass() if hasattr(dia_data, "to_jitclass") else dia_data


class HipCandidateScoring:
    """Calculate features for each precursor candidate on the GPU.

    Drop-in for ``CandidateScoring`` (scoring.py:140-661): same keyword-only
    constructor, same ``__call__``.  ``device`` selects the GPU (default: the
    current HIP device / LOCAL_RANK).
    """

    def __init__(
        self,
        *,
        dia_data,
        precursors_flat: pd.DataFrame,
        fragments_flat: pd.DataFrame,
        rt_column: str,
        mobility_column: str,
        precursor_mz_column: str,
        fragment_mz_column: str,
        config: CandidateScoringConfig | None = None,
        quadrupole_calibration=None,
        device: int | None = None,
    ):
        from alphadia_amd import runtime  # loads libalphadia_hip.so; raises when missing

        self._dia_data = dia_data
        self.precursors_flat_df = precursors_flat.sort_values(by="precursor_idx")
        self.fragments_flat = fragments_flat
        self.config = config if config is not None else CandidateScoringConfig()
        self.config.validate()
        # fitted transfer-function parameters live on the scorer, not in the caller's config object (which
        # may be reused for a run with another or no calibration); ``_kernel_config`` injects them per call
        self._quadrupole = None
        if quadrupole_calibration is not None:
            # SimpleQuadrupole (quadrupole.py:116-259): the transfer function of candidate scoring is
            # jit.predict = logistic_rectangle(cycle + delta_mu, sigma) (quadrupole.py:94-113); a fitted
            # calibration differs from the default one in sigma and delta_mu only (its calibrated cycle is
            # used for plots, candidate.py:207-214)
            jit = getattr(quadrupole_calibration, "jit", None)
            if jit is None:
                raise AttributeError("quadrupole_calibration must have a jit method")
            if not np.array_equal(jit.cycle, _jit_view(dia_data).cycle):
                raise NotImplementedError(
                    "the quadrupole calibration was built for another cycle than the run's "
                    "(SimpleQuadrupole(dia_data.cycle), scoring.py:209-212)"
                )
            sigma, delta_mu = np.asarray(jit.sigma, dtype=np.float64), np.asarray(jit.delta_mu, dtype=np.float64)
            if sigma.shape != (2,) or delta_mu.shape != (2,) or not (sigma > 0).all():
                raise ValueError("quadrupole calibration: sigma and delta_mu must hold two values, sigma > 0")
            if not (np.array_equal(sigma, [0.2, 0.2]) and np.array_equal(delta_mu, [0.0, 0.0])):
                self._quadrupole = ((float(sigma[0]), float(sigma[1])), (float(delta_mu[0]), float(delta_mu[1])))
        self.rt_column = rt_column
        self.mobility_column = mobility_column
        self.precursor_mz_column = precursor_mz_column
        self.fragment_mz_column = fragment_mz_column

        self._ctx = runtime.get_context(device)
        self._ctx.stage_run(_jit_view(dia_data))
        self._ctx.stage_fragments(*fragment_columns(self.fragments_flat, fragment_mz_column))

    @property
    def dia_data(self):
        return self._dia_data

    def _kernel_config(self):
        """The config value object of a call: the caller's settings plus this scorer's calibration (a
        calibration passed to the constructor wins over parameters set on the config by hand)."""
        cfg = self.config.to_jitclass()
        if self._quadrupole is not None:
            cfg.quadrupole_sigma, cfg.quadrupole_delta_mu = self._quadrupole
        return cfg

    def score_soa(self, soa: dict, with_stats: bool = False, reuse_buffers: bool = False) -> OutputPsmDF:
        """Run the kernels on an assembled candidate SoA; returns host OutputPsmDF.
        ``reuse_buffers``: results live in the context's page-locked buffers until the next call."""
        arrays = self._ctx.score_host(
            pack_assembled(soa), self._kernel_config(), with_stats=with_stats, reuse_buffers=reuse_buffers
        )
        return OutputPsmDF(arrays)

    def _sequence_counts(self):
        """K / R / P counts of every library precursor (scoring.py:462-464), computed once."""
        if getattr(self, "_seq_counts", None) is None:
            seq = self.precursors_flat_df["sequence"]
            self._seq_counts = tuple(seq.str.count(a).values for a in ("K", "R", "P"))
        return self._seq_counts

    def __call__(
        self,
        candidates_df: pd.DataFrame,
        thread_count: int = 10,
        debug: bool = False,
        include_decoy_fragment_features: bool = False,
    ):
        del thread_count, include_decoy_fragment_features  # CPU-threading knobs of the reference
        logger.info("Starting candidate scoring")
        import time

        t_0 = time.perf_counter()
        soa = assemble_candidates(
            candidates_df,
            self.precursors_flat_df,
            self.precursor_mz_column,
            score_grouped=self.config.score_grouped,
            reference_channel=self.config.reference_channel,
            pool=self._ctx.pinned,
        )
        if debug:  # scoring.py:628-631: first 10 score groups only
            keep = soa["score_group_idx"] < 10
            soa["flags"] = np.where(keep, soa["flags"], _abi.FLAG_SKIP).astype(np.uint8)
        t_1 = time.perf_counter()
        if not os.environ.get("ADH_OPERATOR_PADDED"):
            return self._call_compact(candidates_df, soa, t_0, t_1)
        # (the round-4 path, kept for comparison: padded tables to the host, then valid rows / filled slots gathered
        # out of them column by column; the frames copy what they need out of the pooled page-locked tables)
        psm_proto_df = self.score_soa(soa, reuse_buffers=True)
        t_2 = time.perf_counter()
        # The two frames are independent.  The features frame spends most of its time gathering five object
        # columns - one thread, under the GIL - while the fragments frame is all NumPy takes on the host pool: they
        # run side by side (the fragments frame in a thread of its own; both hand their gathers to the same pool).
        import threading

        frag_box: dict = {}

        def fragments_job():
            t_a = time.perf_counter()
            try:
                frag_box["df"] = collect_fragments(psm_proto_df, self.precursors_flat_df, prec_rows=soa["prec_row"])
            except BaseException as exc:  # re-raised on the calling thread
                frag_box["exc"] = exc
            frag_box["ms"] = (time.perf_counter() - t_a) * 1e3

        logger.info("Collecting candidate and fragment features")
        worker = threading.Thread(target=fragments_job, name="adh-collect-fragments")
        worker.start()
        try:
            features_df = collect_candidates(
                candidates_df,
                psm_proto_df,
                self.precursors_flat_df,
                self.rt_column,
                self.mobility_column,
                self.precursor_mz_column,
                row_maps=(soa["order"], soa["prec_row"]),
                sequence_counts=self._sequence_counts(),
            )
        finally:
            t_3 = time.perf_counter()
            worker.join()
        if "exc" in frag_box:
            raise frag_box["exc"]
        fragments_df = frag_box["df"]
        t_4 = time.perf_counter()
        # wall time of the stages of this call, in ms (the reference logs them, scoring.py:634-661); the two collect
        # stages overlap: each is its own wall time, collect_ms the wall time of both
        self.last_timings = {"assemble_ms": (t_1 - t_0) * 1e3, "score_ms": (t_2 - t_1) * 1e3,
                             "collect_candidates_ms": (t_3 - t_2) * 1e3, "collect_fragments_ms": frag_box["ms"],
                             "collect_ms": (t_4 - t_2) * 1e3, "total_ms": (t_4 - t_0) * 1e3}
        logger.info("Finished candidate scoring")
        return features_df, fragments_df


def _call_compact(self, candidates_df: pd.DataFrame, soa: dict, t_0: float, t_1: float):
    """The operator on the compacted copy-out (``adh_score_candidates_compact``): the device drops invalid candidates
    and empty fragment slots and transposes the feature table, the columns land in arrays the frames wrap as they
    are; what is left for the host are the columns the frames take over from the candidate and the precursor table."""
    import threading
    import time

    comp = self._ctx.score_host_compact(pack_assembled(soa), self._kernel_config())
    t_2 = time.perf_counter()
    frag_box: dict = {}

    def fragments_job():
        t_a = time.perf_counter()
        try:
            frag_box["df"] = collect_fragments_compact(comp, self.precursors_flat_df, soa["prec_row"])
        except BaseException as exc:  # re-raised on the calling thread
            frag_box["exc"] = exc
        frag_box["ms"] = (time.perf_counter() - t_a) * 1e3

    logger.info("Collecting candidate and fragment features")
    worker = threading.Thread(target=fragments_job, name="adh-collect-fragments")
    worker.start()
    try:
        features_df = collect_candidates(
            candidates_df, None, self.precursors_flat_df, self.rt_column, self.mobility_column, self.precursor_mz_column,
            row_maps=(soa["order"], soa["prec_row"]), sequence_counts=self._sequence_counts(), compact=comp)
    finally:
        t_3 = time.perf_counter()
        worker.join()
    if "exc" in frag_box:
        raise frag_box["exc"]
    t_4 = time.perf_counter()
    self.last_timings = {"assemble_ms": (t_1 - t_0) * 1e3, "score_ms": (t_2 - t_1) * 1e3,
                         "collect_candidates_ms": (t_3 - t_2) * 1e3, "collect_fragments_ms": frag_box["ms"],
                         "collect_ms": (t_4 - t_2) * 1e3, "total_ms": (t_4 - t_0) * 1e3,
                         "wire_bytes": int(comp["features"].nbytes + 5 * len(comp["row"]) + 22 * len(comp["fragment_row"]))}
    logger.info("Finished candidate scoring")
    return features_df, frag_box["df"]


HipCandidateScoring._call_compact = _call_compact


def calculate_score_groups(input_df: pd.DataFrame, group_channels: bool = False) -> pd.DataFrame:
    """``score_group_idx`` for every row (reference: scoring/utils.py:269-410).

    Rows come back ordered by (elution group, decoy, rank, precursor); without ``group_channels``
    every row is its own group, with it one group spans the rows that share elution group, decoy
    flag and rank (the label channels of one peptide)."""
    n = len(input_df)
    rank = input_df["rank"].values if "rank" in input_df.columns else np.zeros(n, dtype=np.uint32)
    eg, dc, pi = (input_df[c].values for c in ("elution_group_idx", "decoy", "precursor_idx"))
    order = np.lexsort((pi, rank, dc, eg))
    out = input_df.iloc[order].reset_index(drop=True)
    if group_channels and n:
        key = np.stack([eg[order], dc[order], rank[order]])
        new_group = np.r_[False, (key[:, 1:] != key[:, :-1]).any(axis=0)]
        out["score_group_idx"] = np.cumsum(new_group).astype(np.uint32)
    else:
        out["score_group_idx"] = np.arange(n, dtype=np.uint32)
    return out


def _first_valid_per_group(values: np.ndarray, group_start: np.ndarray, n: int) -> np.ndarray:
    """First non-null entry of every group of a grouped-and-ordered column (GroupBy.first)."""
    null = pd.isnull(values)
    if not null.any():
        return values[group_start]
    pos = np.where(null, n, np.arange(n))
    first = np.minimum.reduceat(pos, group_start)
    stop = np.r_[group_start[1:], n]
    out = values[np.minimum(first, n - 1)].copy()
    if (first >= stop).any():  # a group without any value keeps the null
        out = out.astype(object) if out.dtype.kind not in "fO" else out
        out[first >= stop] = np.nan
    return out


def multiplex_candidates(
    candidates_df: pd.DataFrame,
    precursors_flat_df: pd.DataFrame,
    remove_decoys: bool = True,
    channels: list[int] | None = None,
) -> pd.DataFrame:
    """Copy the best candidate of every elution group to all its label channels
    (reference: scoring/utils.py:114-200).

    The best candidate of a group is the one with the lowest ``proba`` (ties: lowest precursor_idx);
    it is handed, without its own precursor_idx / channel, to every library precursor of the group
    whose channel is in ``channels`` (targets only unless ``remove_decoys`` is off).  Rows keep the
    order of the library table."""
    channels = [0, 4, 8, 12] if channels is None else channels
    cand = candidates_df
    lib = precursors_flat_df
    if remove_decoys:
        lib = lib[lib["decoy"].values == 0]
        if "decoy" in cand.columns:
            cand = cand[cand["decoy"].values == 0]
    n = len(cand)
    eg = cand["elution_group_idx"].values
    order = np.lexsort((cand["precursor_idx"].values, cand["proba"].values, eg))
    eg_sorted = eg[order]
    group_start = np.flatnonzero(np.r_[True, eg_sorted[1:] != eg_sorted[:-1]]) if n else np.zeros(0, np.int64)
    best_eg = eg_sorted[group_start]

    lib_eg = lib["elution_group_idx"].values
    at = np.searchsorted(best_eg, lib_eg)
    at_c = np.minimum(at, max(len(best_eg) - 1, 0))
    keep = np.isin(lib["channel"].values, channels)
    keep &= (best_eg[at_c] == lib_eg) if len(best_eg) else False
    out = pd.DataFrame({c: lib[c].values[keep] for c in ("elution_group_idx", "precursor_idx", "channel")})
    group_of_row = at_c[keep]
    for col in cand.columns:
        if col in ("elution_group_idx", "precursor_idx", "channel"):
            continue
        best = _first_valid_per_group(cand[col].values[order], group_start, n) if n else cand[col].values[:0]
        out[col] = best[group_of_row]
    return out


def requantify_multiplexed(dia_data, psm_df: pd.DataFrame, precursors_flat: pd.DataFrame, fragments_flat: pd.DataFrame,
                           channels: list[int], reference_channel: int, experimental_xic: bool, column_names: dict,
                           device: int | None = None):
    """The scoring half of multiplex requantification
    (multiplexing_requantification_handler.py:95-140): best candidate of every elution group copied
    to all ``channels``, the channel copies scored as one score group gated on the reference
    channel.  Returns ``(features_df, fragments_df)``; the q-values are the FDR manager's business."""
    cols = ["elution_group_idx", "precursor_idx", "rank", "scan_start", "scan_stop", "scan_center", "frame_start",
            "frame_stop", "frame_center", "proba"]
    multiplexed = multiplex_candidates(psm_df[cols], precursors_flat, channels=channels)
    multiplexed["rank"] = 0
    config = CandidateScoringConfig()
    config.update(dict(score_grouped=True, exclude_shared_ions=True, reference_channel=int(reference_channel),
                       experimental_xic=bool(experimental_xic)))
    scoring = HipCandidateScoring(dia_data=dia_data, precursors_flat=precursors_flat, fragments_flat=fragments_flat,
                                  config=config, device=device, **column_names)
    return scoring(multiplexed)
