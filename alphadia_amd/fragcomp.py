"""Fragment competition on the GPU.

Drop-in for ``alphadia.fragcomp.fragcomp.FragmentCompetition``
(alphadia/fragcomp/fragcomp.py:146-299): same constructor, same ``__call__``
signature and the same returned ``psm_df[psm_df["valid"]]``.  The pandas
preparation (hashing, fragment start/stop indices, window index, sort) follows
the reference line by line; the nested competition loops
(``_compete_for_fragments``, fragcomp.py:51-143) run in a HIP kernel, one
workgroup per DIA window.
"""

from __future__ import annotations

import logging

import numpy as np
import pandas as pd

logger = logging.getLogger(__name__)


def candidate_hash(precursor_idx: np.ndarray, rank: np.ndarray) -> np.ndarray:
    """64-bit hash: precursor_idx in the low 32 bits, rank above (fragcomp/utils.py:48-58)."""
    return (
        np.asarray(precursor_idx).astype(np.int64) + (np.asarray(rank).astype(np.int64) << 32)
    ).astype(np.uint64)


def add_frag_start_stop_idx(psm_df: pd.DataFrame, frag_df: pd.DataFrame) -> pd.DataFrame:
    """fragcomp/utils.py:11-45"""
    if "_frag_start_idx" in psm_df.columns and "_frag_stop_idx" in psm_df.columns:
        logger.warning(
            "Fragment start and stop indices already present in PSM dataframe. Skipping."
        )
        return psm_df
    frag_df["frag_idx"] = np.arange(len(frag_df))
    index_df = frag_df.groupby("_candidate_idx", as_index=False).agg(
        _frag_start_idx=pd.NamedAgg("frag_idx", "min"),
        _frag_stop_idx=pd.NamedAgg("frag_idx", "max"),
    )
    index_df["_frag_stop_idx"] += 1
    return psm_df.merge(index_df, "inner", on="_candidate_idx")


class FragmentCompetition:
    """Remove PSMs that share fragments with better PSMs (GPU implementation)."""

    def __init__(self, rt_tol_seconds: int = 3, mass_tol_ppm: int = 15, thread_count: int = 8,
                 device: int | None = None):
        self.rt_tol_seconds = rt_tol_seconds
        self.mass_tol_ppm = mass_tol_ppm
        self.thread_count = thread_count  # CPU knob of the reference; unused on the GPU
        self.device = device

    @staticmethod
    def _add_window_idx(psm_df: pd.DataFrame, cycle: np.ndarray) -> pd.DataFrame:
        """fragcomp.py:170-202"""
        if "window_idx" in psm_df.columns:
            logger.warning("Window index already present in PSM dataframe. Skipping.")
            return psm_df
        lower_limit = np.min(cycle[0, :, :, 0], axis=1, keepdims=True).T
        upper_limit = np.max(cycle[0, :, :, 1], axis=1, keepdims=True).T
        mz = np.expand_dims(psm_df["mz_observed"].values, axis=-1)
        idx = (mz >= lower_limit) & (mz < upper_limit)
        psm_df["window_idx"] = np.argmax(idx, axis=1)
        return psm_df

    @staticmethod
    def _get_thread_plan_df(psm_df: pd.DataFrame) -> pd.DataFrame:
        """fragcomp.py:204-229"""
        psm_df["_thread_idx"] = np.arange(len(psm_df))
        index_df = psm_df.groupby("window_idx", as_index=False).agg(
            start_idx=pd.NamedAgg("_thread_idx", "min"),
            stop_idx=pd.NamedAgg("_thread_idx", "max"),
        )
        index_df["stop_idx"] += 1
        psm_df.drop(columns=["_thread_idx"], inplace=True)
        return index_df

    def __call__(self, psm_df: pd.DataFrame, frag_df: pd.DataFrame, cycle: np.ndarray) -> pd.DataFrame:
        from alphadia_amd import runtime  # raises when the HIP library is missing

        psm_df["_candidate_idx"] = candidate_hash(psm_df["precursor_idx"].values, psm_df["rank"].values)
        frag_df["_candidate_idx"] = candidate_hash(frag_df["precursor_idx"].values, frag_df["rank"].values)
        psm_df = add_frag_start_stop_idx(psm_df, frag_df)
        psm_df = self._add_window_idx(psm_df, cycle)
        # important to sort by window_idx and proba (fragcomp.py:268-270)
        psm_df.sort_values(by=["window_idx", "proba", "precursor_idx"], inplace=True)
        thread_plan_df = self._get_thread_plan_df(psm_df)
        ctx = runtime.get_context(self.device)
        valid = ctx.fragcomp(
            thread_plan_df["start_idx"].values,
            thread_plan_df["stop_idx"].values,
            psm_df["rt_observed"].values,
            psm_df["_frag_start_idx"].values,
            psm_df["_frag_stop_idx"].values,
            frag_df["mz_observed"].values,
            self.rt_tol_seconds,
            self.mass_tol_ppm,
        )
        psm_df["valid"] = valid
        psm_df.drop(columns=["_frag_start_idx", "_frag_stop_idx", "window_idx"], inplace=True)
        return psm_df[psm_df["valid"]]
