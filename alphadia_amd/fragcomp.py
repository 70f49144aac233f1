"""Fragment competition on the GPU.

Drop-in for ``alphadia.fragcomp.fragcomp.FragmentCompetition``
(alphadia/fragcomp/fragcomp.py:146-299): same constructor, same ``__call__`` signature, same
surviving rows in the same order.  The reference prepares the competition with a chain of pandas
group-bys and merges (fragcomp.py:170-229,268-289; fragcomp/utils.py:11-58); here the whole
preparation is ONE numpy pass over plain arrays (:func:`competition_plan`): candidate keys, the
fragment range of every PSM, the DIA window of every PSM, the processing order and the row range
of every window.  The nested competition loops (``_compete_for_fragments``, fragcomp.py:51-143)
run in a HIP kernel, one workgroup per DIA window.
"""

from __future__ import annotations

import logging
import os
from dataclasses import dataclass

import numpy as np
import pandas as pd

logger = logging.getLogger(__name__)


def candidate_hash(precursor_idx: np.ndarray, rank: np.ndarray) -> np.ndarray:
    """The reference's candidate key (fragcomp/utils.py:48-58): rank << 32 | precursor_idx."""
    key = np.asarray(precursor_idx).astype(np.uint64)
    rank = np.asarray(rank)
    if rank.size and rank.any():  # (a table of first-ranked candidates only: the key is the precursor index)
        key |= rank.astype(np.uint64) << np.uint64(32)
    return key


@dataclass
class CompetitionPlan:
    """Everything the competition kernel needs, in processing order."""

    rows: np.ndarray          # PSM rows (positions in the caller's frame) in processing order
    key: np.ndarray           # candidate key of those rows
    frag_start: np.ndarray    # [start, stop) into the fragment table, per processed row
    frag_stop: np.ndarray
    window: np.ndarray        # DIA window of every processed row (non-decreasing)
    window_start: np.ndarray  # row range of every occupied window
    window_stop: np.ndarray


def competition_plan(precursor_idx, rank, mz_observed, proba, frag_precursor_idx, frag_rank,
                     cycle: np.ndarray) -> CompetitionPlan:
    """What ``FragmentCompetition.__call__`` derives before it competes (fragcomp.py:268-289).

    * a PSM owns the fragment rows that carry its (precursor_idx, rank) key: first such row to last
      such row + 1 (``add_frag_start_stop_idx``, fragcomp/utils.py:11-45); PSMs without fragment
      rows leave the competition (the reference's inner merge)
    * its DIA window is the first cycle row whose [lowest, highest) isolation limit holds its
      observed m/z, row 0 when none does (``_add_window_idx``, fragcomp.py:170-202)
    * rows are processed window by window, best (lowest) ``proba`` first, ties by precursor_idx,
      then by input position (the reference's stable multi-column sort)
    """
    key = candidate_hash(precursor_idx, rank)
    fkey = candidate_hash(frag_precursor_idx, frag_rank)
    n_frag = fkey.shape[0]
    # first / last fragment row of every distinct key.  The fragment table of collect_fragments holds a candidate's
    # rows next to each other: then the runs of equal keys ARE the groups, and only their first keys (one per
    # candidate, a twelfth of the rows) have to be sorted - checked, not assumed: a key that starts two runs sends the
    # table down the general path (a stable sort of all rows)
    lo = hi = uniq = None
    if n_frag:
        starts = np.flatnonzero(np.concatenate(([True], fkey[1:] != fkey[:-1])))
        run_keys = fkey[starts]
        by_run = np.argsort(run_keys, kind="stable")
        sorted_runs = run_keys[by_run]
        if len(sorted_runs) < 2 or (sorted_runs[1:] != sorted_runs[:-1]).all():
            uniq = sorted_runs
            lo = starts[by_run]
            hi = np.append(starts[1:], n_frag)[by_run]
    if uniq is None:
        by_key = np.argsort(fkey, kind="stable")
        sorted_keys = fkey[by_key]
        is_first = np.ones(n_frag, dtype=bool)
        is_first[1:] = sorted_keys[1:] != sorted_keys[:-1]
        group_first = np.flatnonzero(is_first)
        uniq = sorted_keys[group_first]
        # stable sort: inside a group the rows appear in table order -> first is the minimum, last the maximum
        group_last = np.append(group_first[1:], n_frag) - 1
        lo = by_key[group_first] if n_frag else np.zeros(0, np.int64)
        hi = by_key[group_last] + 1 if n_frag else np.zeros(0, np.int64)
    # (sorted needles: a binary search over a million keys in random order is a cache miss per step)
    key_order = np.argsort(key, kind="stable")
    at = np.empty(len(key), dtype=np.int64)
    at[key_order] = np.searchsorted(uniq, key[key_order])
    at_c = np.minimum(at, max(len(uniq) - 1, 0))
    has = (uniq[at_c] == key) if len(uniq) else np.zeros(len(key), dtype=bool)
    rows = np.flatnonzero(has)

    # DIA window: per cycle row the extreme isolation limits over its scans
    lower = cycle[0, :, :, 0].min(axis=1)
    upper = cycle[0, :, :, 1].max(axis=1)
    mz = np.asarray(mz_observed)[rows]
    window = np.zeros(len(rows), dtype=np.int64)
    real = np.flatnonzero(upper > lower)  # (the MS1 row is (-1, -1): it holds no m/z)
    by_lower = real[np.argsort(lower[real], kind="stable")]
    if len(by_lower) and (lower[by_lower][1:] >= upper[by_lower][:-1]).all():
        # disjoint isolation windows (every DIA scheme of the reference's tests): the window that holds an m/z is the
        # last one that starts at or below it - one binary search instead of a pass over the table per window
        pos = np.searchsorted(lower[by_lower], mz, side="right") - 1
        w_at = by_lower[np.maximum(pos, 0)]
        inside = (pos >= 0) & (mz < upper[w_at])
        window[inside] = w_at[inside]
    else:
        unassigned = np.ones(len(rows), dtype=bool)
        for w in range(len(lower)):
            inside = unassigned & (mz >= lower[w]) & (mz < upper[w])
            window[inside] = w
            unassigned &= ~inside

    # (window, proba) as ONE 64-bit key where the probabilities are non-negative float32 (their bit patterns order like
    # the values): two sort keys instead of three
    pr = np.asarray(proba)[rows]
    if pr.dtype == np.float32 and len(pr) and not (pr < 0).any() and not np.isnan(pr).any():
        k64 = (window.astype(np.uint64) << np.uint64(32)) | (pr + np.float32(0.0)).view(np.uint32).astype(np.uint64)
        order = np.lexsort((np.asarray(precursor_idx)[rows], k64))
    else:
        order = np.lexsort((np.asarray(precursor_idx)[rows], pr, window))
    rows = rows[order]
    window = window[order]
    occupied, first = np.unique(window, return_index=True)
    del occupied
    stop = np.append(first[1:], len(window)) if len(first) else first
    return CompetitionPlan(rows=rows, key=key[rows], frag_start=lo[at_c[rows]].astype(np.int64),
                           frag_stop=hi[at_c[rows]].astype(np.int64), window=window,
                           window_start=first.astype(np.int64), window_stop=stop.astype(np.int64))


def compete_sharded(plan: CompetitionPlan, rt, fragment_mz, rt_tol_seconds, mass_tol_ppm, rank: int, world: int,
                    compete, gather_rows) -> np.ndarray:
    """The competition of one run on ``world`` GPUs: DIA windows are independent (fragcomp.py:204-229,278 gives
    every window to its own thread), so rank r competes in the windows ``window_owner == r`` and ONE gather of
    the flags of everybody's own rows puts the column together on every rank.

    ``compete(window_start, window_stop, rt, frag_start, frag_stop, fragment_mz, rt_tol, mass_tol)`` -> bool per
    processed row (rows outside the given windows come back True); ``gather_rows(local, rows_per_rank)`` ->
    concatenation in rank order (``runtime.Context.all_gather_rows``; the CPU tests pass the gloo transport)."""
    from alphadia_amd.distributed import window_owner

    n_win = len(plan.window_start)
    if world <= 1 or n_win == 0:
        return compete(plan.window_start, plan.window_stop, rt, plan.frag_start, plan.frag_stop, fragment_mz,
                       rt_tol_seconds, mass_tol_ppm)
    owner = window_owner(n_win, world)
    mine = np.flatnonzero(owner == rank)
    valid = compete(plan.window_start[mine], plan.window_stop[mine], rt, plan.frag_start, plan.frag_stop, fragment_mz,
                    rt_tol_seconds, mass_tol_ppm)
    # what travels: the flags of the rows of a rank's own windows, window after window
    sizes = (plan.window_stop - plan.window_start).astype(np.int64)
    rows_per_rank = [int(sizes[owner == r].sum()) for r in range(world)]
    local = np.concatenate([valid[plan.window_start[w]:plan.window_stop[w]] for w in mine]) if len(mine) else np.zeros(0, bool)
    flat = gather_rows(np.ascontiguousarray(local, dtype=np.uint8), rows_per_rank).astype(bool)
    out = np.ones(len(plan.rows), dtype=bool)
    at = 0
    for r in range(world):
        for w in np.flatnonzero(owner == r):
            n = int(sizes[w])
            out[plan.window_start[w]:plan.window_stop[w]] = flat[at:at + n]
            at += n
    return out


class FragmentCompetition:
    """Remove PSMs that share fragments with better PSMs (GPU implementation).

    ``rank`` / ``world`` (default: the communicator attached to the context, i.e. one rank without one): with
    several ranks every rank competes in its own DIA windows and the flags are gathered once
    (:func:`compete_sharded`); every rank returns the complete frame."""

    def __init__(self, rt_tol_seconds: int = 3, mass_tol_ppm: int = 15, thread_count: int = 8,
                 device: int | None = None, rank: int | None = None, world: int | None = None):
        self.rt_tol_seconds = rt_tol_seconds
        self.mass_tol_ppm = mass_tol_ppm
        self.thread_count = thread_count  # CPU knob of the reference; unused on the GPU
        self.device = device
        self.rank, self.world = rank, world

    def __call__(self, psm_df: pd.DataFrame, frag_df: pd.DataFrame, cycle: np.ndarray) -> pd.DataFrame:
        from alphadia_amd import runtime  # raises when the HIP library is missing

        ctx = runtime.get_context(self.device)
        rank, world = ctx.comm_info() if self.world is None else (int(self.rank or 0), int(self.world))
        # one rank, float32 probabilities (what a classifier's predict_proba gives): the preparation runs on the device
        # too (adh_fragcomp_frames: 1e6 PSMs / 12 M fragment rows in tens of ms where the NumPy plan below takes 360)
        if (world <= 1 and psm_df["proba"].dtype == np.float32 and psm_df["mz_observed"].dtype == np.float32
                and not os.environ.get("ADH_FRAGCOMP_HOST_PLAN")):
            done = ctx.fragcomp_frames(
                psm_df["precursor_idx"].values, psm_df["rank"].values, psm_df["mz_observed"].values,
                psm_df["rt_observed"].values, psm_df["proba"].values, frag_df["precursor_idx"].values,
                frag_df["rank"].values, frag_df["mz_observed"].values, cycle, self.rt_tol_seconds, self.mass_tol_ppm)
            if done is not None:
                rows, valid = done
                kept = rows[valid]
                out = psm_df.iloc[kept].copy()
                out["_candidate_idx"] = candidate_hash(psm_df["precursor_idx"].values[kept], psm_df["rank"].values[kept])
                out["valid"] = True
                return out
        plan = competition_plan(
            psm_df["precursor_idx"].values, psm_df["rank"].values, psm_df["mz_observed"].values,
            psm_df["proba"].values, frag_df["precursor_idx"].values, frag_df["rank"].values, cycle,
        )
        if world > 1 and ctx.comm_info()[1] != world:
            raise runtime.HipBackendError(f"FragmentCompetition(world={world}) needs a communicator of {world} ranks "
                                          "on the context (Context.comm_init)")
        valid = compete_sharded(plan, psm_df["rt_observed"].values[plan.rows], frag_df["mz_observed"].values,
                                self.rt_tol_seconds, self.mass_tol_ppm, rank, world, ctx.fragcomp, ctx.all_gather_rows)
        # the frame the reference returns: surviving rows in processing order, with the candidate
        # key and the (all-true) flag column it leaves behind (fragcomp.py:291-299)
        out = psm_df.iloc[plan.rows[valid]].copy()
        out["_candidate_idx"] = plan.key[valid]
        out["valid"] = True
        return out
