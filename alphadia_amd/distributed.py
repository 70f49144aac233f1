"""Candidate sharding across the GPUs of one node and table reassembly.

Score groups are independent (reference: disjoint ``output_idx`` rows,
search/scoring/containers/score_group.py:66-75), so the candidate table is cut
into contiguous score-group ranges, one per rank; the run and the library are
replicated in every GPU's HBM.  Each rank fills one packed device buffer holding
its slice of every OutputPsmDF table; ONE all-gather reassembles the tables on
every rank: RCCL over xGMI behind the C ABI (``adh_comm_*``, csrc/adh_comm.hip).
This module holds what the ranks have to agree on - shard bounds, the packed
layout (taken from the library), the columns that do not travel and how they
are rebuilt.  No torch here; the gloo transport the 2-rank CPU tests move the
same packed buffers with lives in tests/torch_transport.py.
"""

from __future__ import annotations

import numpy as np

from alphadia_amd import _abi

_ALIGN = 256


def shard_bounds(score_group_idx: np.ndarray, rank: int, world: int) -> tuple[int, int]:
    """Row range [a, b) of ``rank``: contiguous score groups, balanced by group count.

    ``score_group_idx`` is the non-decreasing group id per candidate row.
    """
    n = len(score_group_idx)
    if n == 0:
        return 0, 0
    n_groups = int(score_group_idx[-1]) + 1
    g0 = (n_groups * rank) // world
    g1 = (n_groups * (rank + 1)) // world
    a = int(np.searchsorted(score_group_idx, g0, side="left"))
    b = int(np.searchsorted(score_group_idx, g1, side="left"))
    return a, b


def slice_soa(soa: dict, a: int, b: int) -> dict:
    return {k: (v[a:b] if isinstance(v, np.ndarray) and v.shape[:1] == (len(soa["precursor_idx"]),) else v)
            for k, v in soa.items()}


# Columns every rank can rebuild from what it already holds stay out of the all-gather: candidate ids
# (from the candidate table) and the library columns of the fragment tables (from the staged library
# and ``fragment_lib_slot``, 2 bytes per slot instead of 13): 449 of 646 bytes per candidate travel
# at top_k = 12.
LIBRARY_COLUMNS = ("fragment_mz_library", "fragment_mz", "fragment_position", "fragment_number", "fragment_type",
                   "fragment_charge", "fragment_loss_type")
LOCAL_COLUMNS = ("precursor_idx", "rank", "fragment_precursor_idx", "fragment_rank", *LIBRARY_COLUMNS,
                 "stat_matched_peaks")


_DTYPES = {("valid", 1): np.uint8}


def packed_layout(n_rows: int, top_k: int, with_stats: bool = True):
    """Byte offsets of every OutputPsmDF table inside one packed buffer of ``n_rows`` rows, AS THE LIBRARY
    LAYS IT OUT (``adh_table_layout``: the buffer behind ``adh_get_device_tables`` / ``adh_comm_gathered``).

    The computed tables come first (the "wire" prefix that is all-gathered and copied to the host);
    ``LOCAL_COLUMNS`` follow.  Returns ({name: (offset, shape, dtype)}, total bytes)."""
    from alphadia_amd import runtime

    fields, total, _ = runtime.table_layout(n_rows, top_k)
    dtypes = {name: np.dtype(dt) for name, (_, dt) in _abi.output_shapes(1, 1).items()}
    dtypes["fragment_lib_slot"] = np.dtype(np.uint16)
    dtypes["stat_matched_peaks"] = np.dtype(np.uint32)
    offsets = {}
    for f in fields:
        if f["name"] == "stat_matched_peaks" and not with_stats:
            continue
        dt = dtypes[f["name"]]
        assert dt.itemsize == f["elem_bytes"], f["name"]
        shape = (n_rows,) if f["row_elems"] == 1 and f["name"] != "features" else (n_rows, f["row_elems"])
        offsets[f["name"]] = (f["offset"], shape, dt)
    return offsets, total


def wire_bytes(offsets: dict) -> int:
    """Length of the all-gathered prefix of a packed buffer."""
    return min(off for name, (off, _, _) in offsets.items() if name in LOCAL_COLUMNS)


def rebuild_local_columns(tables: dict, precursor_idx: np.ndarray, rank: np.ndarray, flags: np.ndarray | None = None,
                          matched_peaks: np.ndarray | None = None, frag_start: np.ndarray | None = None,
                          fragment_cols: tuple | None = None) -> dict:
    """Complete gathered wire tables with the columns that did not travel: candidate ids from the
    candidate table (zero for skipped score groups, score_group.py:50-64), their copies in the
    filled rows of the fragment tables (candidate.py:403-481), and the library columns of the
    filled slots from the library (``fragment_cols`` = ``scoring.fragment_columns``, ``frag_start``
    = first library row of every candidate)."""
    n = tables["valid"].shape[0]
    pi = np.asarray(precursor_idx, dtype=np.uint32)[:n].copy()
    rk = np.asarray(rank, dtype=np.uint8)[:n].copy()
    if flags is not None:
        skip = (np.asarray(flags)[:n] & _abi.FLAG_SKIP) != 0
        pi[skip] = 0
        rk[skip] = 0
    slot = tables["fragment_lib_slot"]
    filled = slot != 0
    out = dict(tables)
    out["precursor_idx"] = pi
    out["rank"] = rk
    out["fragment_precursor_idx"] = np.where(filled, pi[:, None], 0).astype(np.uint32)
    out["fragment_rank"] = np.where(filled, rk[:, None], 0).astype(np.uint8)
    if any(c not in tables for c in LIBRARY_COLUMNS):
        if frag_start is None or fragment_cols is None:
            raise ValueError("the library columns did not travel: pass frag_start and fragment_cols")
        mz_library, mz, _, type_, loss_type, charge, number, position, _ = fragment_cols
        src = dict(fragment_mz_library=(mz_library, np.float32), fragment_mz=(mz, np.float32),
                   fragment_position=(position, np.uint8), fragment_number=(number, np.uint8),
                   fragment_type=(type_, np.uint8), fragment_charge=(charge, np.uint8),
                   fragment_loss_type=(loss_type, np.uint8))
        lib_row = np.asarray(frag_start, dtype=np.int64)[:n, None] + slot.astype(np.int64) - 1
        lib_row = np.where(filled, lib_row, 0)
        for name, (col, dt) in src.items():
            col = np.asarray(col)
            vals = col[lib_row] if len(col) else np.zeros(lib_row.shape, dt)
            out[name] = np.where(filled, vals, 0).astype(dt)
    if matched_peaks is not None:
        out["stat_matched_peaks"] = np.asarray(matched_peaks, dtype=np.uint32)[:n]
    return out


def merge_gathered(tables_per_rank: list[dict], rows_per_rank: list[int]) -> dict:
    """Concatenate the live rows of every rank's tables in rank order."""
    out = {}
    for name in tables_per_rank[0]:
        out[name] = np.concatenate(
            [t[name][:r] for t, r in zip(tables_per_rank, rows_per_rank, strict=True)], axis=0
        )
    return out


# ---------------------------------------------------------------------------
# the two other stages of the path shard without any exchange beyond one gather


def precursor_bounds(n_precursors: int, rank: int, world: int) -> tuple[int, int]:
    """Candidate selection is independent per precursor (selection.py:620-660): contiguous,
    balanced ranges of the precursor table sorted by precursor_idx."""
    base, rem = divmod(int(n_precursors), int(world))
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


def window_owner(n_windows: int, world: int) -> np.ndarray:
    """Fragment competition is independent per DIA window (fragcomp.py:204-229,278): windows are
    dealt round-robin, rank r owns the windows w with ``owner[w] == r``."""
    return (np.arange(int(n_windows)) % int(world)).astype(np.int32)
