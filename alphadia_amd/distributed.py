"""Candidate sharding across the GPUs of one node and table reassembly.

Score groups are independent (reference: disjoint ``output_idx`` rows,
search/scoring/containers/score_group.py:66-75), so the candidate table is cut
into contiguous score-group ranges, one per rank; the run and the library are
replicated in every GPU's HBM.  Each rank fills one packed device buffer holding
its slice of every OutputPsmDF table; ONE all-gather (RCCL over xGMI when the
process group backend is "nccl", gloo on CPU in the tests) reassembles the
tables on every rank.  PyTorch is used here only for device memory, streams and
``torch.distributed``.
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


def packed_layout(n_rows: int, top_k: int, with_stats: bool = True):
    """Byte offsets of every OutputPsmDF table inside one packed buffer of ``n_rows`` rows.

    The computed tables come first (the "wire" prefix that is all-gathered); ``LOCAL_COLUMNS``
    follow.  ``wire_bytes`` gives the length of the prefix."""
    shapes = dict(_abi.output_shapes(n_rows, top_k))
    shapes["fragment_lib_slot"] = ((n_rows, top_k), np.uint16)
    if with_stats:
        shapes["stat_matched_peaks"] = ((n_rows,), np.uint32)
    order = [k for k in shapes if k not in LOCAL_COLUMNS] + [k for k in shapes if k in LOCAL_COLUMNS]
    offsets = {}
    off = 0
    for name in order:
        shape, dt = shapes[name]
        offsets[name] = (off, shape, np.dtype(dt))
        off += int(np.prod(shape)) * np.dtype(dt).itemsize
        off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
    return offsets, off


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


class DeviceTables:
    """OutputPsmDF tables of one rank as ONE packed ``torch.uint8`` device buffer."""

    def __init__(self, n_rows: int, top_k: int, device, with_stats: bool = True):
        import torch

        self.n_rows = int(n_rows)
        self.top_k = int(top_k)
        self.offsets, self.nbytes = packed_layout(self.n_rows, self.top_k, with_stats)
        self.wire_nbytes = wire_bytes(self.offsets)
        self.buffer = torch.zeros(max(self.nbytes, 1), dtype=torch.uint8, device=device)
        self._with_stats = with_stats

    def zero_(self):
        self.buffer.zero_()

    def as_output(self, n: int | None = None) -> _abi.Output:
        """``adh_output_t`` of device pointers; ``n`` <= n_rows is the live row count."""
        base = self.buffer.data_ptr()
        ptrs = {k: base + off for k, (off, _, _) in self.offsets.items()}
        stats = ptrs.pop("stat_matched_peaks", 0)
        slots = ptrs.pop("fragment_lib_slot", 0)
        return _abi.output_from_device_pointers(
            self.n_rows if n is None else int(n), self.top_k, ptrs, stats_ptr=stats, slot_ptr=slots
        )

    def load_host(self, arrays: dict) -> None:
        """Fill the packed buffer from host tables with at most ``n_rows`` rows."""
        import torch

        raw = np.zeros(self.nbytes, dtype=np.uint8)
        for name, (off, shape, dt) in self.offsets.items():
            if name not in arrays:
                continue
            a = np.ascontiguousarray(arrays[name]).astype(dt, copy=False)
            flat = a.reshape(-1).view(np.uint8)
            raw[off : off + flat.size] = flat
        self.buffer.copy_(torch.from_numpy(raw).to(self.buffer.device))

    @property
    def wire(self):
        """The prefix of the packed buffer that is all-gathered (computed tables only)."""
        return self.buffer[: self.wire_nbytes]

    def to_host(self, buffer=None) -> dict:
        """Unpack a packed buffer (this rank's, or one gathered slice) into numpy tables; a
        wire-only slice yields the computed tables (see ``rebuild_local_columns``)."""
        raw = (self.buffer if buffer is None else buffer).cpu().numpy()
        out = {}
        for name, (off, shape, dt) in self.offsets.items():
            cnt = int(np.prod(shape))
            if off + cnt * dt.itemsize > raw.shape[0]:
                continue
            out[name] = raw[off : off + cnt * dt.itemsize].view(dt).reshape(shape).copy()
        return out


def all_gather_tables(local, world: int, group=None):
    """One all-gather of the packed per-rank buffers -> [world, nbytes] tensor."""
    import torch
    import torch.distributed as dist

    gathered = torch.empty((world, local.shape[0]), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(gathered, local, group=group)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        dist.all_gather(list(gathered.unbind(0)), local, group=group)
    return gathered


class PipelinedGather:
    """Double-buffered all-gather of the packed tables: the collective of batch i runs while the
    kernels of batch i+1 fill the other buffer (RCCL works on its own stream; the score kernels
    are bound by VALU / HBM latency, the gather by xGMI links, so the two overlap well).

        pg = PipelinedGather(n_rows, top_k, device, world)
        for batch in batches:
            tables = pg.begin()            # waits until the gather that last read this slot is done
            ... enqueue zero_() + scoring into `tables` on the current stream ...
            pg.end()                       # starts the gather of this slot, returns immediately
        gathered = pg.finish()             # [world, wire_nbytes] of the last batch, all work complete
    """

    def __init__(self, n_rows: int, top_k: int, device, world: int, with_stats: bool = True, group=None):
        import torch

        self.world = int(world)
        self.group = group
        self.tables = [DeviceTables(n_rows, top_k, device, with_stats) for _ in range(2)]
        self.gathered = [
            torch.empty((self.world, self.tables[0].wire_nbytes), dtype=torch.uint8, device=device)
            for _ in range(2)
        ]
        self.pending = [None, None]
        self.slot = 1
        self.overlap = True

    def begin(self) -> DeviceTables:
        self.slot ^= 1
        w = self.pending[self.slot]
        if w is not None:
            w.wait()  # orders the current stream after that collective
            self.pending[self.slot] = None
        return self.tables[self.slot]

    def end(self):
        import torch.distributed as dist

        if self.world <= 1:
            return
        local, out = self.tables[self.slot].wire, self.gathered[self.slot]
        if self.overlap:
            try:
                self.pending[self.slot] = dist.all_gather_into_tensor(out, local, group=self.group, async_op=True)
                return
            except (RuntimeError, NotImplementedError):
                self.overlap = False  # e.g. a backend without the flat variant: synchronous path below
        out.copy_(all_gather_tables(local, self.world, group=self.group))

    def finish(self):
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None
        return self.gathered[self.slot] if self.world > 1 else self.tables[self.slot].buffer


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


def all_gather_rows(local_rows: np.ndarray, n_rows_per_rank: list[int], group=None) -> np.ndarray:
    """Gather variable-length row blocks (any dtype, any trailing shape) from all ranks in rank
    order with ONE collective: blocks are padded to the longest one."""
    import torch
    import torch.distributed as dist

    world = len(n_rows_per_rank)
    width = int(max(n_rows_per_rank)) if n_rows_per_rank else 0
    local_rows = np.ascontiguousarray(local_rows)
    trailing = local_rows.shape[1:]
    row_bytes = int(np.prod(trailing, dtype=np.int64)) * local_rows.dtype.itemsize
    buf = np.zeros(width * row_bytes, dtype=np.uint8)
    raw = local_rows.view(np.uint8).reshape(-1)
    buf[: raw.shape[0]] = raw
    t = torch.from_numpy(buf)
    gathered = all_gather_tables(t, world, group=group).numpy()
    parts = [
        gathered[r, : n_rows_per_rank[r] * row_bytes].view(local_rows.dtype).reshape((n_rows_per_rank[r],) + trailing)
        for r in range(world)
    ]
    return np.concatenate(parts, axis=0)
