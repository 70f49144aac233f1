"""TEST INFRASTRUCTURE: a torch / gloo transport for the packed score tables.

The product's collective is RCCL behind the C ABI (alphadia_amd/csrc/adh_comm.hip) and needs GPUs.  The
2-rank CPU tests move the SAME packed buffers (layout from ``adh_table_layout`` through
``alphadia_amd.distributed.packed_layout``) with ``torch.distributed`` over gloo, so that everything the ranks
have to agree on - shard bounds, layout, wire prefix, rebuilt columns - is exercised without a GPU.
"""

from __future__ import annotations

import numpy as np

from alphadia_amd import _abi
from alphadia_amd.distributed import packed_layout, wire_bytes


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
