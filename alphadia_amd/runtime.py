"""ctypes binding of ``libalphadia_hip.so`` (the C ABI in include/alphadia_hip.h).

There is deliberately no fallback: importing this module without the built
library, or creating a context without a GPU, raises.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from alphadia_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADH_LIB_PATH") or os.path.join(_HERE, "libalphadia_hip.so")  # (developer switch: an experimental build)

EXPORTED_SYMBOLS = [
    "adh_last_error",
    "adh_device_count",
    "adh_create",
    "adh_destroy",
    "adh_stage_alpharaw",
    "adh_stage_timstof",
    "adh_stage_fragments",
    "adh_score_candidates",
    "adh_score_candidates_compact",
    "adh_host_take_objects",
    "adh_trim_device_cache",
    "adh_host_threads",
    "adh_upload_candidates",
    "adh_score_uploaded",
    "adh_get_stream",
    "adh_synchronize",
    "adh_kernel_time_ms",
    "adh_fragcomp",
    "adh_fragcomp_stats",
    "adh_fragcomp_frames",
    "adh_select_candidates",
    "adh_select_time_ms",
    "adh_transpose_timstof",
    "adh_fdr_q_values",
    "adh_fdr_keep_best",
    "adh_mlp_create",
    "adh_mlp_destroy",
    "adh_mlp_param_count",
    "adh_mlp_set_state",
    "adh_mlp_get_state",
    "adh_mlp_stage_rows",
    "adh_mlp_fit",
    "adh_mlp_predict",
    "adh_mlp_time_ms",
    "adh_get_device_tables",
    "adh_zero_device_tables",
    "adh_debug_get_dense",
    "adh_mlp_stage_rows_device",
    "adh_mlp_staged_rows",
    "adh_mlp_predict_resident",
    "adh_fdr_resident",
    "adh_transfer_counters",
    "adh_host_alloc",
    "adh_host_free",
    "adh_copy_to_host",
    "adh_comm_unique_id",
    "adh_comm_init",
    "adh_comm_destroy",
    "adh_comm_wait",
    "adh_comm_gathered",
    "adh_comm_all_reduce_max",
    "adh_comm_barrier",
    "adh_comm_all_gather_host",
    "adh_comm_info",
    "adh_table_layout",
    "adh_device_synchronize",
]


def _process_start_time(pid: str | int = "self") -> float:
    """Wall-clock start of process ``pid`` (this process by default); falls back to now.  Never in the future:
    where /proc/uptime is virtualised (lxcfs) but the start ticks of /proc/<pid>/stat count from the host's
    boot, the difference is meaningless and the result is clamped to the current time."""
    import time

    now = time.time()
    try:
        with open(f"/proc/{pid}/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            uptime = float(f.read().split()[0])
        return min(now - (uptime - ticks / os.sysconf("SC_CLK_TCK")), now)
    except (OSError, ValueError, IndexError):
        return now


_PROCESS_T0 = _process_start_time()


def _attempt_lower_bound() -> float:
    """Earliest wall-clock time at which rank 0 of THIS attempt can have written the id file.

    Under a launcher (torchrun, mpirun, a script) every rank is a child of the launcher process, whose pid and
    start time are part of the nonce: nothing written before the launcher started can belong to this launch,
    and a rank that starts late (staggered ranks, a slow import) still accepts what rank 0 wrote long before.
    Ranks started by hand with a shared ``ADH_RUN_NONCE`` have a parent (the shell) that outlives the run; there
    the bound is this rank's own start minus ``ADH_RENDEZVOUS_STAGGER_S`` (default 600 s) - pick a new nonce
    per run, a file left by a crashed run of the same nonce inside that window would be accepted."""
    if os.environ.get("ADH_RUN_NONCE"):
        return _PROCESS_T0 - float(os.environ.get("ADH_RENDEZVOUS_STAGGER_S", "600"))
    return min(_process_start_time(os.getppid()), _PROCESS_T0)


class HipBackendError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    lib.adh_last_error.restype = C.c_char_p
    for name in EXPORTED_SYMBOLS[1:]:
        getattr(lib, name).restype = C.c_int
    return lib


lib = _load()


def host_threads(n_rows: int) -> tuple[int, int]:
    """(threads a scoring call over ``n_rows`` candidates starts on the host, CPU budget of this process): the
    library's ``adh_host_threads`` - cgroup quota, affinity mask, hardware threads; the rank's share of them."""
    t, b = C.c_int32(0), C.c_int32(0)
    fn = lib.adh_host_threads
    fn.argtypes = [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    _check(fn(int(n_rows), C.byref(t), C.byref(b)), "adh_host_threads")
    return int(t.value), int(b.value)


_pylib = None       # the same library through ctypes.PyDLL: calls keep the GIL (adh_host_take_objects needs that)
_OBJ_TAKE = None    # None: not probed yet; True / False: the threaded object gather is usable


def _probe_take_objects() -> bool:
    """Is ``ob_refcnt`` the first word of an object here (CPython <= 3.11, no trace-refs build)?  Checked on a probe
    object: five references more after a gather of five, and the gathered entries are the object."""
    global _pylib
    import sys

    if sys.implementation.name != "cpython" or sys.version_info[:2] > (3, 11) or os.environ.get("ADH_NO_OBJECT_TAKE"):
        return False
    #  the gather writes ``ob_refcnt`` as the first word of an object: debug and trace-refs builds put two
    # list pointers in front of it, and free-threaded builds keep the count elsewhere - refuse them by their build
    # flags, not only by the probe below
    import sysconfig

    if (sysconfig.get_config_var("Py_DEBUG") or sysconfig.get_config_var("Py_TRACE_REFS")
            or sysconfig.get_config_var("Py_GIL_DISABLED") or hasattr(sys, "gettotalrefcount")):
        return False
    try:
        _pylib = C.PyDLL(LIB_PATH)
        fn = _pylib.adh_host_take_objects
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32]
        probe = object()
        src = np.empty(1, dtype=object)
        src[0] = probe
        idx = np.zeros(5, dtype=np.int64)
        dst = np.empty(5, dtype=object)
        before = sys.getrefcount(probe)
        rc = fn(dst.ctypes.data, src.ctypes.data, idx.ctypes.data, 5, 1, id(None), 1)
        after = sys.getrefcount(probe)
        return rc == 0 and after - before == 5 and all(o is probe for o in dst)
    except Exception:
        return False


def take_objects(src: np.ndarray, idx: np.ndarray, threads: int = 8) -> np.ndarray:
    """``src[idx]`` for a 1-d object array: on several threads through ``adh_host_take_objects`` where that is safe
    (see there), NumPy's own gather otherwise (small inputs, other interpreters, negative indices).

    The call keeps the GIL for its whole duration (``ctypes.PyDLL``: the C threads touch reference counts): Python-level
    code of other threads - the fragments frame's worker, the futures of the host pool when they finish a NumPy call -
    waits meanwhile; only work already inside a GIL-free NumPy loop runs beside it.  ``ADH_NO_OBJECT_TAKE=1`` switches
    it off."""
    global _OBJ_TAKE
    src = np.asarray(src)
    if src.dtype != object or src.ndim != 1 or len(idx) < 262144 or not src.flags.c_contiguous:
        return src[idx]
    if _OBJ_TAKE is None:
        _OBJ_TAKE = _probe_take_objects()
    if not _OBJ_TAKE:
        return src[idx]
    idx64 = np.ascontiguousarray(idx, dtype=np.int64)
    dst = np.empty(len(idx64), dtype=object)
    rc = _pylib.adh_host_take_objects(dst.ctypes.data, src.ctypes.data, idx64.ctypes.data, len(idx64), len(src), id(None),
                                      int(threads))
    if rc != 0:  # (an index NumPy would wrap or reject: let NumPy decide; entries written so far are dropped with dst)
        return src[idx]
    return dst


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib.adh_last_error()
        raise HipBackendError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class _TableField(C.Structure):
    _fields_ = [("name", C.c_char * 40), ("offset", C.c_uint64), ("row_elems", C.c_uint32), ("elem_bytes", C.c_uint32),
                ("wire", C.c_int32), ("reserved", C.c_int32)]


def table_layout(rows: int, top_k: int) -> tuple[list[dict], int, int]:
    """``adh_table_layout``: the packed device buffer of ``rows`` candidates as the library lays it out -
    (fields in buffer order, total bytes, bytes of the wire prefix).  Needs no GPU."""
    n = C.c_int32(0)
    total, wire = C.c_uint64(0), C.c_uint64(0)
    _check(lib.adh_table_layout(C.c_int64(int(rows)), C.c_int32(int(top_k)), C.c_int32(0), None, C.byref(n),
                                C.byref(total), C.byref(wire)), "adh_table_layout")
    arr = (_TableField * n.value)()
    _check(lib.adh_table_layout(C.c_int64(int(rows)), C.c_int32(int(top_k)), C.c_int32(n.value), arr, C.byref(n),
                                C.byref(total), C.byref(wire)), "adh_table_layout")
    fields = [dict(name=f.name.decode(), offset=int(f.offset), row_elems=int(f.row_elems), elem_bytes=int(f.elem_bytes),
                   wire=bool(f.wire)) for f in arr]
    return fields, int(total.value), int(wire.value)


def device_count() -> int:
    n = C.c_int(0)
    _check(lib.adh_device_count(C.byref(n)), "adh_device_count")
    return n.value


class PinnedPool:
    """Page-locked host buffers (``adh_host_alloc``) handed out as numpy arrays and reused by tag.

    H2D / D2H copies from page-locked memory run asynchronously at the full PCIe rate; allocating
    it is slow (it pins pages), hence the pool: a buffer is grown, never shrunk, and the array
    returned for a tag is only valid until the same tag is requested again."""

    def __init__(self):
        self._bufs: dict[str, tuple[int, int]] = {}

    def empty(self, tag: str, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr, cap = self._bufs.get(tag, (0, 0))
        if cap < nbytes or ptr == 0:
            if ptr:
                lib.adh_host_free(C.c_void_p(ptr))
                self._bufs.pop(tag, None)
            want = max(nbytes + nbytes // 8, 64)
            p = C.c_void_p()
            _check(lib.adh_host_alloc(C.byref(p), C.c_uint64(want)), "adh_host_alloc")
            ptr, cap = int(p.value), want
            self._bufs[tag] = (ptr, cap)
        if nbytes == 0:
            return np.empty(shape, dtype=dtype)
        raw = (C.c_uint8 * nbytes).from_address(ptr)
        return np.frombuffer(raw, dtype=dtype).reshape(shape)

    def take(self, tag: str, src: np.ndarray, order: np.ndarray | None = None, dtype=None) -> np.ndarray:
        """``src[order]`` (or a copy of ``src``) written straight into the pooled buffer of ``tag``."""
        src = np.asarray(src)
        dtype = np.dtype(dtype or src.dtype)
        if order is None:
            out = self.empty(tag, src.shape, dtype)
            out[...] = src
            return out
        out = self.empty(tag, (len(order),) + src.shape[1:], dtype)
        if src.dtype == dtype and out.size:
            np.take(src, order, axis=0, out=out)
        elif out.size:
            out[...] = src[order]
        return out

    def close(self):
        for ptr, _ in self._bufs.values():
            lib.adh_host_free(C.c_void_p(ptr))
        self._bufs.clear()


def _launch_nonce() -> bytes:
    """What all ranks of ONE launch share and a later launch does not: the launcher process (parent of every
    rank) and its start time, plus whatever the launcher exports.  ``ADH_RUN_NONCE`` overrides (ranks started
    by hand from one shell share a parent that outlives the run)."""
    import hashlib

    ppid = os.getppid()
    start = ""
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time of the process
    except (OSError, IndexError):
        pass
    # (the restart count: an elastic restart under the same launcher is a new attempt with a new id file)
    tag = "|".join([os.environ.get("ADH_RUN_NONCE", ""), os.environ.get("MASTER_PORT", "0"),
                    os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                    str(ppid), start])
    return hashlib.sha256(tag.encode()).digest()


def _proc_start_ticks(pid: int) -> int:
    """Start time of process ``pid`` in clock ticks since boot (field 22 of /proc/<pid>/stat); -1 when it is gone."""
    try:
        with open(f"/proc/{pid}/stat") as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except (OSError, IndexError, ValueError):
        return -1


def _writer_tag(pid: int) -> bytes:
    """16 bytes that name the process that wrote an id file: pid and its start time (a recycled pid differs in the latter)."""
    import struct

    return struct.pack("<qq", int(pid), _proc_start_ticks(pid))


def _writer_alive(tag: bytes) -> bool:
    import struct

    pid, start = struct.unpack("<qq", tag)
    return pid > 0 and start >= 0 and _proc_start_ticks(pid) == start


_EXIT_REMOVALS: list = []


def _remove_at_exit(path: str) -> None:
    import atexit

    if not _EXIT_REMOVALS:
        def _cleanup():
            for p, tag in _EXIT_REMOVALS:
                try:  # (only a file this process wrote: a later attempt may have replaced it)
                    with open(p, "rb") as f:
                        data = f.read()
                    if data[32:48] == tag:
                        os.remove(p)
                except OSError:
                    pass

        atexit.register(_cleanup)
    _EXIT_REMOVALS.append((path, _writer_tag(os.getpid())))


def rendezvous_path() -> str:
    if os.environ.get("ADH_RENDEZVOUS_FILE"):
        return os.environ["ADH_RENDEZVOUS_FILE"]
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    folder = os.path.join(base, f"adh_{os.getuid()}")  # a directory of this user only: nobody else can pre-create
    os.makedirs(folder, mode=0o700, exist_ok=True)      # or redirect the id file
    import stat as _stat

    st = os.lstat(folder)  # (not stat: a planted symlink to some other private directory of the user must not pass)
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise HipBackendError(f"{folder} is not a private directory of this user")
    return os.path.join(folder, f"rccl_id_{_launch_nonce().hex()[:24]}")


def rendezvous_unique_id(rank: int, world: int, timeout: float = 300.0, make_id=None) -> bytes:
    """Hand rank 0's RCCL unique id to the other ranks of THIS NODE through a file in /dev/shm.

    The file is named after a per-launch nonce (launcher pid + start time, MASTER_PORT, run id) and starts
    with that nonce; rank 0 removes whatever a crashed earlier launch left under the name, creates the file
    exclusively (0600, inside a 0700 directory of this user) and removes it again once the communicator exists
    (``Context.comm_init``); readers only accept a file that carries their own nonce.  Single node: the ranks
    must share a file system and a launcher (checked through LOCAL_WORLD_SIZE)."""
    import time

    local_world = os.environ.get("LOCAL_WORLD_SIZE")
    if local_world is not None and int(local_world) != int(world):
        raise HipBackendError("the RCCL id rendezvous goes through a node-local file: all ranks must run on one node "
                              f"(WORLD_SIZE {world}, LOCAL_WORLD_SIZE {local_world})")
    path = rendezvous_path()
    nonce = _launch_nonce()
    if rank == 0:
        if make_id is None:
            buf = C.create_string_buffer(128)
            _check(lib.adh_comm_unique_id(buf), "adh_comm_unique_id")
            uid = buf.raw
        else:
            uid = bytes(make_id())
        try:
            os.unlink(path)  # a stale file of a crashed launch with the same nonce inputs
        except FileNotFoundError:
            pass
        tmp = f"{path}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(nonce + _writer_tag(os.getpid()) + uid)
        os.replace(tmp, path)  # atomic: readers see all of it or nothing
        _remove_at_exit(path)  # (a rank 0 that fails before Context.comm_init removes the file still cleans up)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                data = f.read()
                written = os.fstat(f.fileno()).st_mtime
            # a file older than the launch is what an earlier attempt with the same nonce inputs left behind
            # (rank 0 replaces it); the bound is the LAUNCHER's start, not this rank's: ranks may start seconds
            # or minutes apart (_attempt_lower_bound)
            fresh = written >= _attempt_lower_bound() - 2.0
            if fresh and len(data) == len(nonce) + 16 + 128 and data[: len(nonce)] == nonce:
                # ... and a file whose writer is gone is what a crashed attempt of THIS launch left (same launcher,
                # same port, seconds ago): rank 0 is alive while its peers read - it waits for them inside
                # ncclCommInitRank - so a dead writer means "not this attempt's"; the new rank 0 replaces the file
                if _writer_alive(data[len(nonce): len(nonce) + 16]):
                    return data[len(nonce) + 16:]
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise HipBackendError(f"rank {rank}: no RCCL unique id at {path} after {timeout:.0f} s")
        time.sleep(0.01)


class Context:
    """One GPU: staged run + library + candidate table (an ``adh_handle_t``)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib.adh_create(C.byref(self._h), C.c_int(device)), "adh_create")
        self.device = device
        self._run_key = None
        self._lib_key = None
        self.n_candidates = 0
        self.pinned = PinnedPool()
        self.comm_world = 1
        self.comm_rank = 0

    def close(self):
        if self._h:
            lib.adh_destroy(self._h)
            self._h = C.c_void_p()
            self.pinned.close()

    # -- multi-GPU (one process per GPU; RCCL behind the C ABI) -------------
    def comm_init(self, rank: int, world: int, max_rows_per_rank: int, unique_id: bytes | None = None,
                  timeout: float = 600.0) -> None:
        """Attach an RCCL communicator: every later ``score_host`` also all-gathers the computed
        tables of all ranks into HBM (``gathered_tables``).  Creating the communicator is collective; a rank
        whose peers never arrive (e.g. a stale unique id) fails after ``timeout`` seconds instead of hanging."""
        import threading

        uid = unique_id if unique_id is not None else rendezvous_unique_id(rank, world)
        buf = C.create_string_buffer(bytes(uid), 128)
        result: list = []

        def call():
            result.append(lib.adh_comm_init(self._h, C.c_int(rank), C.c_int(world), buf, C.c_int64(int(max_rows_per_rank))))
            msg = lib.adh_last_error()  # (thread-local: fetched on the thread that made the call)
            result.append(msg.decode() if msg else "")

        t = threading.Thread(target=call, name="adh_comm_init", daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            raise HipBackendError(f"adh_comm_init: rank {rank} of {world} still waits for its peers after {timeout:.0f} s "
                                  "(ncclCommInitRank is collective: did every rank start, with the same unique id?)")
        if result[0] != 0:
            raise HipBackendError(f"adh_comm_init failed ({result[0]}): {result[1]}")
        self.comm_rank, self.comm_world = int(rank), int(world)
        self._comm_attached = True
        if unique_id is None and rank == 0:
            # communicator creation is collective: every rank has read the id by now
            try:
                os.remove(rendezvous_path())
            except OSError:
                pass

    def comm_destroy(self) -> None:
        _check(lib.adh_comm_destroy(self._h), "adh_comm_destroy")
        self.comm_rank, self.comm_world = 0, 1
        self._comm_attached = False

    def comm_wait(self) -> None:
        _check(lib.adh_comm_wait(self._h), "adh_comm_wait")

    def barrier(self) -> None:
        _check(lib.adh_comm_barrier(self._h), "adh_comm_barrier")

    def all_gather_rows(self, local: np.ndarray, rows_per_rank: list[int]) -> np.ndarray:
        """Concatenation, in rank order, of every rank's ``local`` array (``rows_per_rank[r]`` rows on rank r,
        equal trailing shape and dtype): ``adh_comm_all_gather_host`` with the shards padded to the longest
        one.  With one rank (no communicator) the array comes back as it is."""
        rank, world = self.comm_info()
        if len(rows_per_rank) != world:
            raise ValueError(f"rows_per_rank has {len(rows_per_rank)} entries for {world} ranks")
        local = np.ascontiguousarray(local)
        if local.shape[0] != rows_per_rank[rank]:
            raise ValueError(f"rank {rank} holds {local.shape[0]} rows, rows_per_rank says {rows_per_rank[rank]}")
        if world == 1 and not getattr(self, "_comm_attached", False):
            return local
        cap = max(int(r) for r in rows_per_rank)
        row_bytes = int(np.prod(local.shape[1:], dtype=np.int64)) * local.dtype.itemsize
        send = np.zeros((cap,) + local.shape[1:], dtype=local.dtype)
        send[: local.shape[0]] = local
        recv = np.empty((world, cap) + local.shape[1:], dtype=local.dtype)
        _check(lib.adh_comm_all_gather_host(self._h, send.ctypes.data_as(C.c_void_p), C.c_uint64(cap * row_bytes),
                                            recv.ctypes.data_as(C.c_void_p)), "adh_comm_all_gather_host")
        return np.concatenate([recv[r, : int(rows_per_rank[r])] for r in range(world)], axis=0)

    def all_reduce_max(self, value: float) -> float:
        v = C.c_double(float(value))
        _check(lib.adh_comm_all_reduce_max(self._h, C.byref(v)), "adh_comm_all_reduce_max")
        return float(v.value)

    def comm_info(self) -> tuple[int, int]:
        """(rank, world) as RCCL reports them for the attached communicator; (0, 1) without one."""
        r, w = C.c_int(0), C.c_int(1)
        _check(lib.adh_comm_info(self._h, C.byref(r), C.byref(w)), "adh_comm_info")
        return int(r.value), int(w.value)

    def device_synchronize(self) -> None:
        _check(lib.adh_device_synchronize(self._h), "adh_device_synchronize")

    def _view_to_host(self, view: _abi.Output, rows: int, names=None) -> dict:
        out = {}
        for name, (shape, dt) in _abi.output_shapes(rows, int(view.top_k), extras=True).items():
            if names is not None and name not in names:
                continue
            ptr = C.cast(getattr(view, name), C.c_void_p).value
            if not ptr:
                continue
            a = np.empty(shape, dtype=dt)
            _check(lib.adh_copy_to_host(self._h, a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_uint64(a.nbytes)),
                   "adh_copy_to_host")
            out[name] = a
        return out

    def device_tables(self) -> _abi.Output:
        """Device view of the tables of the last ``score_host`` call (they stay in HBM)."""
        view = _abi.Output()
        _check(lib.adh_get_device_tables(self._h, C.byref(view)), "adh_get_device_tables")
        return view

    def debug_get_dense(self, frame_start, frame_stop, mz_query, mass_tolerance, quad_lo, quad_hi,
                        scan_start: int = 0, scan_stop: int = 1):
        """The gather kernel's dense tile for one query (test entry): ``(dense[2, K, O, S, F], obs)``
        as ``get_dense(..., absolute_masses=True)`` returns them (S = 1 for AlphaRaw runs)."""
        q = _abi.as_c(mz_query, np.float32)
        cap = 2 * q.shape[0] * 8 * max(int(scan_stop) - int(scan_start), 1) * (int(frame_stop) - int(frame_start) + 1)
        buf = np.zeros(max(cap, 1), dtype=np.float32)
        obs = np.zeros(8, dtype=np.int32)
        n_obs, n_s, n_f = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _check(
            lib.adh_debug_get_dense(
                self._h, C.c_int64(int(frame_start)), C.c_int64(int(frame_stop)), C.c_int64(int(scan_start)),
                C.c_int64(int(scan_stop)), q.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(q.shape[0]),
                C.c_float(float(mass_tolerance)), C.c_float(float(quad_lo)), C.c_float(float(quad_hi)),
                buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(buf.shape[0]),
                obs.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_obs), C.byref(n_s), C.byref(n_f)),
            "adh_debug_get_dense",
        )
        K, O, S, F = q.shape[0], n_obs.value, n_s.value, n_f.value
        return buf[: 2 * K * O * S * F].reshape(2, K, O, S, F).copy(), obs[:O].astype(np.int64)

    def d2h_bytes(self, reset: bool = False) -> int:
        """Bytes the library copied device -> host on this GPU since the last reset."""
        v = C.c_uint64(0)
        _check(lib.adh_transfer_counters(self._h, C.byref(v), C.c_int(int(reset))), "adh_transfer_counters")
        return int(v.value)

    def fdr_resident(self, mlp: "DeviceMlp", group_a, group_b=None, tiebreak=None, cycle=None,
                     rt_tol_seconds: float = 3.0, mass_tol_ppm: float = 15.0, fdr_heuristic: float = 0.1):
        """fdr.py:134-178 on the device for rows staged with ``DeviceMlp.stage_rows_device`` and scored
        with ``predict_resident``: returns ``(candidate_row, proba, qval)`` of the surviving PSMs."""
        ga = _abi.as_c(group_a, np.int64)
        gb = _abi.as_c(group_b, np.int64) if group_b is not None else None
        tb = _abi.as_c(tiebreak, np.int64) if tiebreak is not None else None
        cyc = _abi.as_c(cycle, np.float64) if cycle is not None else None
        if cyc is not None and (cyc.ndim != 4 or cyc.shape[0] != 1 or cyc.shape[3] != 2):
            raise ValueError("cycle must have shape (1, cycle_len, cycle_scans, 2)")
        m = mlp.n_rows
        rows, proba, qval = np.zeros(m, np.int64), np.zeros(m, np.float32), np.zeros(m, np.float64)
        n_out = C.c_int64(0)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t)) if a is not None else None  # noqa: E731
        _check(
            lib.adh_fdr_resident(
                self._h, mlp._m, p(ga, C.c_int64), p(gb, C.c_int64), p(tb, C.c_int64), p(cyc, C.c_double),
                C.c_int32(cyc.shape[1] if cyc is not None else 0), C.c_int32(cyc.shape[2] if cyc is not None else 0),
                C.c_double(float(rt_tol_seconds)), C.c_double(float(mass_tol_ppm)), C.c_double(float(fdr_heuristic)),
                C.byref(n_out), p(rows, C.c_int64), p(proba, C.c_float), p(qval, C.c_double)),
            "adh_fdr_resident",
        )
        k = int(n_out.value)
        return rows[:k], proba[:k], qval[:k]

    def zero_device_tables(self, stream: int = 0) -> None:
        _check(lib.adh_zero_device_tables(self._h, C.c_void_p(stream)), "adh_zero_device_tables")

    def device_tables_to_host(self, names=None) -> dict:
        view = self.device_tables()
        return self._view_to_host(view, int(view.n), names)

    def gathered_tables(self, rank: int, rows: int | None = None) -> dict:
        """Host copy of the computed ("wire") tables rank ``rank`` contributed to the last all-gather."""
        view = _abi.Output()
        n = C.c_int64(0)
        _check(lib.adh_comm_gathered(self._h, C.c_int(rank), C.byref(view), C.byref(n)), "adh_comm_gathered")
        return self._view_to_host(view, int(n.value) if rows is None else int(rows))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- staging ---------------------------------------------------------
    @staticmethod
    def _key(*arrays):
        """Identity of host arrays that are staged in HBM: address, shape, dtype and a checksum of a
        strided sample (<= 64 K elements per array).  The arrays themselves are kept alive while they
        are staged (``_run_keepalive`` / ``_lib_keepalive``), so an address cannot be handed to a new
        array; the checksum catches in-place edits."""
        import zlib

        key = []
        for a in map(np.asarray, arrays):
            flat = a.reshape(-1)
            step = max(1, flat.shape[0] // 65536)
            crc = zlib.crc32(np.ascontiguousarray(flat[::step]).view(np.uint8)) if flat.shape[0] else 0
            key.append((a.__array_interface__["data"][0], a.shape, str(a.dtype), crc))
        return tuple(key)

    def stage_run(self, dia, force: bool = False) -> bool:
        """Copy the run to HBM unless this very run is already staged."""
        if hasattr(dia, "tof_indptr"):  # TimsTOFTransposeJIT layout (bruker_jit.py:22-137)
            key = self._key(dia.mz_values, dia.intensity_values, dia.push_indices, dia.tof_indptr)
            if not force and key == self._run_key:
                return False
            m = _abi.pack_timstof(dia)
            _check(lib.adh_stage_timstof(self._h, m.ref()), "adh_stage_timstof")
            self._run_key = key
            self._run_keepalive = (dia.mz_values, dia.intensity_values, dia.push_indices, dia.tof_indptr)
            return True
        key = self._key(dia.mz_values, dia.intensity_values, dia.peak_start_idx_list, dia.rt_values)
        if not force and key == self._run_key:
            return False
        m = _abi.pack_alpharaw(dia)
        _check(lib.adh_stage_alpharaw(self._h, m.ref()), "adh_stage_alpharaw")
        self._run_key = key
        self._run_keepalive = (dia.mz_values, dia.intensity_values, dia.peak_start_idx_list, dia.rt_values)
        return True

    def stage_fragments(self, *columns, force: bool = False) -> bool:
        key = self._key(*columns)
        if not force and key == self._lib_key:
            return False
        m = _abi.pack_fragments(*columns)
        _check(lib.adh_stage_fragments(self._h, m.ref()), "adh_stage_fragments")
        self._lib_key = key
        self._lib_keepalive = columns
        return True

    # -- scoring ---------------------------------------------------------
    def score_host(self, cands: _abi.Marshalled, cfg_jit, with_stats: bool = False,
                   reuse_buffers: bool = False) -> dict:
        """Host table in, host OutputPsmDF arrays out (chunked, pipelined H2D / kernels / D2H).

        ``reuse_buffers=True`` writes into this context's page-locked output buffers (full PCIe
        rate, no allocation): the returned arrays are then only valid until the next such call."""
        n = int(cands.struct.n)
        # the call copies every table back completely: no need to clear the host buffers first
        alloc = (lambda name, shape, dt: self.pinned.empty("out:" + name, shape, dt)) if reuse_buffers else None
        # production calls (reuse_buffers) fetch exactly the OutputPsmDF tables; the diagnostic columns
        # (matched-peak counts, library slots) stay in HBM unless asked for
        width = _abi.output_width(cands, int(cfg_jit.top_k_fragments))
        if self.comm_world > 1:
            # the all-gather needs ONE table layout on all ranks: the widest shard decides (a collective)
            width = int(round(self.all_reduce_max(float(width))))
        m_out, arrays = _abi.alloc_output(n, width,
                                          with_stats=with_stats, zero=False, alloc=alloc,
                                          with_slots=with_stats or not reuse_buffers)
        cfg = _abi.pack_config(cfg_jit)
        _check(
            lib.adh_score_candidates(self._h, cands.ref(), C.byref(cfg), m_out.ref()),
            "adh_score_candidates",
        )
        self.n_candidates = n
        return arrays

    def score_host_compact(self, cands: _abi.Marshalled, cfg_jit, slots_per_row: float | None = None,
                           buffers: dict | None = None) -> dict:
        """``adh_score_candidates_compact``: the valid candidates and the filled fragment slots, column by column, in
        arrays this call allocates (the caller owns them): ``row``, ``precursor_idx``, ``rank``, ``features`` as
        [46, n_rows] (every feature a contiguous row) and the ``fragment_*`` columns per filled slot with
        ``fragment_row`` = row of the slot's candidate in the candidate table.

        The arrays are allocated at a capacity - all candidates, ``slots_per_row`` filled slots per candidate (default:
        the width of the padded tables, which always suffices) - and returned as views of their first n entries: the
        pages behind the unused tail of a large fresh allocation are never touched, so they cost address space only.
        A call whose ``slots_per_row`` turns out too small is repeated with what it needs.

        ``buffers``: a dict a caller keeps between calls of the same shape; the capacity arrays of the first call are
        parked in it and written again by the next ones (a steady-state loop - a benchmark, a search that scores batch
        after batch - then touches no fresh pages; the returned views are overwritten by the next call)."""
        n = int(cands.struct.n)
        width = _abi.output_width(cands, int(cfg_jit.top_k_fragments))
        cfg = _abi.pack_config(cfg_jit)
        rows_cap = max(n, 1)
        slots_cap = max(int(n * (width if slots_per_row is None else min(float(slots_per_row), width))) + 1, 1)
        while True:
            out = _abi.CompactOutput()
            out.rows_capacity, out.slots_capacity, out.top_k = rows_cap, slots_cap, width
            key = (rows_cap, slots_cap)
            arrays = buffers.get(key) if buffers is not None else None
            if arrays is None:
                arrays = {}
                for name, dt in _abi.COMPACT_ROW_FIELDS:
                    arrays[name] = np.empty(rows_cap, dtype=dt)
                arrays["features"] = np.empty((_abi.NUM_FEATURES, rows_cap), dtype=np.float32)
                for name, dt in _abi.COMPACT_SLOT_FIELDS:
                    arrays[name] = np.empty(slots_cap, dtype=dt)
                if buffers is not None:
                    buffers.clear()
                    buffers[key] = arrays
            for name, a in arrays.items():
                setattr(out, name, a.ctypes.data_as(dict(out._fields_)[name]))
            rc = lib.adh_score_candidates_compact(self._h, cands.ref(), C.byref(cfg), C.byref(out))
            if rc != 0 and (int(out.n_rows) > rows_cap or int(out.n_slots) > slots_cap):
                rows_cap, slots_cap = max(rows_cap, int(out.n_rows)), max(slots_cap, int(out.n_slots))
                continue  # (only a caller's own slots_per_row can be too small)
            _check(rc, "adh_score_candidates_compact")
            break
        self.n_candidates = n
        nr, ns = int(out.n_rows), int(out.n_slots)
        res = {name: arrays[name][:nr] for name, _ in _abi.COMPACT_ROW_FIELDS}
        res["features"] = arrays["features"][:, :nr]
        res.update({name: arrays[name][:ns] for name, _ in _abi.COMPACT_SLOT_FIELDS})
        res["top_k"] = width
        return res

    def upload_candidates(self, cands: _abi.Marshalled) -> None:
        _check(lib.adh_upload_candidates(self._h, cands.ref()), "adh_upload_candidates")
        self.n_candidates = int(cands.struct.n)

    def stream_handle(self) -> int:
        """The context's own hipStream_t as an integer."""
        p = C.c_void_p()
        _check(lib.adh_get_stream(self._h, C.byref(p)), "adh_get_stream")
        return p.value or 0

    def score_uploaded(self, cfg_jit, out_struct: _abi.Output, stream: int = 0) -> None:
        """Enqueue scoring of the uploaded table into device buffers on ``stream`` (a raw
        hipStream_t; 0 = HIP's default stream); does not synchronise.  The caller must have
        zeroed the buffers on the same stream."""
        cfg = _abi.pack_config(cfg_jit)
        _check(
            lib.adh_score_uploaded(self._h, C.byref(cfg), C.byref(out_struct), C.c_void_p(stream)),
            "adh_score_uploaded",
        )

    def synchronize(self) -> None:
        _check(lib.adh_synchronize(self._h), "adh_synchronize")

    def kernel_time_ms(self, reset: bool = True):
        """(gather_ms, feature_ms, launches): HIP-event averages per scoring call."""
        g = C.c_double(0)
        f = C.c_double(0)
        n = C.c_int64(0)
        _check(
            lib.adh_kernel_time_ms(self._h, C.byref(g), C.byref(f), C.byref(n), C.c_int(int(reset))),
            "adh_kernel_time_ms",
        )
        return g.value, f.value, n.value

    # -- fragment competition -------------------------------------------
    def select_candidates(self, precursors: _abi.Marshalled, cfg, kernel) -> dict:
        """Candidate selection on the staged AlphaRaw run and library: packed precursor table in,
        the CandidateContainer arrays out (rows without a candidate keep score 0)."""
        c = _abi.pack_selection_config(cfg)
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        if k.ndim != 2:
            raise ValueError("kernel must be 2-D (scan, cycle)")
        n = int(precursors.struct.n) * int(c.candidate_count)
        m_out, arrays = _abi.alloc_candidate_table(n)
        _check(
            lib.adh_select_candidates(
                self._h, precursors.ref(), C.byref(c), k.ctypes.data_as(C.POINTER(C.c_float)),
                C.c_int32(k.shape[0]), C.c_int32(k.shape[1]), m_out.ref(),
            ),
            "adh_select_candidates",
        )
        return arrays

    def select_time_ms(self) -> float:
        ms = C.c_double(0.0)
        _check(lib.adh_select_time_ms(self._h, C.byref(ms)), "adh_select_time_ms")
        return float(ms.value)

    def transpose_timstof(self, tof_indices, push_indptr, n_tof_indices: int, values):
        """Drop-in for ``_transpose`` (alphadia/raw_data/bruker.py:201-280): returns
        ``(push_indices, tof_indptr, new_values)`` in the TOF-major layout."""
        tof = _abi.as_c(tof_indices, np.uint32)
        ptr = _abi.as_c(push_indptr, np.int64)
        val = np.ascontiguousarray(values, dtype=np.uint16)
        n = tof.shape[0]
        if val.shape[0] != n or ptr.shape[0] < 1:
            raise ValueError("tof_indices / values / push_indptr have inconsistent lengths")
        push_out = np.zeros(n, dtype=np.uint32)
        val_out = np.zeros(n, dtype=np.uint16)
        indptr_out = np.zeros(int(n_tof_indices) + 1, dtype=np.int64)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        _check(
            lib.adh_transpose_timstof(
                self._h, p(tof, C.c_uint32), p(ptr, C.c_int64), C.c_int64(ptr.shape[0] - 1),
                C.c_int64(int(n_tof_indices)), p(val, C.c_uint16), C.c_int64(n),
                p(push_out, C.c_uint32), p(indptr_out, C.c_int64), p(val_out, C.c_uint16),
            ),
            "adh_transpose_timstof",
        )
        return push_out, indptr_out, val_out

    def fragcomp(self, window_start, window_stop, rt, frag_start, frag_stop, fragment_mz,
                 rt_tol_seconds: float, mass_tol_ppm: float, valid=None) -> np.ndarray:
        ws = _abi.as_c(window_start, np.int64)
        we = _abi.as_c(window_stop, np.int64)
        rtv = _abi.as_c(rt, np.float32)
        fs = _abi.as_c(frag_start, np.int64)
        fe = _abi.as_c(frag_stop, np.int64)
        fm = _abi.as_c(fragment_mz, np.float32)
        # the reference starts from an all-true column (fragcomp.py:281); `valid` lets a caller start from less
        valid = (np.ones(rtv.shape[0], dtype=np.uint8) if valid is None
                 else np.array(valid, dtype=bool).astype(np.uint8))
        if valid.shape[0] != rtv.shape[0] or fs.shape[0] != rtv.shape[0] or fe.shape[0] != rtv.shape[0]:
            raise ValueError("rt / fragment ranges / valid must have one entry per PSM")
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        _check(
            lib.adh_fragcomp(
                self._h,
                C.c_int64(ws.shape[0]),
                p(ws, C.c_int64),
                p(we, C.c_int64),
                C.c_int64(rtv.shape[0]),
                p(rtv, C.c_float),
                p(fs, C.c_int64),
                p(fe, C.c_int64),
                C.c_int64(fm.shape[0]),
                p(fm, C.c_float),
                C.c_double(float(rt_tol_seconds)),
                C.c_double(float(mass_tol_ppm)),
                p(valid, C.c_uint8),
            ),
            "adh_fragcomp",
        )
        return valid.view(np.bool_)

    def fragcomp_frames(self, psm_precursor_idx, psm_rank, psm_mz_observed, psm_rt_observed, psm_proba,
                        frag_precursor_idx, frag_rank, frag_mz_observed, cycle, rt_tol_seconds: float, mass_tol_ppm: float):
        """``adh_fragcomp_frames``: plan and competition on the device from the frames' columns.  Returns
        ``(rows, valid)`` - input position and flag of every processed PSM in processing order - or ``None`` when the
        fragment table is not grouped by candidate (the caller prepares the plan itself)."""
        pp, pr = _abi.as_c(psm_precursor_idx, np.uint32), _abi.as_c(psm_rank, np.uint8)
        pmz, prt = _abi.as_c(psm_mz_observed, np.float32), _abi.as_c(psm_rt_observed, np.float32)
        ppb = _abi.as_c(psm_proba, np.float32)
        fp, fr = _abi.as_c(frag_precursor_idx, np.uint32), _abi.as_c(frag_rank, np.uint8)
        fmz = _abi.as_c(frag_mz_observed, np.float32)
        lower = np.ascontiguousarray(cycle[0, :, :, 0].min(axis=1), dtype=np.float64)
        upper = np.ascontiguousarray(cycle[0, :, :, 1].max(axis=1), dtype=np.float64)
        n = pp.shape[0]
        if not (pr.shape[0] == pmz.shape[0] == prt.shape[0] == ppb.shape[0] == n) or fr.shape[0] != fp.shape[0] \
                or fmz.shape[0] != fp.shape[0]:
            raise ValueError("columns of one frame differ in length")
        rows, valid = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.uint8)
        n_rows, grouped = C.c_int64(0), C.c_int32(1)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        _check(
            lib.adh_fragcomp_frames(
                self._h, C.c_int64(n), p(pp, C.c_uint32), p(pr, C.c_uint8), p(pmz, C.c_float), p(prt, C.c_float),
                p(ppb, C.c_float), C.c_int64(fp.shape[0]), p(fp, C.c_uint32), p(fr, C.c_uint8), p(fmz, C.c_float),
                C.c_int32(lower.shape[0]), p(lower, C.c_double), p(upper, C.c_double), C.c_double(float(rt_tol_seconds)),
                C.c_double(float(mass_tol_ppm)), p(rows, C.c_int64), p(valid, C.c_uint8), C.byref(n_rows), C.byref(grouped)),
            "adh_fragcomp_frames",
        )
        if not grouped.value:
            return None
        k = int(n_rows.value)
        return rows[:k], valid[:k].astype(bool)

    def fragcomp_stats(self) -> dict:
        """``adh_fragcomp_stats``: device time and work counters of the last competition."""
        ms = C.c_double(0.0)
        pairs, waiting = C.c_int64(0), C.c_int64(0)
        rounds, serial = C.c_int32(0), C.c_int32(0)
        _check(lib.adh_fragcomp_stats(self._h, C.byref(ms), C.byref(pairs), C.byref(waiting), C.byref(rounds),
                                      C.byref(serial)), "adh_fragcomp_stats")
        return dict(kernel_ms=ms.value, pairs=pairs.value, waiting=waiting.value, rounds=rounds.value,
                    serial=bool(serial.value))


    # -- FDR stage ---------------------------------------------------------
    def fdr_q_values(self, score, decoy, tiebreak=None):
        """``get_q_values`` (alphadia/fdr/fdr.py:232-297): returns ``(order, qval)``, the input row
        at every sorted position and its q-value."""
        sc = _abi.as_c(score, np.float64)
        de = _abi.as_c(np.asarray(decoy) != 0, np.uint8)
        n = sc.shape[0]
        if de.shape[0] != n:
            raise ValueError("score and decoy differ in length")
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        tb = None
        if tiebreak is not None:
            tb = _abi.as_c(tiebreak, np.int64)
            if tb.shape[0] != n:
                raise ValueError("score and tiebreak differ in length")
        order = np.zeros(n, dtype=np.int64)
        qval = np.zeros(n, dtype=np.float64)
        _check(
            lib.adh_fdr_q_values(self._h, C.c_int64(n), p(sc, C.c_double), p(de, C.c_uint8),
                                 p(tb, C.c_int64) if tb is not None else None, p(order, C.c_int64),
                                 p(qval, C.c_double)),
            "adh_fdr_q_values",
        )
        return order, qval

    def fdr_keep_best(self, score, group_a, group_b=None) -> np.ndarray:
        """``keep_best`` (alphadia/fdr/fdr.py:181-213) as a mask over the input rows."""
        sc = _abi.as_c(score, np.float64)
        ga = _abi.as_c(group_a, np.int64)
        gb = _abi.as_c(group_b, np.int64) if group_b is not None else None
        n = sc.shape[0]
        if ga.shape[0] != n or (gb is not None and gb.shape[0] != n):
            raise ValueError("score and group columns differ in length")
        keep = np.zeros(n, dtype=np.uint8)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        _check(
            lib.adh_fdr_keep_best(self._h, C.c_int64(n), p(sc, C.c_double), p(ga, C.c_int64),
                                  p(gb, C.c_int64) if gb is not None else None, p(keep, C.c_uint8)),
            "adh_fdr_keep_best",
        )
        return keep.view(np.bool_)


class DeviceMlp:
    """An ``adh_mlp_t``: the classifier network with its parameters, optimiser moments and the
    staged feature matrix in HBM."""

    def __init__(self, ctx: Context, input_dim: int, layers, output_dim: int = 2):
        self._ctx = ctx  # keeps the handle alive
        self.arch = _abi.pack_mlp_arch(input_dim, layers, output_dim)
        self.input_dim, self.layers, self.output_dim = int(input_dim), [int(x) for x in layers], int(output_dim)
        n = C.c_int64(0)
        _check(lib.adh_mlp_param_count(C.byref(self.arch), C.byref(n)), "adh_mlp_param_count")
        self.n_params = int(n.value)
        self._m = C.c_void_p()
        _check(lib.adh_mlp_create(ctx._h, C.byref(self.arch), C.byref(self._m)), "adh_mlp_create")
        self.n_rows = 0

    def close(self):
        if self._m:
            lib.adh_mlp_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _p(a, t):
        return a.ctypes.data_as(C.POINTER(t))

    def set_state(self, params, running_mean, running_var, num_batches_tracked: int = 0):
        pa = _abi.as_c(params, np.float32)
        rm = _abi.as_c(running_mean, np.float32)
        rv = _abi.as_c(running_var, np.float32)
        if pa.shape != (self.n_params,) or rm.shape != (self.input_dim,) or rv.shape != (self.input_dim,):
            raise ValueError("state arrays do not match the architecture")
        _check(lib.adh_mlp_set_state(self._m, self._p(pa, C.c_float), self._p(rm, C.c_float), self._p(rv, C.c_float),
                                     C.c_int64(int(num_batches_tracked))), "adh_mlp_set_state")

    def get_state(self):
        pa = np.zeros(self.n_params, dtype=np.float32)
        rm = np.zeros(self.input_dim, dtype=np.float32)
        rv = np.zeros(self.input_dim, dtype=np.float32)
        nbt = C.c_int64(0)
        _check(lib.adh_mlp_get_state(self._m, self._p(pa, C.c_float), self._p(rm, C.c_float), self._p(rv, C.c_float),
                                     C.byref(nbt)), "adh_mlp_get_state")
        return pa, rm, rv, int(nbt.value)

    def stage_rows(self, x, y=None):
        xa = _abi.as_c(x, np.float32)
        if xa.ndim != 2:
            raise ValueError("x must be (n_samples, n_features)")
        ya = None
        if y is not None:
            ya = _abi.as_c(y, np.float32)
            if ya.shape != (xa.shape[0],):
                raise ValueError("y must hold one target per row")
        _check(lib.adh_mlp_stage_rows(self._m, self._p(xa, C.c_float), C.c_int64(xa.shape[0]), C.c_int32(xa.shape[1]),
                                      self._p(ya, C.c_float) if ya is not None else None), "adh_mlp_stage_rows")
        self.n_rows = xa.shape[0]

    def stage_rows_device(self, src_cols, decoy, extra_cols=()):
        """Stage the usable rows of the device tables of the last ``score_host`` call (targets first,
        then decoys); returns ``(n_targets, n_decoys)``.  See ``adh_mlp_stage_rows_device``."""
        sc = _abi.as_c(src_cols, np.int32)
        de = _abi.as_c(np.asarray(decoy) != 0, np.uint8)
        extras = [_abi.as_c(e, np.float32) for e in extra_cols]
        if any(e.shape != de.shape for e in extras):
            raise ValueError("extra columns must have one value per candidate row")
        arr = (C.POINTER(C.c_float) * max(len(extras), 1))(*[self._p(e, C.c_float) for e in extras])
        nt, nd = C.c_int64(0), C.c_int64(0)
        _check(lib.adh_mlp_stage_rows_device(self._m, self._p(sc, C.c_int32), C.c_int32(sc.shape[0]), arr,
                                             C.c_int32(len(extras)), self._p(de, C.c_uint8), C.c_int64(de.shape[0]),
                                             C.byref(nt), C.byref(nd)), "adh_mlp_stage_rows_device")
        self.n_rows = int(nt.value) + int(nd.value)
        return int(nt.value), int(nd.value)

    def staged_rows(self) -> np.ndarray:
        rows = np.zeros(self.n_rows, dtype=np.int64)
        _check(lib.adh_mlp_staged_rows(self._m, self._p(rows, C.c_int64), C.c_int64(rows.shape[0])), "adh_mlp_staged_rows")
        return rows

    def predict_resident(self) -> None:
        _check(lib.adh_mlp_predict_resident(self._m), "adh_mlp_predict_resident")

    def fit(self, train_rows, batch_start, batch_size: int, learning_rate: float, weight_decay: float,
            dropout: float, seed: int = 0, first_step: int = 0, betas=(0.9, 0.999), eps: float = 1e-8) -> np.ndarray:
        tr = _abi.as_c(train_rows, np.int64)
        bs = _abi.as_c(batch_start, np.int64)
        f = _abi.MlpFit()
        f.train_rows, f.n_train = self._p(tr, C.c_int64), tr.shape[0]
        f.batch_start, f.n_steps = self._p(bs, C.c_int64), bs.shape[0]
        f.batch_size = int(batch_size)
        f.learning_rate, f.weight_decay, f.dropout = float(learning_rate), float(weight_decay), float(dropout)
        f.beta1, f.beta2, f.eps = float(betas[0]), float(betas[1]), float(eps)
        f.seed, f.first_step = int(seed) & (2**64 - 1), int(first_step)
        loss = np.zeros(bs.shape[0], dtype=np.float32)
        _check(lib.adh_mlp_fit(self._m, C.byref(f), self._p(loss, C.c_float)), "adh_mlp_fit")
        return loss

    def predict(self, rows=None) -> np.ndarray:
        if rows is None:
            n, rp = self.n_rows, None
        else:
            ra = _abi.as_c(rows, np.int64)
            n, rp = ra.shape[0], self._p(ra, C.c_int64)
        out = np.zeros((n, self.output_dim), dtype=np.float32)
        _check(lib.adh_mlp_predict(self._m, rp, C.c_int64(n), self._p(out, C.c_float)), "adh_mlp_predict")
        return out

    def time_ms(self):
        a, b = C.c_double(0.0), C.c_double(0.0)
        _check(lib.adh_mlp_time_ms(self._m, C.byref(a), C.byref(b)), "adh_mlp_time_ms")
        return float(a.value), float(b.value)


_contexts: dict[int, Context] = {}


def default_device() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def get_context(device: int | None = None) -> Context:
    """Process-wide context per GPU (so the staged run survives between batches)."""
    device = default_device() if device is None else int(device)
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = Context(device)
        _contexts[device] = ctx
    return ctx
