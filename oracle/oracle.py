"""ctypes front-end of the CPU parity oracle (oracle/adh_oracle.cpp).

TEST INFRASTRUCTURE: imported only by tests/, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``.  The product never imports this.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from alphadia_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libadh_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "adh_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.adh_oracle_score.restype = C.c_int
        _lib.adh_oracle_get_dense.restype = C.c_int
        _lib.adh_oracle_search_sorted_left.restype = C.c_int64
        _lib.adh_oracle_save_corrcoeff.restype = C.c_double
        _lib.adh_oracle_fragcomp.restype = C.c_int
        _lib.adh_oracle_select.restype = C.c_int
        _lib.adh_oracle_select_timstof.restype = C.c_int
        _lib.adh_oracle_find_peaks_1d.restype = C.c_int32
        _lib.adh_oracle_symetric_limits_1d.restype = None
    return _lib


def score(dia, fragment_cols, cand_marshalled, cfg_jit, n_threads: int = 1, with_stats=False,
          reuse=None):
    """Run the restated ``Candidate.process`` over a packed candidate table.

    ``reuse``: the ``(marshalled, arrays)`` pair of an earlier call with the same shape
    (its pages are already touched; used when timing the CPU baseline)."""
    m_dia = _abi.pack_alpharaw(dia)
    m_frag = _abi.pack_fragments(*fragment_cols)
    cfg = _abi.pack_config(cfg_jit)
    n = int(cand_marshalled.struct.n)
    if reuse is not None:
        m_out, arrays = reuse
        for a in arrays.values():
            a.fill(0)
    else:
        m_out, arrays = _abi.alloc_output(n, _abi.output_width(cand_marshalled, int(cfg_jit.top_k_fragments)),
                                          with_stats=with_stats)
    score.last_buffers = (m_out, arrays)
    rc = lib().adh_oracle_score(
        m_dia.ref(), m_frag.ref(), cand_marshalled.ref(), C.byref(cfg), m_out.ref(), C.c_int(n_threads)
    )
    if rc != 0:
        raise RuntimeError(f"adh_oracle_score failed: {rc}")
    return arrays


def score_timstof(dia, fragment_cols, cand_marshalled, cfg_jit, n_threads: int = 1, with_stats=False):
    """The same restatement over an ion-mobility run (TimsTOFTransposeJIT layout)."""
    m_dia = _abi.pack_timstof(dia)
    m_frag = _abi.pack_fragments(*fragment_cols)
    cfg = _abi.pack_config(cfg_jit)
    n = int(cand_marshalled.struct.n)
    m_out, arrays = _abi.alloc_output(n, _abi.output_width(cand_marshalled, int(cfg_jit.top_k_fragments)),
                                      with_stats=with_stats)
    lib().adh_oracle_score_timstof.restype = C.c_int
    rc = lib().adh_oracle_score_timstof(
        m_dia.ref(), m_frag.ref(), cand_marshalled.ref(), C.byref(cfg), m_out.ref(), C.c_int(n_threads)
    )
    if rc != 0:
        raise RuntimeError(f"adh_oracle_score_timstof failed: {rc}")
    return arrays


def get_dense_timstof(dia, frame_start, frame_stop, scan_start, scan_stop, mz_query, tol, quad_lo, quad_hi):
    m_dia = _abi.pack_timstof(dia)
    mzq = np.ascontiguousarray(mz_query, dtype=np.float32)
    K = mzq.shape[0]
    L = m_dia.struct.cycle_len
    z = m_dia.struct.zeroth_frame
    F = max((int(frame_stop) - z) // L - (int(frame_start) - z) // L, 0)
    S = int(scan_stop) - int(scan_start)
    cap = 2 * K * L * S * F
    dense = np.zeros(max(cap, 1), dtype=np.float32)
    pidx = np.zeros(L, dtype=np.int64)
    n_obs, n_sc, n_fr = C.c_int32(), C.c_int32(), C.c_int32()
    lib().adh_oracle_get_dense_timstof.restype = C.c_int
    rc = lib().adh_oracle_get_dense_timstof(
        m_dia.ref(), C.c_int64(int(frame_start)), C.c_int64(int(frame_stop)), C.c_int64(int(scan_start)),
        C.c_int64(int(scan_stop)), mzq.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(K),
        C.c_float(float(tol)), C.c_double(float(quad_lo)), C.c_double(float(quad_hi)),
        dense.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(cap),
        pidx.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n_obs), C.byref(n_sc), C.byref(n_fr),
    )
    if rc != 0:
        raise RuntimeError(f"adh_oracle_get_dense_timstof failed: {rc}")
    O, S2, F2 = n_obs.value, n_sc.value, n_fr.value
    return dense[: 2 * K * O * S2 * F2].reshape(2, K, O, S2, F2).copy(), pidx[:O].copy()


def set_numpy_typing(on: bool) -> None:
    """Switch the three shim-vs-Numba promotion sites (see adh_oracle.cpp)."""
    lib().adh_oracle_set_numpy_typing(C.c_int(int(bool(on))))


def get_dense(dia, frame_start, frame_stop, mz_query, tol, quad_lo, quad_hi, absolute):
    m_dia = _abi.pack_alpharaw(dia)
    mzq = np.ascontiguousarray(mz_query, dtype=np.float32)
    K = mzq.shape[0]
    L = m_dia.struct.cycle_len
    F = max(int(frame_stop) // L - int(frame_start) // L, 0)
    omax = L * m_dia.struct.cycle_scans
    cap = 2 * K * omax * 2 * F
    dense = np.zeros(max(cap, 1), dtype=np.float32)
    pidx = np.zeros(omax, dtype=np.int64)
    n_obs = C.c_int32()
    n_frames = C.c_int32()
    rc = lib().adh_oracle_get_dense(
        m_dia.ref(),
        C.c_int64(int(frame_start)),
        C.c_int64(int(frame_stop)),
        mzq.ctypes.data_as(C.POINTER(C.c_float)),
        C.c_int32(K),
        C.c_float(float(tol)),
        C.c_double(float(quad_lo)),
        C.c_double(float(quad_hi)),
        C.c_int32(int(bool(absolute))),
        dense.ctypes.data_as(C.POINTER(C.c_float)),
        C.c_int64(cap),
        pidx.ctypes.data_as(C.POINTER(C.c_int64)),
        C.byref(n_obs),
        C.byref(n_frames),
    )
    if rc != 0:
        raise RuntimeError(f"adh_oracle_get_dense failed: {rc}")
    O, F = n_obs.value, n_frames.value
    return dense[: 2 * K * O * 2 * F].reshape(2, K, O, 2, F).copy(), pidx[:O].copy()


def search_sorted_left(arr, value) -> int:
    a = np.ascontiguousarray(arr, dtype=np.float32)
    return int(
        lib().adh_oracle_search_sorted_left(
            a.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(a.shape[0]), C.c_float(value)
        )
    )


def center_envelope_1d(x):
    a = np.ascontiguousarray(x, dtype=np.float32).copy()
    rows, n = a.shape
    lib().adh_oracle_center_envelope(
        a.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(rows), C.c_int32(n)
    )
    return a


def fragment_correlation(x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    K, O, N = a.shape
    out = np.zeros((O, K, K), dtype=np.float32)
    lib().adh_oracle_fragment_correlation(
        a.ctypes.data_as(C.POINTER(C.c_float)),
        C.c_int32(K),
        C.c_int32(O),
        C.c_int32(N),
        out.ctypes.data_as(C.POINTER(C.c_float)),
    )
    return out


def save_corrcoeff(x, y) -> float:
    a = np.ascontiguousarray(x, dtype=np.float32)
    b = np.ascontiguousarray(y, dtype=np.float32)
    return float(
        lib().adh_oracle_save_corrcoeff(
            a.ctypes.data_as(C.POINTER(C.c_float)),
            b.ctypes.data_as(C.POINTER(C.c_float)),
            C.c_int32(a.shape[0]),
        )
    )


def quadrupole_transfer_function(cycle, observation_indices, scan_indices, isotope_mz):
    cy = np.ascontiguousarray(cycle, dtype=np.float64)
    obs = np.ascontiguousarray(observation_indices, dtype=np.int64)
    sc = np.ascontiguousarray(scan_indices, dtype=np.int64)
    iso = np.ascontiguousarray(isotope_mz, dtype=np.float64)
    out = np.zeros((iso.shape[0], obs.shape[0], sc.shape[0]), dtype=np.float64)
    lib().adh_oracle_qtf(
        cy.ctypes.data_as(C.POINTER(C.c_double)),
        C.c_int32(cy.shape[2]),
        obs.ctypes.data_as(C.POINTER(C.c_int64)),
        C.c_int32(obs.shape[0]),
        sc.ctypes.data_as(C.POINTER(C.c_int64)),
        C.c_int32(sc.shape[0]),
        iso.ctypes.data_as(C.POINTER(C.c_double)),
        C.c_int32(iso.shape[0]),
        out.ctypes.data_as(C.POINTER(C.c_double)),
    )
    return out


def fragcomp(window_start, window_stop, rt, frag_start, frag_stop, fragment_mz, rt_tol, mass_tol, n_threads=1,
             valid=None):
    ws = np.ascontiguousarray(window_start, dtype=np.int64)
    we = np.ascontiguousarray(window_stop, dtype=np.int64)
    rtv = np.ascontiguousarray(rt, dtype=np.float32)
    fs = np.ascontiguousarray(frag_start, dtype=np.int64)
    fe = np.ascontiguousarray(frag_stop, dtype=np.int64)
    fm = np.ascontiguousarray(fragment_mz, dtype=np.float32)
    valid = (np.ones(rtv.shape[0], dtype=np.uint8) if valid is None
             else np.array(valid, dtype=bool).astype(np.uint8))
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
    rc = lib().adh_oracle_fragcomp(
        C.c_int64(ws.shape[0]),
        p(ws, C.c_int64),
        p(we, C.c_int64),
        p(rtv, C.c_float),
        p(fs, C.c_int64),
        p(fe, C.c_int64),
        p(fm, C.c_float),
        C.c_double(rt_tol),
        C.c_double(mass_tol),
        p(valid, C.c_uint8),
        C.c_int(n_threads),
    )
    if rc != 0:
        raise RuntimeError(f"adh_oracle_fragcomp failed: {rc}")
    return valid.view(np.bool_)


def select(dia, fragment_cols, precursors_marshalled, cfg, kernel, n_threads: int = 1) -> dict:
    """Restated ``_select_candidates_pjit`` (selection.py:78-203) over a packed precursor table.

    Returns the full CandidateContainer arrays (rows without a candidate keep score 0)."""
    m_dia = _abi.pack_alpharaw(dia)
    m_frag = _abi.pack_fragments(*fragment_cols)
    c = _abi.pack_selection_config(cfg)
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    n = int(precursors_marshalled.struct.n) * int(c.candidate_count)
    m_out, arrays = _abi.alloc_candidate_table(n)
    rc = lib().adh_oracle_select(
        m_dia.ref(), m_frag.ref(), precursors_marshalled.ref(), C.byref(c),
        k.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(k.shape[0]), C.c_int32(k.shape[1]),
        m_out.ref(), C.c_int32(n_threads),
    )
    if rc != 0:
        raise RuntimeError(f"adh_oracle_select failed ({rc})")
    return arrays


def symetric_limits_1d(a, center, f, center_fraction, min_size, max_size):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.zeros(2, dtype=np.int32)
    lib().adh_oracle_symetric_limits_1d(
        a.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(a.shape[0]), C.c_int32(int(center)), C.c_double(f),
        C.c_double(center_fraction), C.c_int64(int(min_size)), C.c_int64(int(max_size)),
        out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def find_peaks_1d(score_row, top_n):
    a = np.ascontiguousarray(score_row, dtype=np.float64)
    cyc = np.zeros(max(a.shape[0], 1), dtype=np.int32)
    val = np.zeros(max(a.shape[0], 1), dtype=np.float64)
    n = lib().adh_oracle_find_peaks_1d(a.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(a.shape[0]),
                                       C.c_int32(int(top_n)), cyc.ctypes.data_as(C.POINTER(C.c_int32)),
                                       val.ctypes.data_as(C.POINTER(C.c_double)))
    return cyc[:n], val[:n]


def select_timstof(dia, fragment_cols, precursors_marshalled, cfg, kernel, n_threads: int = 1) -> dict:
    """Restated candidate selection on an ion-mobility run (TimsTOFTransposeJIT fields)."""
    m_dia = _abi.pack_timstof(dia)
    m_frag = _abi.pack_fragments(*fragment_cols)
    c = _abi.pack_selection_config(cfg)
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    n = int(precursors_marshalled.struct.n) * int(c.candidate_count)
    m_out, arrays = _abi.alloc_candidate_table(n)
    rc = lib().adh_oracle_select_timstof(
        m_dia.ref(), m_frag.ref(), precursors_marshalled.ref(), C.byref(c),
        k.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(k.shape[0]), C.c_int32(k.shape[1]),
        m_out.ref(), C.c_int32(n_threads),
    )
    if rc != 0:
        raise RuntimeError(f"adh_oracle_select_timstof failed ({rc})")
    return arrays


# ---- test hooks of the selection restatement (adh_oracle.cpp: select_oracle::smooth_log_hook_t) -------------------
_SMOOTH_LOG_T = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_float))
_SCORE_T = C.CFUNCTYPE(None, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_double))
_hooks_alive = []  # (ctypes callbacks must outlive their registration)


def set_selection_hooks(smooth_log=None, score=None) -> None:
    """``smooth_log(tile: float32 (S, F)) -> float32 (S, F)`` replaces the restatement's exact circular convolution AND
    the log (it returns ``log(smooth + 1)``); ``score(precursor: int, matrix: float64 (S, F))`` sees every score
    matrix.  ``None`` switches a hook off.  Single-threaded calls only (``n_threads=1``)."""
    def wrap_smooth(tile, S, F, out):
        a = np.ctypeslib.as_array(tile, shape=(S, F))
        res = np.ascontiguousarray(smooth_log(a), dtype=np.float32)
        assert res.shape == (S, F)
        np.ctypeslib.as_array(out, shape=(S, F))[:] = res

    def wrap_score(i, S, F, p):
        score(int(i), np.ctypeslib.as_array(p, shape=(S, F)).copy())

    cs = _SMOOTH_LOG_T(wrap_smooth) if smooth_log is not None else C.cast(None, _SMOOTH_LOG_T)
    cc = _SCORE_T(wrap_score) if score is not None else C.cast(None, _SCORE_T)
    _hooks_alive[:] = [cs, cc]
    fn = lib().adh_oracle_set_selection_hooks
    fn.argtypes = [_SMOOTH_LOG_T, _SCORE_T]
    fn.restype = None
    fn(cs, cc)
