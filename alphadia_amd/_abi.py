"""ctypes mirror of ``include/alphadia_hip.h`` (struct layouts + array marshalling).

Pure data-layout code: no compute happens here.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

NUM_FEATURES = 46  # alphadia/constants/settings.py:5
FLAG_SKIP = 1

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


class AlphaRaw(C.Structure):
    _fields_ = [
        ("cycle", _f64p),
        ("cycle_len", C.c_int32),
        ("cycle_scans", C.c_int32),
        ("rt_values", _f32p),
        ("n_spectra", C.c_int64),
        ("mobility_values", _f32p),
        ("n_mobility", C.c_int64),
        ("peak_start_idx", _i64p),
        ("peak_stop_idx", _i64p),
        ("mz_values", _f32p),
        ("intensity_values", _f32p),
        ("n_peaks", C.c_int64),
    ]


class TimsTOF(C.Structure):
    _fields_ = [
        ("cycle", _f64p),
        ("cycle_len", C.c_int32),
        ("scan_max_index", C.c_int32),
        ("dia_precursor_cycle", _i64p),
        ("rt_values", _f64p),
        ("n_frames", C.c_int64),
        ("mobility_values", _f64p),
        ("mz_values", _f64p),
        ("n_tof", C.c_int64),
        ("tof_indptr", _i64p),
        ("push_indices", _u32p),
        ("intensity_values", C.POINTER(C.c_uint16)),
        ("n_events", C.c_int64),
        ("zeroth_frame", C.c_int32),
    ]


class Fragments(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("mz_library", _f32p),
        ("mz", _f32p),
        ("intensity", _f32p),
        ("type", _u8p),
        ("loss_type", _u8p),
        ("charge", _u8p),
        ("number", _u8p),
        ("position", _u8p),
        ("cardinality", _u8p),
    ]


class Candidates(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("precursor_idx", _u32p),
        ("rank", _u8p),
        ("flags", _u8p),
        ("frag_start_idx", _u32p),
        ("frag_stop_idx", _u32p),
        ("scan_start", _i64p),
        ("scan_stop", _i64p),
        ("scan_center", _i64p),
        ("frame_start", _i64p),
        ("frame_stop", _i64p),
        ("frame_center", _i64p),
        ("charge", _u8p),
        ("precursor_mz", _f32p),
        ("isotope_intensity", _f32p),
        ("n_isotope_cols", C.c_int32),
    ]


class ScoringConfig(C.Structure):
    _fields_ = [
        ("collect_fragments", C.c_int32),
        ("score_grouped", C.c_int32),
        ("exclude_shared_ions", C.c_int32),
        ("top_k_fragments", C.c_uint32),
        ("top_k_isotopes", C.c_uint32),
        ("reference_channel", C.c_int32),
        ("quant_window", C.c_uint32),
        ("quant_all", C.c_int32),
        ("precursor_mz_tolerance", C.c_float),
        ("fragment_mz_tolerance", C.c_float),
        ("experimental_xic", C.c_int32),
        ("quadrupole_sigma", C.c_double * 2),
        ("quadrupole_delta_mu", C.c_double * 2),
    ]


class Output(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("top_k", C.c_int32),
        ("valid", _u8p),
        ("precursor_idx", _u32p),
        ("rank", _u8p),
        ("features", _f32p),
        ("fragment_precursor_idx", _u32p),
        ("fragment_rank", _u8p),
        ("fragment_mz_library", _f32p),
        ("fragment_mz", _f32p),
        ("fragment_mz_observed", _f32p),
        ("fragment_height", _f32p),
        ("fragment_intensity", _f32p),
        ("fragment_mass_error", _f32p),
        ("fragment_correlation", _f32p),
        ("fragment_position", _u8p),
        ("fragment_number", _u8p),
        ("fragment_type", _u8p),
        ("fragment_charge", _u8p),
        ("fragment_loss_type", _u8p),
        ("stat_matched_peaks", _u32p),
        ("fragment_lib_slot", C.POINTER(C.c_uint16)),
    ]


class CompactOutput(C.Structure):
    """adh_compact_output_t (include/alphadia_hip.h): valid candidates and filled fragment slots, column by column."""

    _fields_ = [
        ("rows_capacity", C.c_int64),
        ("slots_capacity", C.c_int64),
        ("top_k", C.c_int32),
        ("reserved", C.c_int32),
        ("n_rows", C.c_int64),
        ("n_slots", C.c_int64),
        ("row", _u32p),
        ("precursor_idx", _u32p),
        ("rank", _u8p),
        ("features", _f32p),
        ("fragment_row", _u32p),
        ("fragment_precursor_idx", _u32p),
        ("fragment_rank", _u8p),
        ("fragment_mz_library", _f32p),
        ("fragment_mz", _f32p),
        ("fragment_mz_observed", _f32p),
        ("fragment_height", _f32p),
        ("fragment_intensity", _f32p),
        ("fragment_mass_error", _f32p),
        ("fragment_correlation", _f32p),
        ("fragment_position", _u8p),
        ("fragment_number", _u8p),
        ("fragment_type", _u8p),
        ("fragment_charge", _u8p),
        ("fragment_loss_type", _u8p),
    ]


# per-row and per-slot columns of CompactOutput: (name, dtype)
COMPACT_ROW_FIELDS = [("row", np.uint32), ("precursor_idx", np.uint32), ("rank", np.uint8)]
COMPACT_SLOT_FIELDS = [
    ("fragment_row", np.uint32), ("fragment_precursor_idx", np.uint32), ("fragment_rank", np.uint8),
    ("fragment_mz_library", np.float32), ("fragment_mz", np.float32), ("fragment_mz_observed", np.float32),
    ("fragment_height", np.float32), ("fragment_intensity", np.float32), ("fragment_mass_error", np.float32),
    ("fragment_correlation", np.float32), ("fragment_position", np.uint8), ("fragment_number", np.uint8),
    ("fragment_type", np.uint8), ("fragment_charge", np.uint8), ("fragment_loss_type", np.uint8),
]


# (name, dtype, per-row width: 1 | "features" | "top_k")
OUTPUT_FIELDS = [
    ("valid", np.uint8, 1),
    ("precursor_idx", np.uint32, 1),
    ("rank", np.uint8, 1),
    ("features", np.float32, "features"),
    ("fragment_precursor_idx", np.uint32, "top_k"),
    ("fragment_rank", np.uint8, "top_k"),
    ("fragment_mz_library", np.float32, "top_k"),
    ("fragment_mz", np.float32, "top_k"),
    ("fragment_mz_observed", np.float32, "top_k"),
    ("fragment_height", np.float32, "top_k"),
    ("fragment_intensity", np.float32, "top_k"),
    ("fragment_mass_error", np.float32, "top_k"),
    ("fragment_correlation", np.float32, "top_k"),
    ("fragment_position", np.uint8, "top_k"),
    ("fragment_number", np.uint8, "top_k"),
    ("fragment_type", np.uint8, "top_k"),
    ("fragment_charge", np.uint8, "top_k"),
    ("fragment_loss_type", np.uint8, "top_k"),
]


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


_CT = {
    np.dtype(np.float32): C.c_float,
    np.dtype(np.float64): C.c_double,
    np.dtype(np.int64): C.c_int64,
    np.dtype(np.uint32): C.c_uint32,
    np.dtype(np.uint8): C.c_uint8,
    np.dtype(np.uint16): C.c_uint16,
}


def as_c(a, dtype) -> np.ndarray:
    """Contiguous array of exactly ``dtype`` (no copy when already so)."""
    a = np.asarray(a)
    if a.dtype == np.bool_ and np.dtype(dtype) == np.uint8:
        a = a.view(np.uint8)
    return np.ascontiguousarray(a, dtype=dtype)


class Marshalled:
    """A ctypes struct plus the numpy arrays that keep its pointers alive."""

    def __init__(self, struct, keep):
        self.struct = struct
        self.keep = keep

    def ref(self):
        return C.byref(self.struct)


def pack_alpharaw(dia) -> Marshalled:
    """``dia`` exposes the AlphaRawJIT field names (alpharaw_jit.py:78-138)."""
    cycle = as_c(dia.cycle, np.float64)
    if cycle.ndim != 4 or cycle.shape[0] != 1 or cycle.shape[3] != 2:
        raise ValueError("cycle must have shape (1, n_precursor, n_scan, 2)")
    rt = as_c(dia.rt_values, np.float32)
    mob = as_c(dia.mobility_values, np.float32)
    ps = as_c(dia.peak_start_idx_list, np.int64)
    pe = as_c(dia.peak_stop_idx_list, np.int64)
    mz = as_c(dia.mz_values, np.float32)
    it = as_c(dia.intensity_values, np.float32)
    if ps.shape != rt.shape or pe.shape != rt.shape:
        raise ValueError("peak_start/stop_idx_list must have one entry per spectrum")
    if mz.shape != it.shape:
        raise ValueError("mz_values and intensity_values differ in length")
    s = AlphaRaw(
        _ptr(cycle, C.c_double),
        cycle.shape[1],
        cycle.shape[2],
        _ptr(rt, C.c_float),
        rt.shape[0],
        _ptr(mob, C.c_float),
        mob.shape[0],
        _ptr(ps, C.c_int64),
        _ptr(pe, C.c_int64),
        _ptr(mz, C.c_float),
        _ptr(it, C.c_float),
        mz.shape[0],
    )
    return Marshalled(s, [cycle, rt, mob, ps, pe, mz, it])


def pack_timstof(dia) -> Marshalled:
    """``dia`` exposes the TimsTOFTransposeJIT field names (bruker_jit.py:22-137)."""
    cycle = as_c(dia.cycle, np.float64)
    if cycle.ndim != 4 or cycle.shape[0] != 1 or cycle.shape[3] != 2:
        raise ValueError("cycle must have shape (1, n_frames_per_cycle, n_scans, 2)")
    if cycle.shape[2] != int(dia.scan_max_index):
        raise ValueError("cycle scan axis must equal scan_max_index")
    dpc = as_c(dia.dia_precursor_cycle, np.int64)
    rt = as_c(dia.rt_values, np.float64)
    mob = as_c(dia.mobility_values, np.float64)
    mz = as_c(dia.mz_values, np.float64)
    ptr = as_c(dia.tof_indptr, np.int64)
    push = as_c(dia.push_indices, np.uint32)
    inten = np.ascontiguousarray(dia.intensity_values, dtype=np.uint16)
    if dpc.shape[0] != cycle.shape[1] * cycle.shape[2]:
        raise ValueError("dia_precursor_cycle must have one entry per (frame, scan) of the cycle")
    if ptr.shape[0] != mz.shape[0] + 1 or push.shape != inten.shape:
        raise ValueError("inconsistent TOF index arrays")
    s = TimsTOF(
        _ptr(cycle, C.c_double), cycle.shape[1], cycle.shape[2], _ptr(dpc, C.c_int64),
        _ptr(rt, C.c_double), rt.shape[0], _ptr(mob, C.c_double), _ptr(mz, C.c_double), mz.shape[0],
        _ptr(ptr, C.c_int64), _ptr(push, C.c_uint32), inten.ctypes.data_as(C.POINTER(C.c_uint16)),
        push.shape[0], int(bool(dia.zeroth_frame)),
    )
    return Marshalled(s, [cycle, dpc, rt, mob, mz, ptr, push, inten])


def pack_fragments(mz_library, mz, intensity, type_, loss_type, charge, number, position, cardinality):
    arrs = [as_c(mz_library, np.float32), as_c(mz, np.float32), as_c(intensity, np.float32)]
    arrs += [as_c(a, np.uint8) for a in (type_, loss_type, charge, number, position, cardinality)]
    n = arrs[0].shape[0]
    if any(a.shape != (n,) for a in arrs):
        raise ValueError("fragment columns differ in length")
    s = Fragments(
        n,
        *[_ptr(a, C.c_float) for a in arrs[:3]],
        *[_ptr(a, C.c_uint8) for a in arrs[3:]],
    )
    return Marshalled(s, arrs)


def pack_candidates(
    precursor_idx,
    rank,
    frag_start_idx,
    frag_stop_idx,
    scan_start,
    scan_stop,
    scan_center,
    frame_start,
    frame_stop,
    frame_center,
    charge,
    precursor_mz,
    isotope_intensity,
    flags=None,
) -> Marshalled:
    pi = as_c(precursor_idx, np.uint32)
    n = pi.shape[0]
    rk = as_c(rank, np.uint8)
    fs = as_c(frag_start_idx, np.uint32)
    fe = as_c(frag_stop_idx, np.uint32)
    i64 = [
        as_c(a, np.int64)
        for a in (scan_start, scan_stop, scan_center, frame_start, frame_stop, frame_center)
    ]
    ch = as_c(charge, np.uint8)
    pm = as_c(precursor_mz, np.float32)
    iso = as_c(isotope_intensity, np.float32)
    if iso.ndim != 2 or iso.shape[0] != n:
        raise ValueError("isotope_intensity must be (n_candidates, n_isotopes)")
    if iso.shape[1] == 0:
        raise ValueError("precursor isotopes empty")  # score_group.py:198-199
    fl = as_c(flags, np.uint8) if flags is not None else None
    for a in [rk, fs, fe, ch, pm, *i64] + ([fl] if fl is not None else []):
        if a.shape != (n,):
            raise ValueError("candidate columns differ in length")
    s = Candidates(
        n,
        _ptr(pi, C.c_uint32),
        _ptr(rk, C.c_uint8),
        _ptr(fl, C.c_uint8) if fl is not None else None,
        _ptr(fs, C.c_uint32),
        _ptr(fe, C.c_uint32),
        *[_ptr(a, C.c_int64) for a in i64],
        _ptr(ch, C.c_uint8),
        _ptr(pm, C.c_float),
        _ptr(iso, C.c_float),
        iso.shape[1],
    )
    return Marshalled(s, [pi, rk, fl, fs, fe, *i64, ch, pm, iso])


def output_width(cands: "Marshalled", top_k: int) -> int:
    """Columns the fragment tables need: ``top_k_fragments`` clamped to the longest library slice
    of the table.  The reference allocates ``top_k_fragments`` columns literally (output.py:44-70),
    9999 of them in transfer-library requantification
    (transfer_library_requantification_handler.py:117-124); the columns beyond the longest slice
    can never be filled (fragment_container.py:75-90), so the frames built from them are the same."""
    st = cands.struct
    n = int(st.n)
    if n == 0:
        return max(1, min(int(top_k), 1))
    # (kept on the marshalled table: a table scored again - every step of the bench, every pass of the optimisation
    # loop - pays the pass over its 3e6 rows once; unsigned arithmetic: a slice with stop < start wraps to a huge
    # length, the clamp below makes that top_k, and the C side rejects the table)
    longest = getattr(cands, "_longest_slice", None)
    if longest is None:
        start = np.ctypeslib.as_array(st.frag_start_idx, shape=(n,))
        stop = np.ctypeslib.as_array(st.frag_stop_idx, shape=(n,))
        longest = int((stop - start).max())
        try:
            cands._longest_slice = longest
        except AttributeError:
            pass
    return max(1, min(int(top_k), max(longest, 1)))


def output_shapes(n: int, top_k: int, extras: bool = False):
    shapes = {}
    for name, dt, w in OUTPUT_FIELDS:
        if w == 1:
            shapes[name] = ((n,), dt)
        elif w == "features":
            shapes[name] = ((n, NUM_FEATURES), dt)
        else:
            shapes[name] = ((n, top_k), dt)
    if extras:
        shapes["stat_matched_peaks"] = ((n,), np.uint32)
        shapes["fragment_lib_slot"] = ((n, top_k), np.uint16)
    return shapes


def alloc_output(n: int, top_k: int, with_stats: bool = False, zero: bool = True, alloc=None,
                 with_slots: bool = True):
    """Host OutputPsmDF buffers (output.py:44-70) and the ctypes view of them; ``zero=False`` skips the
    fill for callers that overwrite every byte (``adh_score_candidates`` copies whole tables back).
    ``fragment_lib_slot`` (1 + position of a slot's fragment in the library slice) rides along.
    ``alloc(name, shape, dtype)`` supplies the buffers (e.g. page-locked ones) instead of numpy."""
    if alloc is None:
        new = np.zeros if zero else np.empty
        alloc = lambda name, shape, dt: new(shape, dtype=dt)  # noqa: E731
    arrays = {k: alloc(k, shape, dt) for k, (shape, dt) in output_shapes(n, top_k).items()}
    stats = alloc("stat_matched_peaks", (n,), np.uint32) if with_stats else None
    slots = alloc("fragment_lib_slot", (n, top_k), np.uint16) if with_slots else None
    s = Output(
        n,
        top_k,
        *[_ptr(arrays[name], _CT[np.dtype(dt)]) for name, dt, _ in OUTPUT_FIELDS],
        _ptr(stats, C.c_uint32) if stats is not None else None,
        _ptr(slots, C.c_uint16) if slots is not None else None,
    )
    if stats is not None:
        arrays["stat_matched_peaks"] = stats
    if slots is not None:
        arrays["fragment_lib_slot"] = slots
    return Marshalled(s, arrays), arrays


def output_from_device_pointers(n: int, top_k: int, ptrs: dict, stats_ptr: int = 0, slot_ptr: int = 0) -> Output:
    """Build an ``adh_output_t`` from raw device addresses (ints)."""
    vals = []
    for name, dt, _ in OUTPUT_FIELDS:
        vals.append(C.cast(C.c_void_p(ptrs[name]), C.POINTER(_CT[np.dtype(dt)])))
    st = C.cast(C.c_void_p(stats_ptr), C.POINTER(C.c_uint32)) if stats_ptr else None
    sl = C.cast(C.c_void_p(slot_ptr), C.POINTER(C.c_uint16)) if slot_ptr else None
    return Output(n, top_k, *vals, st, sl)


def pack_config(cfg) -> ScoringConfig:
    """``cfg`` exposes the CandidateScoringConfigJIT attribute names (config.py:13-60)."""
    return ScoringConfig(
        int(bool(cfg.collect_fragments)),
        int(bool(cfg.score_grouped)),
        int(bool(cfg.exclude_shared_ions)),
        int(cfg.top_k_fragments),
        int(cfg.top_k_isotopes),
        int(cfg.reference_channel),
        int(cfg.quant_window),
        int(bool(cfg.quant_all)),
        float(cfg.precursor_mz_tolerance),
        float(cfg.fragment_mz_tolerance),
        int(bool(cfg.experimental_xic)),
        _pair(getattr(cfg, "quadrupole_sigma", None), 0.0),  # (0: the class defaults, sigma 0.2 / delta_mu 0)
        _pair(getattr(cfg, "quadrupole_delta_mu", None), 0.0),
    )


def _pair(values, default: float):
    out = (C.c_double * 2)(default, default)
    if values is not None:
        out[0], out[1] = float(values[0]), float(values[1])
    return out


# ---------------------------------------------------------------------------
# candidate selection (include/alphadia_hip.h: adh_precursors_t, adh_selection_config_t,
# adh_candidate_table_t)


class Precursors(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("precursor_idx", C.POINTER(C.c_uint32)),
        ("frag_start_idx", C.POINTER(C.c_uint32)),
        ("frag_stop_idx", C.POINTER(C.c_uint32)),
        ("charge", C.POINTER(C.c_uint8)),
        ("rt", C.POINTER(C.c_float)),
        ("mobility", C.POINTER(C.c_float)),
        ("mz", C.POINTER(C.c_float)),
        ("isotope_intensity", C.POINTER(C.c_float)),
        ("n_isotope_cols", C.c_int32),
    ]


class SelectionConfig(C.Structure):
    _fields_ = [
        ("rt_tolerance", C.c_double),
        ("precursor_mz_tolerance", C.c_double),
        ("fragment_mz_tolerance", C.c_double),
        ("candidate_count", C.c_int64),
        ("top_k_precursors", C.c_int64),
        ("kernel_size", C.c_int64),
        ("f_mobility", C.c_double),
        ("f_rt", C.c_double),
        ("center_fraction", C.c_double),
        ("min_size_mobility", C.c_int64),
        ("min_size_rt", C.c_int64),
        ("max_size_mobility", C.c_int64),
        ("max_size_rt", C.c_int64),
        ("join_close_candidates_scan_threshold", C.c_double),
        ("join_close_candidates_cycle_threshold", C.c_double),
        ("feature_mean", C.c_double),
        ("feature_std", C.c_double),
        ("feature_weight", C.c_double),
        ("exclude_shared_ions", C.c_uint8),
        ("use_weighted_score", C.c_uint8),
        ("join_close_candidates", C.c_uint8),
        ("pad0", C.c_uint8),
        ("pad1", C.c_uint8 * 4),
        ("mobility_tolerance", C.c_double),
    ]


CANDIDATE_TABLE_FIELDS = [
    ("precursor_idx", np.uint32),
    ("rank", np.uint8),
    ("score", np.float32),
    ("scan_center", np.uint32),
    ("scan_start", np.uint32),
    ("scan_stop", np.uint32),
    ("frame_center", np.uint32),
    ("frame_start", np.uint32),
    ("frame_stop", np.uint32),
]


class CandidateTable(C.Structure):
    _fields_ = [("n", C.c_int64)] + [
        (name, C.POINTER(_CT[np.dtype(dt)])) for name, dt in CANDIDATE_TABLE_FIELDS
    ]


def pack_precursors(precursor_idx, frag_start_idx, frag_stop_idx, charge, rt, mobility, mz,
                    isotope_intensity) -> Marshalled:
    """The columns of PrecursorFlatContainer (selection.py:712-737), sorted by precursor_idx."""
    iso = as_c(isotope_intensity, np.float32)
    if iso.ndim != 2:
        raise ValueError("isotope_intensity must be 2-D (precursors x isotopes)")
    u32 = [as_c(a, np.uint32) for a in (precursor_idx, frag_start_idx, frag_stop_idx)]
    ch = as_c(charge, np.uint8)
    f32 = [as_c(a, np.float32) for a in (rt, mobility, mz)]
    n = u32[0].shape[0]
    if any(a.shape != (n,) for a in u32 + [ch] + f32) or iso.shape[0] != n:
        raise ValueError("precursor columns differ in length")
    s = Precursors(n, *[_ptr(a, C.c_uint32) for a in u32], _ptr(ch, C.c_uint8),
                   *[_ptr(a, C.c_float) for a in f32], _ptr(iso, C.c_float), iso.shape[1])
    return Marshalled(s, u32 + [ch] + f32 + [iso])


def pack_selection_config(cfg) -> SelectionConfig:
    """``cfg`` exposes the CandidateSelectionConfigJIT fields (config_df.py:15-110)."""

    def first(x):
        a = np.atleast_1d(np.asarray(x, dtype=np.float64))
        if a.size != 1:
            raise ValueError("only the single-feature score of the reference is supported")
        return float(a[0])

    return SelectionConfig(
        float(cfg.rt_tolerance), float(cfg.precursor_mz_tolerance), float(cfg.fragment_mz_tolerance),
        int(cfg.candidate_count), int(cfg.top_k_precursors), int(cfg.kernel_size),
        float(cfg.f_mobility), float(cfg.f_rt), float(cfg.center_fraction),
        int(cfg.min_size_mobility), int(cfg.min_size_rt), int(cfg.max_size_mobility), int(cfg.max_size_rt),
        float(cfg.join_close_candidates_scan_threshold), float(cfg.join_close_candidates_cycle_threshold),
        first(cfg.feature_mean), first(cfg.feature_std), first(cfg.feature_weight),
        int(bool(cfg.exclude_shared_ions)), int(bool(cfg.use_weighted_score)),
        int(bool(cfg.join_close_candidates)), 0, (C.c_uint8 * 4)(),
        float(getattr(cfg, "mobility_tolerance", 0.0)),
    )


def alloc_candidate_table(n_rows: int):
    arrays = {name: np.zeros(n_rows, dtype=dt) for name, dt in CANDIDATE_TABLE_FIELDS}
    s = CandidateTable(n_rows, *[_ptr(arrays[name], _CT[np.dtype(dt)]) for name, dt in CANDIDATE_TABLE_FIELDS])
    return Marshalled(s, arrays), arrays


# ---- FDR stage (include/alphadia_hip.h: adh_mlp_arch_t, adh_mlp_fit_t) ----------------------
MLP_MAX_LINEAR = 8


class MlpArch(C.Structure):
    _fields_ = [
        ("n_linear", C.c_int32),
        ("dims", C.c_int32 * (MLP_MAX_LINEAR + 1)),
        ("bn_eps", C.c_float),
        ("bn_momentum", C.c_float),
    ]


class MlpFit(C.Structure):
    _fields_ = [
        ("train_rows", _i64p),
        ("n_train", C.c_int64),
        ("batch_start", _i64p),
        ("n_steps", C.c_int64),
        ("batch_size", C.c_int32),
        ("learning_rate", C.c_float),
        ("weight_decay", C.c_float),
        ("dropout", C.c_float),
        ("beta1", C.c_float),
        ("beta2", C.c_float),
        ("eps", C.c_float),
        ("seed", C.c_uint64),
        ("first_step", C.c_int64),
    ]


def pack_mlp_arch(input_dim: int, layers, output_dim: int, bn_eps: float = 1e-5, bn_momentum: float = 0.1) -> MlpArch:
    dims = [int(input_dim), *[int(x) for x in layers], int(output_dim)]
    if len(dims) - 1 > MLP_MAX_LINEAR:
        raise ValueError(f"at most {MLP_MAX_LINEAR - 1} hidden layers")
    a = MlpArch()
    a.n_linear = len(dims) - 1
    for i, v in enumerate(dims):
        a.dims[i] = v
    a.bn_eps = bn_eps
    a.bn_momentum = bn_momentum
    return a
