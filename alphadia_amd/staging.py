"""Staging a run straight from the reference's array contract.

The reference turns an AlphaRaw spectrum table + peak table into the arrays of ``AlphaRawJIT`` in
two steps: ``determine_dia_cycle`` (alphadia/raw_data/dia_cycle.py:18-82: period of the isolation
window pattern, first complete cycle, consistency check) and ``AlphaRaw._preprocess_raw_data``
(alphadia/raw_data/alpharaw_wrapper.py:72-117: drop an MS1 that does not follow the cycle, cut the
non-DIA prefix, convert units and dtypes).  This module does the same on plain column arrays -
vectorised, no per-spectrum Python or jit loops - and hands the result to ``adh_stage_alpharaw``,
so a vendor reader only has to deliver the two tables.  Decisions (cycle length, cycle start,
validity, has_ms1) are pinned against outputs of the reference's own functions
(tests/golden/staging.npz).

A cached device image of a staged run is deliberately not offered for this layout: the transposed
copy is larger than the raw arrays and is rebuilt on the GPU in ~0.25 s for a 2 h run.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

DEFAULT_VALUE_NO_MOBILITY = 1e-6  # alpharaw_wrapper.py:14


class NotValidDiaDataError(ValueError):
    """The spectra do not follow one repeating DIA cycle (alphadia/exceptions.py)."""


def _autocorrelation(x: np.ndarray) -> np.ndarray:
    """Autocorrelation of the mean-free signal at lags 0..n-1, normalised to lag 0."""
    x = x - x.mean()
    c = np.correlate(x, x, mode="full")[len(x) - 1:]
    return c / c[0]


def cycle_length(signature: np.ndarray) -> int:
    """Spectra per DIA cycle: the lag of the highest local maximum of the autocorrelation of the
    window signature (dia_cycle.py:85-113); -1 when the autocorrelation has no interior peak."""
    corr = _autocorrelation(np.asarray(signature, dtype=np.float64))
    interior = corr[1:-1]
    peaks = np.flatnonzero((interior > corr[:-2]) & (interior > corr[2:])) + 1
    return int(peaks[np.argmax(corr[peaks])]) if len(peaks) else -1


def _repeats(signature: np.ndarray, period: int) -> np.ndarray:
    """ok[i]: the ``period`` values from i on equal the ``period`` values after them."""
    n = len(signature)
    same = signature[: n - period] == signature[period:]
    run = np.concatenate([[0], np.cumsum(same)])
    starts = np.arange(max(n - 2 * period + 1, 0))
    return (run[starts + period] - run[starts]) == period


def cycle_start(signature: np.ndarray, period: int) -> int:
    """First spectrum from which two consecutive windows of ``period`` spectra are identical and the
    window itself is not constant (a constant stretch is the settling phase before the method
    cycles; dia_cycle.py:136-174); -1 when there is none."""
    signature = np.asarray(signature)
    n = len(signature)
    last = n - 2 * period  # the reference scans i < n - 2 * period
    if last <= 0:
        return -1
    repeats = _repeats(signature, period)[:last]
    differs = signature[1:] != signature[:-1]
    run = np.concatenate([[0], np.cumsum(differs)])
    idx = np.arange(last)
    constant = (run[idx + period - 1] - run[idx]) == 0  # no change inside [i, i + period)
    hit = np.flatnonzero(repeats & ~constant)
    return int(hit[0]) if len(hit) else -1


def cycle_is_consistent(signature: np.ndarray, period: int, start: int) -> bool:
    """Every window of ``period`` spectra from ``start`` on equals the next one (dia_cycle.py:177-214)."""
    signature = np.asarray(signature)
    count = len(signature) - 2 * period - start
    if count <= 0:
        return True
    return bool(_repeats(signature[start:], period)[:count].all())


def determine_dia_cycle(isolation_lower_mz, isolation_upper_mz, rt=None, subset_for_cycle_detection: int = 10000):
    """``(cycle[1, L, 1, 2], cycle_start, cycle_length)`` of a run (dia_cycle.py:18-82)."""
    lower = np.asarray(isolation_lower_mz, dtype=np.float64)
    upper = np.asarray(isolation_upper_mz, dtype=np.float64)
    signature = lower[:subset_for_cycle_detection] + upper[:subset_for_cycle_detection]
    if len(signature) < 3:
        raise NotValidDiaDataError("Failed to determine length of DIA cycle.")
    period = cycle_length(signature)
    if period == -1:
        raise NotValidDiaDataError("Failed to determine length of DIA cycle.")
    start = cycle_start(signature, period)
    if start == -1:
        raise NotValidDiaDataError("Failed to determine start of DIA cycle.")
    if not cycle_is_consistent(signature, period, start):
        at = f" {float(np.asarray(rt)[start]):.2f} min" if rt is not None else f" spectrum {start}"
        raise NotValidDiaDataError(f"Cycle with start{at} and length {period} detected, but is not consistent.")
    cycle = np.zeros((1, period, 1, 2), dtype=np.float64)
    cycle[0, :, 0, 0] = lower[start : start + period]
    cycle[0, :, 0, 1] = upper[start : start + period]
    return cycle, start, period


@dataclass
class AlphaRawArrays:
    """The ``AlphaRawJIT`` fields (search/jitclasses/alpharaw_jit.py:78-138) as host arrays: what
    ``Context.stage_run`` / ``adh_stage_alpharaw`` take."""

    cycle: np.ndarray
    rt_values: np.ndarray
    peak_start_idx_list: np.ndarray
    peak_stop_idx_list: np.ndarray
    mz_values: np.ndarray
    intensity_values: np.ndarray
    mobility_values: np.ndarray
    has_ms1: bool
    cycle_start: int
    max_mz_value: np.float32
    min_mz_value: np.float32
    quad_max_mz_value: np.float32
    quad_min_mz_value: np.float32
    zeroth_frame: int = 0
    scan_max_index: int = 1
    has_mobility: bool = False

    @property
    def cycle_len(self) -> int:
        return int(self.cycle.shape[1])

    @property
    def n_spectra(self) -> int:
        return int(self.rt_values.shape[0])

    @property
    def frame_max_index(self) -> int:
        return self.n_spectra - 1

    @property
    def precursor_cycle_max_index(self) -> int:
        return self.n_spectra // self.cycle_len


def preprocess_spectra(spectrum: dict, peaks: dict) -> AlphaRawArrays:
    """Spectrum table + peak table -> run arrays (``AlphaRaw._preprocess_raw_data``).

    ``spectrum``: columns ``spec_idx, rt`` (minutes), ``ms_level, precursor_mz, isolation_lower_mz,
    isolation_upper_mz, peak_start_idx, peak_stop_idx`` (a DataFrame or a dict of arrays);
    ``peaks``: columns ``mz, intensity``."""
    col = {k: np.asarray(spectrum[k]) for k in ("spec_idx", "rt", "ms_level", "precursor_mz", "isolation_lower_mz",
                                               "isolation_upper_mz", "peak_start_idx", "peak_stop_idx")}
    # an MS1 that is not taken once per cycle (time-based loop count) cannot be used: its spacing in
    # spectrum numbers must be one single value (alpharaw_wrapper.py:119-122)
    ms1_idx = col["spec_idx"][col["ms_level"] == 1]
    has_ms1 = len(np.unique(np.diff(ms1_idx))) == 1
    keep = np.ones(len(col["rt"]), dtype=bool) if has_ms1 else col["ms_level"] > 1
    col = {k: v[keep] for k, v in col.items()}
    cycle, start, _ = determine_dia_cycle(col["isolation_lower_mz"], col["isolation_upper_mz"], col["rt"])
    col = {k: v[start:] for k, v in col.items()}
    ms2 = col["ms_level"] == 2
    return AlphaRawArrays(
        cycle=cycle,
        rt_values=col["rt"].astype(np.float32) * 60,  # minutes -> seconds, in float32 like the reference
        peak_start_idx_list=col["peak_start_idx"].astype(np.int64),
        peak_stop_idx_list=col["peak_stop_idx"].astype(np.int64),
        mz_values=np.asarray(peaks["mz"]).astype(np.float32),
        intensity_values=np.asarray(peaks["intensity"]).astype(np.float32),
        mobility_values=np.array([DEFAULT_VALUE_NO_MOBILITY, 0], dtype=np.float32),
        has_ms1=bool(has_ms1),
        cycle_start=int(start),
        max_mz_value=np.float32(col["precursor_mz"].max()),
        min_mz_value=np.float32(col["precursor_mz"].min()),
        quad_max_mz_value=np.float32(col["isolation_upper_mz"][ms2].max()),
        quad_min_mz_value=np.float32(col["isolation_lower_mz"][ms2].min()),
    )


def stage_spectra(spectrum, peaks, device: int | None = None) -> AlphaRawArrays:
    """Preprocess the two tables and stage the run in the GPU's HBM (transposed, see adh_gather.hip)."""
    from alphadia_amd import runtime

    arrays = preprocess_spectra(spectrum, peaks)
    runtime.get_context(device).stage_run(arrays)
    return arrays
