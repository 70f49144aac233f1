"""Host-side mirror of the reference's candidate selection operator (the step before scoring).

``HipCandidateSelection`` has the constructor and call signature of
``alphadia.search.selection.selection.CandidateSelection`` (selection.py:529-660) and returns the
same candidate DataFrame; the per-precursor work runs in ``adh_select_kernel`` behind the C ABI
(``adh_select_candidates``), for AlphaRaw runs and for ion-mobility (timsTOF) runs.
"""

from __future__ import annotations

import re

import numpy as np
import pandas as pd

from . import _abi, runtime

CANDIDATE_COLUMNS = [name for name, _ in _abi.CANDIDATE_TABLE_FIELDS]


class CandidateSelectionConfig:
    """Field names and defaults of the reference's ``CandidateSelectionConfig``
    (search/selection/config_df.py:113-224)."""

    _DEFAULTS = dict(
        rt_tolerance=60.0, precursor_mz_tolerance=10.0, fragment_mz_tolerance=15.0,
        mobility_tolerance=0.1, isotope_tolerance=0.01, peak_len_rt=10.0, sigma_scale_rt=0.1,
        peak_len_mobility=0.013, sigma_scale_mobility=1.0, candidate_count=5, top_k_precursors=3,
        top_k_fragments=12, exclude_shared_ions=True, kernel_size=30, f_mobility=1.0, f_rt=0.99,
        center_fraction=0.5, min_size_mobility=8, min_size_rt=3, max_size_mobility=30, max_size_rt=15,
        group_channels=False, use_weighted_score=True, join_close_candidates=True,
        join_close_candidates_scan_threshold=0.01, join_close_candidates_cycle_threshold=0.6,
    )

    def __init__(self):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        self.feature_std = np.array([1.0])
        self.feature_mean = np.array([0.0])
        self.feature_weight = np.array([1.0])

    def update(self, values: dict) -> None:
        for k, v in values.items():
            if not hasattr(self, k):
                raise KeyError(f"unknown candidate selection setting {k!r}")
            setattr(self, k, v)

    def validate(self) -> None:
        assert self.rt_tolerance >= 0 and self.precursor_mz_tolerance >= 0 and self.fragment_mz_tolerance >= 0
        assert self.candidate_count > 0 and self.top_k_precursors > 0 and self.kernel_size > 0
        assert np.size(self.feature_std) == np.size(self.feature_mean) == np.size(self.feature_weight) == 1, (
            "only the single-feature score of the reference is implemented"
        )

    def to_jitclass(self):
        self.validate()
        return self


def gaussian_kernel(dia, fwhm_rt: float, sigma_scale_rt: float, kernel_size: int,
                    fwhm_mobility: float = 0.012, sigma_scale_mobility: float = 1.0) -> np.ndarray:
    """The smoothing kernel of ``GaussianKernel.get_dense_matrix`` (selection/kernel.py:37-218):
    float32, shape (kernel_height, kernel_width).

    Restated literally, including the reference's use of sigma (not sigma squared) on the
    diagonal of the covariance matrix and its normalisation constant."""
    cycle = np.asarray(dia.cycle)
    rt_datapoints = cycle.shape[1]
    rt_values = np.asarray(dia.rt_values)
    rt_resolution = np.mean(np.diff(rt_values[::rt_datapoints]))
    sigma_rt = fwhm_rt / 2.3548 * sigma_scale_rt / rt_resolution
    if getattr(dia, "has_mobility", False):
        mobility_resolution = np.mean(np.diff(np.asarray(dia.mobility_values)[::-1]))
        sigma_mob = fwhm_mobility / 2.3548 * sigma_scale_mobility / mobility_resolution
    else:
        sigma_mob = 1.0  # determine_mobility_sigma without a mobility dimension (kernel.py:126-128)
    width = int(np.ceil(kernel_size / 2) * 2)
    height = int(np.ceil(min(kernel_size, int(dia.scan_max_index) + 1) / 2) * 2)
    x, y = np.meshgrid(np.arange(-width // 2, width // 2), np.arange(-height // 2, height // 2))
    xy = np.column_stack((x.flatten(), y.flatten())).astype("float32")
    sigma = np.array([[sigma_rt, 0.0], [0.0, sigma_mob]])
    dx = xy - np.array([[0.0, 0.0]])
    a = np.exp(-1 / 2 * np.einsum("ij,jk,ik->i", dx, np.linalg.inv(sigma), dx))
    b = (np.pi * 2) ** (-1 / 2) * np.linalg.det(sigma) ** (-1 / 2)  # mu.shape[0] == 1 in the reference
    return (a * b).reshape(height, width).astype(np.float32)


def isotope_columns(columns) -> list:
    """Sorted ``i_<n>`` columns (alphadia/utils.py get_isotope_columns)."""
    found = []
    for c in columns:
        m = re.fullmatch(r"i_(\d+)", str(c))
        if m:
            found.append(int(m.group(1)))
    return [f"i_{i}" for i in sorted(found)]


def select_sharded(n_precursors: int, candidate_count: int, rank: int, world: int, select, gather_rows) -> dict:
    """Candidate selection of one run on ``world`` GPUs: precursors are independent (selection.py:620-660 hands
    them to threads one by one), so rank r selects for the range ``precursor_bounds(n, r, world)`` of the table
    sorted by precursor_idx and the candidate columns - ``candidate_count`` rows per precursor - are gathered
    once, in rank order = precursor order: every rank ends with the table one GPU would have produced.

    ``select(a, b)`` -> the candidate columns of precursors [a, b); ``gather_rows(local, rows_per_rank)`` ->
    concatenation in rank order (``runtime.Context.all_gather_rows``; the CPU tests pass the gloo transport)."""
    from alphadia_amd.distributed import precursor_bounds

    if world <= 1:
        return select(0, n_precursors)
    a, b = precursor_bounds(n_precursors, rank, world)
    local = select(a, b)
    rows = [(precursor_bounds(n_precursors, r, world)[1] - precursor_bounds(n_precursors, r, world)[0]) * candidate_count
            for r in range(world)]
    return {k: gather_rows(np.ascontiguousarray(v), rows) for k, v in local.items()}


class HipCandidateSelection:
    def __init__(self, dia_data, precursors_flat: pd.DataFrame, fragments_flat: pd.DataFrame,
                 config: CandidateSelectionConfig, rt_column: str, mobility_column: str,
                 precursor_mz_column: str, fragment_mz_column: str, fwhm_rt: float = 5.0,
                 fwhm_mobility: float = 0.012, device: int | None = None, rank: int | None = None,
                 world: int | None = None) -> None:
        self.dia_data = dia_data.to_jitclass() if hasattr(dia_data, "to_jitclass") else dia_data
        self.precursors_flat = precursors_flat.sort_values("precursor_idx").reset_index(drop=True)
        self.fragments_flat = fragments_flat
        self.config = config
        self.config_jit = config.to_jitclass()
        self.rt_column = rt_column
        self.mobility_column = mobility_column
        self.precursor_mz_column = precursor_mz_column
        self.fragment_mz_column = fragment_mz_column
        self.kernel = gaussian_kernel(self.dia_data, fwhm_rt, self.config_jit.sigma_scale_rt,
                                      self.config_jit.kernel_size, fwhm_mobility,
                                      self.config_jit.sigma_scale_mobility)
        self._device = device
        # several GPUs: every rank selects for its own contiguous range of the precursor table and the candidate
        # columns are gathered once (default: the communicator attached to the context)
        self.rank, self.world = rank, world

    def _pack_precursors(self, a: int = 0, b: int | None = None) -> _abi.Marshalled:
        df = self.precursors_flat if (a == 0 and b is None) else self.precursors_flat.iloc[a:b]
        iso = df[isotope_columns(df.columns)].values
        return _abi.pack_precursors(
            df["precursor_idx"].values, df["flat_frag_start_idx"].values, df["flat_frag_stop_idx"].values,
            df["charge"].values, df[self.rt_column].values, df[self.mobility_column].values,
            df[self.precursor_mz_column].values, iso,
        )

    def __call__(self, thread_count: int = 10, debug: bool = False) -> pd.DataFrame:
        from .scoring import fragment_columns

        ctx = runtime.get_context(self._device)
        ctx.stage_run(self.dia_data)
        if "cardinality" not in self.fragments_flat.columns:
            self.fragments_flat["cardinality"] = np.ones(len(self.fragments_flat), dtype=np.uint8)
        ctx.stage_fragments(*fragment_columns(self.fragments_flat, self.fragment_mz_column))
        rank, world = ctx.comm_info() if self.world is None else (int(self.rank or 0), int(self.world))
        if world > 1 and ctx.comm_info()[1] != world:
            raise runtime.HipBackendError(f"HipCandidateSelection(world={world}) needs a communicator of {world} ranks "
                                          "on the context (Context.comm_init)")
        arrays = select_sharded(len(self.precursors_flat), int(self.config_jit.candidate_count), rank, world,
                                lambda a, b: ctx.select_candidates(self._pack_precursors(a, b), self.config_jit, self.kernel),
                                ctx.all_gather_rows)
        keep = arrays["score"] > 0  # candidate_container_to_df (config_df.py:270-298)
        candidate_df = pd.DataFrame({c: arrays[c][keep] for c in CANDIDATE_COLUMNS})
        return candidate_df.merge(
            self.precursors_flat[["precursor_idx", "elution_group_idx", "decoy"]], on="precursor_idx", how="left"
        )
