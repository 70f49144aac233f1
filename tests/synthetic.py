"""Deterministic synthetic DIA runs, spectral libraries and candidate tables.

These generators produce exactly the arrays the reference hands across the
scoring boundary:

* a Thermo-style run in the ``AlphaRawJIT`` layout
  (reference ``alphadia/search/jitclasses/alpharaw_jit.py:78-138``): CSR peak
  lists per spectrum (``peak_start_idx_list`` / ``peak_stop_idx_list``), float32
  ``mz_values`` sorted ascending inside every spectrum and float32
  ``intensity_values``;
* a flat spectral library with the dtypes enforced by the reference schemas
  (``alphadia/validation/schemas.py:11-48``);
* a candidates table with the ``candidates_schema`` dtypes
  (``alphadia/validation/schemas.py:51-73``).

Nothing here is derived from reference code; it is our own data synthesis used
by the tests, the golden-vector generator and ``bench.py``.
"""

from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np
import pandas as pd

ISOTOPE_DELTA = 1.0033548350700006  # same constant as candidate.py:160
BASE_SEED = 20260928


@dataclass
class AlphaRawArrays:
    """Host arrays of a non-ion-mobility run (field names follow AlphaRawJIT)."""

    cycle: np.ndarray  # float64 (1, L, 1, 2); MS1 row is (-1, -1)
    rt_values: np.ndarray  # float32 [n_spec]
    peak_start_idx_list: np.ndarray  # int64 [n_spec]
    peak_stop_idx_list: np.ndarray  # int64 [n_spec]
    mz_values: np.ndarray  # float32 [n_peaks]
    intensity_values: np.ndarray  # float32 [n_peaks]
    mobility_values: np.ndarray = field(
        default_factory=lambda: np.array([1e-6, 0.0], dtype=np.float32)
    )
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
    def n_cycles(self) -> int:
        return self.n_spectra // self.cycle_len

    @property
    def frame_max_index(self) -> int:
        return self.n_spectra - 1

    @property
    def precursor_cycle_max_index(self) -> int:
        return self.n_spectra // self.cycle_len

    @property
    def max_mz_value(self) -> np.float32:
        return np.float32(self.mz_values.max()) if self.mz_values.size else np.float32(0)

    @property
    def min_mz_value(self) -> np.float32:
        return np.float32(self.mz_values.min()) if self.mz_values.size else np.float32(0)

    def nbytes(self) -> int:
        return int(
            self.mz_values.nbytes
            + self.intensity_values.nbytes
            + self.peak_start_idx_list.nbytes
            + self.peak_stop_idx_list.nbytes
            + self.rt_values.nbytes
        )


@dataclass
class SyntheticLibrary:
    precursor_df: pd.DataFrame
    fragment_df: pd.DataFrame


def make_cycle(n_ms2: int = 60, mz_lo: float = 400.0, mz_hi: float = 1000.0) -> np.ndarray:
    """DIA cycle: 1 MS1 row (-1,-1) + ``n_ms2`` contiguous isolation windows."""
    cycle = np.zeros((1, n_ms2 + 1, 1, 2), dtype=np.float64)
    cycle[0, 0, 0, :] = -1.0
    edges = np.linspace(mz_lo, mz_hi, n_ms2 + 1)
    cycle[0, 1:, 0, 0] = edges[:-1]
    cycle[0, 1:, 0, 1] = edges[1:]
    return cycle


def make_library(
    n_precursors: int,
    seed: int,
    k_fragments: int | tuple = 12,
    n_isotopes: int = 4,
    mz_lo: float = 400.0,
    mz_hi: float = 1000.0,
    rt_max: float = 600.0,
    few_fragment_fraction: float = 0.0,
    frag_mz_lo: float = 200.0,
    frag_mz_hi: float = 1800.0,
) -> SyntheticLibrary:
    """Flat library: 50 % decoys, target/decoy pairs share an elution group.

    ``few_fragment_fraction`` > 0 gives that share of precursors only 2-3
    fragments (exercises the "<=3 fragments" early exit, candidate.py:190).
    ``k_fragments`` = (lo, hi) draws the fragment count of every precursor uniformly from
    lo..hi (libraries with more than 12 / 16 fragments: transfer-library requantification scores
    with ``top_k_fragments = 9999``, transfer_library_requantification_handler.py:117-124).
    """
    rng = np.random.default_rng([seed, 1])
    n = int(n_precursors)
    precursor_idx = np.arange(n, dtype=np.uint32)
    elution_group_idx = (precursor_idx // 2).astype(np.uint32)
    decoy = (precursor_idx % 2).astype(np.uint8)
    charge = rng.choice(np.array([2, 3], dtype=np.uint8), size=n, p=[0.6, 0.4])
    mz = rng.uniform(mz_lo, mz_hi, n).astype(np.float32)
    rt = rng.uniform(0.0, rt_max, n).astype(np.float32)

    # averagine-like isotope envelope, normalised to 1
    lam = (mz.astype(np.float64) * charge) / 1800.0
    iso = np.empty((n, n_isotopes), dtype=np.float64)
    term = np.exp(-lam)
    for i in range(n_isotopes):
        iso[:, i] = term
        term = term * lam / (i + 1)
    iso *= rng.uniform(0.9, 1.1, iso.shape)
    iso /= iso.sum(axis=1, keepdims=True)
    iso = iso.astype(np.float32)

    if isinstance(k_fragments, (tuple, list)):
        n_frag = np.random.default_rng([seed, 11]).integers(int(k_fragments[0]), int(k_fragments[1]) + 1, n).astype(np.int64)
    else:
        n_frag = np.full(n, k_fragments, dtype=np.int64)
    if few_fragment_fraction > 0:
        few = rng.random(n) < few_fragment_fraction
        n_frag[few] = rng.integers(2, 4, few.sum())
    stop = np.cumsum(n_frag)
    start = stop - n_frag
    total = int(stop[-1]) if n else 0

    frag_mz = rng.uniform(frag_mz_lo, frag_mz_hi, total).astype(np.float32)
    # distinct intensities inside one precursor: a random permutation of a jittered grid
    base = rng.random(total)
    frag_int = (0.01 + 0.99 * base).astype(np.float32)
    within = np.arange(total, dtype=np.int64) - np.repeat(start, n_frag)
    frag_type = np.where(rng.random(total) < 0.5, 98, 121).astype(np.uint8)
    frag_number = (within + 1).astype(np.uint8)
    frag_position = within.astype(np.uint8)

    precursor_df = pd.DataFrame(
        {
            "elution_group_idx": elution_group_idx,
            "precursor_idx": precursor_idx,
            "channel": np.zeros(n, dtype=np.uint32),
            "decoy": decoy,
            "flat_frag_start_idx": start.astype(np.uint32),
            "flat_frag_stop_idx": stop.astype(np.uint32),
            "charge": charge,
            "rt_library": rt,
            "mobility_library": np.zeros(n, dtype=np.float32),
            "mz_library": mz,
            "proteins": np.full(n, "P", dtype=object),
            "genes": np.full(n, "G", dtype=object),
            "sequence": np.full(n, "PEPTIDEK", dtype=object),
            "mods": np.full(n, "", dtype=object),
            "mod_sites": np.full(n, "", dtype=object),
        }
    )
    for i in range(n_isotopes):
        precursor_df[f"i_{i}"] = iso[:, i]

    fragment_df = pd.DataFrame(
        {
            "mz_library": frag_mz,
            "intensity": frag_int,
            "cardinality": np.ones(total, dtype=np.uint8),
            "type": frag_type,
            "loss_type": np.zeros(total, dtype=np.uint8),
            "charge": np.ones(total, dtype=np.uint8),
            "number": frag_number,
            "position": frag_position,
        }
    )
    return SyntheticLibrary(precursor_df, fragment_df)


@dataclass
class PlantedPeaks:
    """Peaks to be merged into the noise of a run (global spectrum index)."""

    spec_idx: np.ndarray  # int64
    mz: np.ndarray  # float32
    intensity: np.ndarray  # float32
    apex_cycle: np.ndarray  # int64 [n_precursors], -1 when not planted


def plant_peptides(
    library: SyntheticLibrary,
    cycle: np.ndarray,
    n_cycles: int,
    seed: int,
    fraction: float = 0.3,
    sigma_cycles: float = 2.5,
    half_width: int = 8,
    n_isotopes: int = 3,
    apex: np.ndarray | None = None,
) -> PlantedPeaks:
    """Gaussian elution profiles for ``fraction`` of the target precursors (or for the precursors
    whose entry of ``apex`` is a cycle >= 0: label channels of one peptide elute together)."""
    rng = np.random.default_rng([seed, 2])
    pdf, fdf = library.precursor_df, library.fragment_df
    n = len(pdf)
    L = cycle.shape[1]
    if apex is not None:
        apex = np.asarray(apex, dtype=np.int64).copy()
        chosen = np.flatnonzero(apex >= 0)
    else:
        targets = np.flatnonzero(pdf["decoy"].values == 0)
        chosen = targets[rng.random(targets.size) < fraction]
        apex = np.full(n, -1, dtype=np.int64)
        lo, hi = 16, max(17, n_cycles - 16)
        apex[chosen] = rng.integers(lo, hi, chosen.size)

    win_lo = cycle[0, 1:, 0, 0]
    win_hi = cycle[0, 1:, 0, 1]

    spec_parts, mz_parts, int_parts = [], [], []
    offs = np.arange(-half_width, half_width + 1)
    gauss = np.exp(-0.5 * (offs / sigma_cycles) ** 2)

    p_mz = pdf["mz_library"].values.astype(np.float64)[chosen]
    p_ch = pdf["charge"].values.astype(np.float64)[chosen]
    p_apex = apex[chosen]
    cyc = p_apex[:, None] + offs[None, :]  # (P, W)
    ok = (cyc >= 0) & (cyc < n_cycles)

    # MS1 isotopes
    for i in range(n_isotopes):
        iso_int = pdf[f"i_{i}"].values.astype(np.float64)[chosen]
        iso_mz = p_mz + i * ISOTOPE_DELTA / p_ch
        ppm = rng.normal(2.0, 1.0, cyc.shape)
        mzv = iso_mz[:, None] * (1.0 + ppm * 1e-6)
        inten = 2e4 * iso_int[:, None] * gauss[None, :]
        spec = cyc * L
        spec_parts.append(spec[ok])
        mz_parts.append(mzv[ok])
        int_parts.append(inten[ok])

    # MS2 fragments in the window that contains the precursor m/z
    w = np.searchsorted(win_hi, p_mz, side="right")
    w = np.clip(w, 0, len(win_lo) - 1)
    fstart = pdf["flat_frag_start_idx"].values.astype(np.int64)[chosen]
    fstop = pdf["flat_frag_stop_idx"].values.astype(np.int64)[chosen]
    kmax = int((fstop - fstart).max()) if chosen.size else 0
    fmz_all = fdf["mz_library"].values
    fint_all = fdf["intensity"].values
    for k in range(kmax):
        has = (fstart + k) < fstop
        idx = np.where(has, fstart + k, fstart)
        fmz = fmz_all[idx].astype(np.float64)
        fin = fint_all[idx].astype(np.float64)
        ppm = rng.normal(2.0, 1.0, cyc.shape)
        mzv = fmz[:, None] * (1.0 + ppm * 1e-6)
        inten = 8e3 * fin[:, None] * gauss[None, :]
        spec = cyc * L + 1 + w[:, None]
        m = ok & has[:, None]
        spec_parts.append(spec[m])
        mz_parts.append(mzv[m])
        int_parts.append(inten[m])

    if spec_parts:
        spec_idx = np.concatenate(spec_parts).astype(np.int64)
        mz = np.concatenate(mz_parts).astype(np.float32)
        inten = np.concatenate(int_parts).astype(np.float32)
    else:
        spec_idx = np.zeros(0, np.int64)
        mz = np.zeros(0, np.float32)
        inten = np.zeros(0, np.float32)
    return PlantedPeaks(spec_idx, mz, inten, apex)


def _gen_chunk(args):
    (seed, chunk_id, c0, c1, L, ms1_peaks, ms2_peaks, p_spec, p_mz, p_int, r1, r2) = args
    rng = np.random.default_rng([seed, 3, chunk_id])
    n_cyc = c1 - c0
    n_spec = n_cyc * L
    per_spec = np.full(L, ms2_peaks, dtype=np.int64)
    per_spec[0] = ms1_peaks
    counts = np.tile(per_spec, n_cyc)
    n_noise = int(counts.sum())
    spec_local = np.repeat(np.arange(n_spec, dtype=np.int64), counts)
    is_ms1 = (spec_local % L) == 0
    u = rng.random(n_noise, dtype=np.float32)
    mz = np.where(is_ms1, r1[0] + (r1[1] - r1[0]) * u, r2[0] + (r2[1] - r2[0]) * u).astype(np.float32)
    inten = np.exp(rng.standard_normal(n_noise, dtype=np.float32) + np.float32(3.0)).astype(
        np.float32
    )
    if p_spec.size:
        spec_local = np.concatenate([spec_local, p_spec - c0 * L])
        mz = np.concatenate([mz, p_mz])
        inten = np.concatenate([inten, p_int])
    key = (spec_local.astype(np.uint64) << np.uint64(32)) | mz.view(np.uint32).astype(np.uint64)
    order = np.argsort(key, kind="stable")
    mz = mz[order]
    inten = inten[order]
    cnt = np.bincount(spec_local, minlength=n_spec).astype(np.int64)
    return mz, inten, cnt


def make_thermo_run(
    n_cycles: int,
    seed: int,
    cycle: np.ndarray | None = None,
    ms1_peaks: int = 4000,
    ms2_peaks: int = 1500,
    cycle_time: float = 1.5,
    planted: PlantedPeaks | None = None,
    chunk_cycles: int = 100,
    threads: int = 8,
    ms1_mz_range: tuple = (350.0, 1100.0),
    ms2_mz_range: tuple = (150.0, 1600.0),
) -> AlphaRawArrays:
    """Thermo-style run "T" of SURVEY.md section 8(d)."""
    if cycle is None:
        cycle = make_cycle()
    L = cycle.shape[1]
    n_spec = n_cycles * L
    rt = (np.arange(n_spec, dtype=np.float64) * (cycle_time / L)).astype(np.float32)

    if planted is not None and planted.spec_idx.size:
        order = np.argsort(planted.spec_idx, kind="stable")
        ps, pm, pi = planted.spec_idx[order], planted.mz[order], planted.intensity[order]
    else:
        ps = np.zeros(0, np.int64)
        pm = np.zeros(0, np.float32)
        pi = np.zeros(0, np.float32)

    jobs = []
    for cid, c0 in enumerate(range(0, n_cycles, chunk_cycles)):
        c1 = min(n_cycles, c0 + chunk_cycles)
        a = np.searchsorted(ps, c0 * L, side="left")
        b = np.searchsorted(ps, c1 * L, side="left")
        jobs.append(
            (seed, cid, c0, c1, L, ms1_peaks, ms2_peaks, ps[a:b], pm[a:b], pi[a:b],
             ms1_mz_range, ms2_mz_range)
        )

    if threads > 1 and len(jobs) > 1:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            parts = list(ex.map(_gen_chunk, jobs))
    else:
        parts = [_gen_chunk(j) for j in jobs]

    mz = np.concatenate([p[0] for p in parts])
    inten = np.concatenate([p[1] for p in parts])
    cnt = np.concatenate([p[2] for p in parts])
    stop = np.cumsum(cnt)
    start = stop - cnt
    return AlphaRawArrays(
        cycle=cycle,
        rt_values=rt,
        peak_start_idx_list=start.astype(np.int64),
        peak_stop_idx_list=stop.astype(np.int64),
        mz_values=mz,
        intensity_values=inten,
    )


def make_candidates(
    library: SyntheticLibrary,
    n_cycles: int,
    cycle_len: int,
    seed: int,
    per_precursor: int = 3,
    apex_cycle: np.ndarray | None = None,
    h_lo: int = 3,
    h_hi: int = 14,
    even_fraction: float = 0.0,
) -> pd.DataFrame:
    """Candidate boxes: rank 0 sits on the planted apex when there is one."""
    rng = np.random.default_rng([seed, 4])
    pdf = library.precursor_df
    n = len(pdf)
    C = per_precursor
    pidx = np.repeat(pdf["precursor_idx"].values.astype(np.uint32), C)
    eg = np.repeat(pdf["elution_group_idx"].values.astype(np.uint32), C)
    rank = np.tile(np.arange(C, dtype=np.uint8), n)
    h = rng.integers(h_lo, h_hi + 1, n * C)
    c = rng.integers(0, n_cycles, n * C)
    if apex_cycle is not None:
        ap = np.repeat(apex_cycle, C)
        use = (ap >= 0) & (rank == 0)
        c = np.where(use, ap, c)
    c = np.clip(c, h, n_cycles - h - 1)
    frame_center = c * cycle_len
    frame_start = (c - h) * cycle_len
    frame_stop = (c + h + 1) * cycle_len
    if even_fraction > 0:
        ev = rng.random(n * C) < even_fraction
        frame_stop = np.where(ev, frame_stop - cycle_len, frame_stop)
    df = pd.DataFrame(
        {
            "elution_group_idx": eg,
            "precursor_idx": pidx,
            "rank": rank,
            "scan_start": np.zeros(n * C, dtype=np.int64),
            "scan_stop": np.ones(n * C, dtype=np.int64),
            "scan_center": np.zeros(n * C, dtype=np.int64),
            "frame_start": frame_start.astype(np.int64),
            "frame_stop": frame_stop.astype(np.int64),
            "frame_center": frame_center.astype(np.int64),
            "score": rng.uniform(0, 100, n * C).astype(np.float32),
        }
    )
    return df


@dataclass
class SyntheticCase:
    dia: AlphaRawArrays
    library: SyntheticLibrary
    candidates_df: pd.DataFrame
    apex_cycle: np.ndarray


def make_case(
    n_precursors: int,
    n_cycles: int,
    config_id: int = 1,
    per_precursor: int = 3,
    ms1_peaks: int = 4000,
    ms2_peaks: int = 1500,
    n_ms2: int = 60,
    planted_fraction: float = 0.3,
    few_fragment_fraction: float = 0.0,
    even_fraction: float = 0.0,
    threads: int = 8,
    seed: int | None = None,
    mz_lo: float = 400.0,
    mz_hi: float = 1000.0,
    frag_mz_lo: float = 200.0,
    frag_mz_hi: float = 1800.0,
    ms1_mz_range: tuple = (350.0, 1100.0),
    ms2_mz_range: tuple = (150.0, 1600.0),
    k_fragments: int | tuple = 12,
    run: bool = True,
) -> SyntheticCase:
    """One full synthetic workload (run + library + candidates).  ``run=False`` leaves the run out
    (``dia`` is None): library, planted apexes and candidates are the same as with it."""
    seed = BASE_SEED + config_id if seed is None else seed
    cycle = make_cycle(n_ms2=n_ms2, mz_lo=mz_lo, mz_hi=mz_hi)
    lib = make_library(
        n_precursors,
        seed,
        mz_lo=mz_lo,
        mz_hi=mz_hi,
        rt_max=n_cycles * 1.5,
        few_fragment_fraction=few_fragment_fraction,
        frag_mz_lo=frag_mz_lo,
        frag_mz_hi=frag_mz_hi,
        k_fragments=k_fragments,
    )
    planted = plant_peptides(lib, cycle, n_cycles, seed, fraction=planted_fraction)
    dia = make_thermo_run(
        n_cycles,
        seed,
        cycle=cycle,
        ms1_peaks=ms1_peaks,
        ms2_peaks=ms2_peaks,
        planted=planted,
        threads=threads,
        ms1_mz_range=ms1_mz_range,
        ms2_mz_range=ms2_mz_range,
    ) if run else None
    cands = make_candidates(
        lib,
        n_cycles,
        cycle.shape[1],
        seed,
        per_precursor=per_precursor,
        apex_cycle=planted.apex_cycle,
        even_fraction=even_fraction,
    )
    return SyntheticCase(dia, lib, cands, planted.apex_cycle)


def multiplex_library(
    library: SyntheticLibrary,
    channels: tuple = (0, 4, 8, 12),
    label_mass: float = 1.0018,
    seed: int = 0,
) -> SyntheticLibrary:
    """Channel-multiplexed copy of a library.

    Every precursor of ``library`` becomes one precursor per channel in the same elution
    group: precursor m/z shifted by ``channel * (1 + n_K) / charge * label_mass``, y-ions
    (type 121) shifted by ``channel * label_mass``, b-ions (type 98) shared by all channels
    and therefore flagged with ``cardinality = len(channels)``.
    """
    rng = np.random.default_rng([seed, 7])
    pdf, fdf = library.precursor_df, library.fragment_df
    n, C = len(pdf), len(channels)
    n_k = rng.integers(0, 2, n)
    start = pdf["flat_frag_start_idx"].values.astype(np.int64)
    stop = pdf["flat_frag_stop_idx"].values.astype(np.int64)
    nf = stop - start
    prec_parts, frag_parts = [], []
    offset = 0
    total = int(nf.sum())
    owner = np.repeat(np.arange(n), nf)
    for ci, ch in enumerate(channels):
        pp = pdf.copy()
        pp["channel"] = np.uint32(ch)
        pp["precursor_idx"] = (pdf["precursor_idx"].values.astype(np.int64) * C + ci).astype(np.uint32)
        pp["mz_library"] = (
            pdf["mz_library"].values.astype(np.float64)
            + ch * (1 + n_k) / pdf["charge"].values.astype(np.float64) * label_mass
        ).astype(np.float32)
        new_stop = np.cumsum(nf) + offset
        pp["flat_frag_start_idx"] = (new_stop - nf).astype(np.uint32)
        pp["flat_frag_stop_idx"] = new_stop.astype(np.uint32)
        ff = fdf.iloc[np.concatenate([np.arange(a, b) for a, b in zip(start, stop)])].copy() if total else fdf.copy()
        is_y = ff["type"].values == 121
        mz = ff["mz_library"].values.astype(np.float64)
        mz = np.where(is_y, mz + ch * label_mass, mz)
        ff["mz_library"] = mz.astype(np.float32)
        ff["cardinality"] = np.where(is_y, 1, C).astype(np.uint8)
        prec_parts.append(pp)
        frag_parts.append(ff)
        offset += total
    del owner
    precursor_df = pd.concat(prec_parts, ignore_index=True)
    # interleave: all channels of an elution group next to each other, as a real library has them
    order = np.argsort(precursor_df["precursor_idx"].values, kind="stable")
    precursor_df = precursor_df.iloc[order].reset_index(drop=True)
    fragment_df = pd.concat(frag_parts, ignore_index=True)
    return SyntheticLibrary(precursor_df, fragment_df)


@dataclass
class MultiplexCase:
    dia: AlphaRawArrays
    library: SyntheticLibrary      # every elution group in all label channels
    psm_df: pd.DataFrame           # the identifications multiplex requantification starts from
    channels: tuple
    apex_cycle: np.ndarray         # per library precursor, -1 = nothing planted


def make_multiplex_case(n_groups: int, n_cycles: int, config_id: int = 5, channels: tuple = (0, 4, 8, 12),
                        planted_fraction: float = 0.3, threads: int = 8, **run_kwargs) -> MultiplexCase:
    """BASELINE configs[4] (SURVEY.md section 8d "Multiplex"): ``n_groups`` elution groups in the label
    channels ``channels`` (y-ions shifted per channel, b-ions shared with ``cardinality = len(channels)``),
    the channels of a planted peptide eluting together, and one identification per elution group (its
    reference-channel precursor, the box of rank 0) - the table
    ``MultiplexingRequantificationHandler`` hands to ``multiplex_candidates``
    (multiplexing_requantification_handler.py:75-98)."""
    seed = BASE_SEED + config_id
    cycle = make_cycle()
    base = make_library(n_groups, seed, rt_max=n_cycles * 1.5)
    bp = base.precursor_df
    bp["decoy"] = np.uint8(0)                       # requantification starts from targets (decoys: a channel)
    bp["elution_group_idx"] = bp["precursor_idx"].values.astype(np.uint32)
    lib = multiplex_library(base, channels=channels, seed=seed)
    C = len(channels)
    rng = np.random.default_rng([seed, 5])
    group_apex = np.where(rng.random(n_groups) < planted_fraction, rng.integers(16, max(17, n_cycles - 16), n_groups), -1)
    apex = group_apex[lib.precursor_df["elution_group_idx"].values.astype(np.int64)]
    planted = plant_peptides(lib, cycle, n_cycles, seed, apex=apex)
    dia = make_thermo_run(n_cycles, seed, cycle=cycle, planted=planted, threads=threads, **run_kwargs)
    # the identification of every group: its channel-0 precursor with the rank-0 box
    ref = SyntheticLibrary(lib.precursor_df[lib.precursor_df["channel"].values == channels[0]].reset_index(drop=True),
                           lib.fragment_df)
    cands = make_candidates(ref, n_cycles, cycle.shape[1], seed, per_precursor=1, apex_cycle=planted.apex_cycle[::C])
    cands["proba"] = rng.uniform(0.0, 0.01, len(cands)).astype(np.float32)
    return MultiplexCase(dia, lib, cands, tuple(channels), planted.apex_cycle)


# --------------------------------------------------------------------------- timsTOF-style run
@dataclass
class TimsTOFArrays:
    """Host arrays of an ion-mobility run in the transposed (TOF-major) layout.

    Field names follow ``TimsTOFTransposeJIT``
    (alphadia/search/jitclasses/bruker_jit.py:22-137): for every TOF index the detector
    events are listed as (push index, intensity) with push = frame * scan_max_index + scan,
    ascending inside a TOF bin.
    """

    cycle: np.ndarray  # float64 (1, L, scan_max_index, 2); (-1, -1) = unfragmented
    dia_precursor_cycle: np.ndarray  # int64 [L * scan_max_index]: cycle row of every push
    rt_values: np.ndarray  # float64 [n_frames]
    mobility_values: np.ndarray  # float64 [scan_max_index], descending
    mz_values: np.ndarray  # float64 [n_tof], ascending
    tof_indptr: np.ndarray  # int64 [n_tof + 1]
    push_indices: np.ndarray  # uint32 [n_events]
    intensity_values: np.ndarray  # uint16 [n_events]
    scan_max_index: int
    zeroth_frame: bool = True
    has_mobility: bool = True

    @property
    def cycle_len(self) -> int:
        return int(self.cycle.shape[1])

    @property
    def frame_max_index(self) -> int:
        return int(self.rt_values.shape[0]) - 1

    @property
    def n_cycles(self) -> int:
        return (int(self.rt_values.shape[0]) - int(self.zeroth_frame)) // self.cycle_len

    @property
    def dia_mz_cycle(self) -> np.ndarray:
        return self.cycle.reshape(-1, 2)


def make_timstof_cycle(n_ms2_frames: int, windows_per_frame: int, scan_max_index: int,
                       mz_lo: float, mz_hi: float, uncovered_scans: int = 2) -> np.ndarray:
    """diaPASEF-like cycle: frame 0 = MS1 (all scans -1), every MS2 frame carries
    ``windows_per_frame`` isolation windows stacked over scan ranges."""
    L = n_ms2_frames + 1
    cycle = np.full((1, L, scan_max_index, 2), -1.0, dtype=np.float64)
    n_win = n_ms2_frames * windows_per_frame
    edges = np.linspace(mz_lo, mz_hi, n_win + 1)
    usable = scan_max_index - uncovered_scans
    bounds = np.linspace(0, usable, windows_per_frame + 1).astype(int)
    w = 0
    for j in range(windows_per_frame):
        for fr in range(1, L):
            # high m/z at low scan numbers (high mobility), like real diaPASEF schemes
            cycle[0, fr, bounds[j] : bounds[j + 1], 0] = edges[n_win - 1 - w]
            cycle[0, fr, bounds[j] : bounds[j + 1], 1] = edges[n_win - w]
            w += 1
    return cycle


@dataclass
class TimsTOFCase:
    dia: TimsTOFArrays
    library: SyntheticLibrary
    candidates_df: pd.DataFrame


def _sorted_noise(seed: int, n_tof: int, n_push: int, S: int, events_per_push: float, threads: int):
    """Uniform (TOF, push) noise events in (TOF, push) order without a sort: the TOF axis is cut into ranges,
    every range draws its Poisson count and the order statistics of that many uniform keys over
    range x pushes (normalised cumulative exponential gaps).  -> (push uint32, tof int32, intensity uint16)."""
    from concurrent.futures import ThreadPoolExecutor

    width = n_push - S  # (the S pushes of the empty zeroth frame carry nothing)
    n_parts = max(1, min(n_tof, 4 * max(threads, 1), int(events_per_push * width // 2_000_000) + 1))
    cuts = np.linspace(0, n_tof, n_parts + 1).astype(np.int64)

    def part(i):
        t0, t1 = int(cuts[i]), int(cuts[i + 1])
        r = np.random.default_rng([seed, 11, i])
        space = float(t1 - t0) * float(width)
        n = int(r.poisson(events_per_push * width * (t1 - t0) / n_tof))
        g = r.standard_exponential(n + 1)
        np.cumsum(g, out=g)
        keys = np.minimum((g[:-1] * (space / g[-1])).astype(np.int64), int(space) - 1)
        tof = (keys // width + t0).astype(np.int32)
        push = (keys % width + S).astype(np.uint32)
        inten = np.clip(r.lognormal(3.0, 1.0, n), 1, 60000).astype(np.uint16)
        return push, tof, inten

    with ThreadPoolExecutor(max_workers=max(threads, 1)) as ex:
        parts = list(ex.map(part, range(n_parts)))
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]))


def make_timstof_case(
    n_precursors: int = 200,
    n_cycles: int = 40,
    config_id: int = 4,
    per_precursor: int = 2,
    n_ms2_frames: int = 4,
    windows_per_frame: int = 2,
    scan_max_index: int = 64,
    n_tof: int = 60000,
    events_per_push: float = 40.0,
    tof_mz_lo: float = 195.0,
    tof_mz_hi: float = 490.0,
    mz_lo: float = 400.0,
    mz_hi: float = 480.0,
    frag_mz_lo: float = 200.0,
    frag_mz_hi: float = 350.0,
    planted_fraction: float = 0.5,
    seed: int | None = None,
    h_range: tuple = (2, 10),
    hs_range: tuple = (3, 12),
    candidates_on_window: bool = False,
    sorted_noise: bool = False,
    threads: int = 16,
) -> TimsTOFCase:
    """Run "B" of SURVEY.md section 8(d) at a configurable (test) scale.

    ``sorted_noise``: draw the noise events already in (TOF, push) order - order statistics of uniform keys
    as cumulative sums of exponential gaps, TOF range by TOF range on ``threads`` threads - instead of
    drawing and lexsorting them: the full-size run (5e8 events) in seconds instead of minutes.  Another
    random stream, hence another run: the committed fixtures use the default."""
    seed = BASE_SEED + config_id if seed is None else seed
    rng = np.random.default_rng([seed, 7])
    cycle = make_timstof_cycle(n_ms2_frames, windows_per_frame, scan_max_index, mz_lo, mz_hi)
    L = cycle.shape[1]
    S = scan_max_index
    n_frames = n_cycles * L + 1  # frame 0 is alphatims' empty zeroth frame
    rt = np.arange(n_frames, dtype=np.float64) * 0.1
    mobility = np.linspace(1.6, 0.6, S).astype(np.float64)
    # quadratic TOF -> m/z, spanning fragments and precursors
    t = np.arange(n_tof, dtype=np.float64)
    mz_table = (np.sqrt(tof_mz_lo) + t * (np.sqrt(tof_mz_hi) - np.sqrt(tof_mz_lo)) / (n_tof - 1)) ** 2

    lib = make_library(n_precursors, seed, mz_lo=mz_lo, mz_hi=mz_hi, rt_max=float(rt[-1]),
                       frag_mz_lo=frag_mz_lo, frag_mz_hi=frag_mz_hi)
    pdf, fdf = lib.precursor_df, lib.fragment_df
    pdf["mobility_library"] = rng.uniform(0.7, 1.5, n_precursors).astype(np.float32)

    # ---- noise events
    n_push = n_frames * S
    if sorted_noise:
        ev_push, ev_tof, ev_int = _sorted_noise(seed, n_tof, n_push, S, events_per_push, threads)
    else:
        n_noise = rng.poisson(events_per_push * (n_push - S))
        ev_push = rng.integers(S, n_push, n_noise).astype(np.int64)
        ev_tof = rng.integers(0, n_tof, n_noise).astype(np.int64)
        ev_int = np.clip(rng.lognormal(3.0, 1.0, n_noise), 1, 60000).astype(np.int64)

    # ---- planted peptides: Gaussian in cycle and in scan (vectorised over precursors x cells)
    targets = np.flatnonzero(pdf["decoy"].values == 0)
    chosen = targets[rng.random(targets.size) < planted_fraction]
    apex_cycle = np.full(n_precursors, -1, dtype=np.int64)
    apex_scan = np.full(n_precursors, -1, dtype=np.int64)
    p_mz_all = pdf["mz_library"].values.astype(np.float64)
    # a scan at which some MS2 frame isolates the precursor: the home scan of every precursor
    home_scan = np.full(n_precursors, -1, dtype=np.int64)
    todo = np.arange(n_precursors)
    for _ in range(64):  # rejection sampling over scans; windows cover contiguous scan ranges
        if todo.size == 0:
            break
        sc_try = rng.integers(0, S, todo.size)
        lo = cycle[0, 1:, sc_try, 0]  # (todo, L-1)
        hi = cycle[0, 1:, sc_try, 1]
        ok = ((lo <= p_mz_all[todo, None]) & (hi > p_mz_all[todo, None])).any(axis=1)
        home_scan[todo[ok]] = sc_try[ok]
        todo = todo[~ok]
    chosen = chosen[home_scan[chosen] >= 0]
    apex_cycle[chosen] = rng.integers(8, max(9, n_cycles - 8), chosen.size)
    apex_scan[chosen] = home_scan[chosen]
    if chosen.size:
        dc = np.arange(-6, 7)
        ds = np.arange(-5, 6)
        g_cell = (np.exp(-0.5 * (dc / 2.0) ** 2)[:, None] * np.exp(-0.5 * (ds / 1.8) ** 2)[None, :]).reshape(-1)
        dc_cell = np.repeat(dc, ds.size)
        ds_cell = np.tile(ds, dc.size)
        cell_p = np.repeat(chosen, g_cell.size)
        cell_c = apex_cycle[cell_p] + np.tile(dc_cell, chosen.size)
        cell_s = apex_scan[cell_p] + np.tile(ds_cell, chosen.size)
        cell_g = np.tile(g_cell, chosen.size)
        inside = (cell_c >= 0) & (cell_c < n_cycles) & (cell_s >= 0) & (cell_s < S)
        cell_p, cell_c, cell_s, cell_g = cell_p[inside], cell_c[inside], cell_s[inside], cell_g[inside]
        cell_mz = p_mz_all[cell_p]
        charge = pdf["charge"].values.astype(np.float64)[cell_p]
        parts = []

        def emit(push, mz_true, amp):
            keep = amp >= 1
            if keep.any():
                mzv = mz_true[keep] * (1 + rng.normal(2e-6, 1e-6, int(keep.sum())))
                parts.append((push[keep], np.clip(np.searchsorted(mz_table, mzv), 0, n_tof - 1),
                              np.clip(amp[keep].astype(np.int64), 1, 60000)))

        push1 = (cell_c * L + 1) * S + cell_s  # the MS1 frame of the cycle (frame 0 is the empty zeroth frame)
        for i in range(3):
            emit(push1, cell_mz + i * ISOTOPE_DELTA / charge, 3000.0 * pdf[f"i_{i}"].values.astype(np.float64)[cell_p] * cell_g)
        f_start = pdf["flat_frag_start_idx"].values.astype(np.int64)
        f_stop = pdf["flat_frag_stop_idx"].values.astype(np.int64)
        f_mz = fdf["mz_library"].values.astype(np.float64)
        f_int = fdf["intensity"].values.astype(np.float64)
        k_max = int((f_stop - f_start)[chosen].max())
        for fr in range(1, L):  # fragments in every MS2 frame row of this scan that isolates the precursor
            sel = (cycle[0, fr, cell_s, 0] <= cell_mz) & (cell_mz < cycle[0, fr, cell_s, 1])
            if not sel.any():
                continue
            sp, sg = cell_p[sel], cell_g[sel]
            push2 = (cell_c[sel] * L + 1 + fr) * S + cell_s[sel]
            for k in range(k_max):
                has = f_start[sp] + k < f_stop[sp]
                idx = (f_start[sp] + k)[has]
                emit(push2[has], f_mz[idx], 1500.0 * f_int[idx] * sg[has])
        if parts and sorted_noise:
            # the planted events (a few million) are sorted and merged into the sorted noise
            pp = np.concatenate([x[0] for x in parts]).astype(np.int64)
            pt = np.concatenate([x[1] for x in parts]).astype(np.int64)
            pi = np.concatenate([x[2] for x in parts]).astype(np.uint16)
            o = np.lexsort((pp, pt))
            pp, pt, pi = pp[o], pt[o], pi[o]
            at = np.searchsorted(ev_tof.astype(np.int64) * n_push + ev_push, pt * n_push + pp, side="right")
            ev_push = np.insert(ev_push, at, pp.astype(ev_push.dtype))
            ev_tof = np.insert(ev_tof, at, pt.astype(ev_tof.dtype))
            ev_int = np.insert(ev_int, at, pi)
        elif parts:
            ev_push = np.concatenate([ev_push] + [x[0] for x in parts])
            ev_tof = np.concatenate([ev_tof] + [x[1] for x in parts])
            ev_int = np.concatenate([ev_int] + [x[2] for x in parts])
    if not sorted_noise:
        order = np.lexsort((ev_push, ev_tof))
        ev_push, ev_tof, ev_int = ev_push[order], ev_tof[order], ev_int[order]
    tof_indptr = np.concatenate([[0], np.cumsum(np.bincount(ev_tof, minlength=n_tof))]).astype(np.int64)

    dia = TimsTOFArrays(
        cycle=cycle,
        dia_precursor_cycle=np.repeat(np.arange(L, dtype=np.int64), S),
        rt_values=rt,
        mobility_values=mobility,
        mz_values=mz_table,
        tof_indptr=tof_indptr,
        push_indices=ev_push.astype(np.uint32, copy=False),
        intensity_values=ev_int.astype(np.uint16, copy=False),
        scan_max_index=S,
    )

    # ---- candidates: rank 0 on the planted apex; boxes of 2h+1 cycles x 2hs scans (defaults 5-21 x 6-24).
    # ``candidates_on_window``: every box is centred on a scan at which the precursor is isolated (what
    # candidate selection delivers); otherwise the scan centre of the unplanted boxes is arbitrary and
    # most of them see no isolation window at all (early exit)
    C = per_precursor
    n = n_precursors * C
    pidx = np.repeat(pdf["precursor_idx"].values.astype(np.uint32), C)
    rank = np.tile(np.arange(C, dtype=np.uint8), n_precursors)
    h = rng.integers(h_range[0], h_range[1] + 1, n)
    cc = rng.integers(0, n_cycles, n)
    ap = np.repeat(apex_cycle, C)
    use = (ap >= 0) & (rank == 0)
    cc = np.where(use, ap, cc)
    cc = np.clip(cc, h, n_cycles - h - 1)
    even = rng.random(n) < 0.3
    c_stop = cc + h + 1 - even.astype(np.int64)
    hs = rng.integers(hs_range[0], hs_range[1] + 1, n)
    sc = rng.integers(0, S, n)
    sc = np.where(use, np.repeat(np.maximum(apex_scan, 0), C), sc)
    if candidates_on_window:
        home = np.repeat(home_scan, C)
        sc = np.where((home >= 0) & ~use, np.clip(home + rng.integers(-2, 3, n), 0, S - 1), sc)
    sc = np.clip(sc, hs, S - hs - 1)
    cands = pd.DataFrame(
        {
            "elution_group_idx": np.repeat(pdf["elution_group_idx"].values.astype(np.uint32), C),
            "precursor_idx": pidx,
            "rank": rank,
            "scan_start": (sc - hs).astype(np.int64),
            "scan_stop": (sc + hs).astype(np.int64),
            "scan_center": sc.astype(np.int64),
            "frame_start": ((cc - h) * L + 1).astype(np.int64),
            "frame_stop": (c_stop * L + 1).astype(np.int64),
            "frame_center": (cc * L + 1).astype(np.int64),
            "score": rng.uniform(0, 100, n).astype(np.float32),
        }
    )
    return TimsTOFCase(dia, lib, cands)


def make_competition_table(n_psm: int, seed: int = 0, n_windows: int = 60, run_seconds: float = 7200.0, k: int = 12,
                           shared_fraction: float = 0.1) -> dict:
    """The table fragment competition receives inside FDR (alphadia/fdr/fdr.py:146-163, prepared by
    FragmentCompetition.__call__, fragcomp.py:268-289): PSMs below the heuristic FDR, sorted by
    (DIA window, proba), every PSM with `k` observed fragment masses.

    RT uniform over the run, windows equally filled.  Random fragment lists never share three masses, so
    a `shared_fraction` of the PSMs are made "the same signal seen twice" - the case the step exists
    for: such a PSM takes 3...k of its fragment masses (ppm-level jitter) and its RT (within a second)
    from another PSM of its window; half of those donors are themselves copies, which gives the greedy
    rule chains to resolve.
    """
    rng = np.random.default_rng(BASE_SEED + 7000 + seed)
    per = n_psm // n_windows
    sizes = np.full(n_windows, per, dtype=np.int64)
    sizes[: n_psm - per * n_windows] += 1
    window_stop = np.cumsum(sizes)
    window_start = window_stop - sizes
    rt = rng.uniform(0.0, run_seconds, n_psm).astype(np.float32)
    mz = rng.uniform(200.0, 1800.0, (n_psm, k)).astype(np.float32)
    window = np.repeat(np.arange(n_windows), sizes)
    n_copy = int(n_psm * shared_fraction)
    copies = rng.choice(n_psm, n_copy, replace=False)
    # donor: another row of the same window (position drawn inside the window)
    donors = window_start[window[copies]] + (rng.random(n_copy) * sizes[window[copies]]).astype(np.int64)
    donors = np.where(donors == copies, window_start[window[copies]] + (donors - window_start[window[copies]] + 1)
                      % sizes[window[copies]], donors)
    order = np.argsort(rng.random(n_copy))  # copies of copies: apply in a random order, one after another group-wise
    copies, donors = copies[order], donors[order]
    n_shared = rng.integers(3, k + 1, n_copy)
    for lo in range(0, n_copy, 4096):  # small groups so that later copies see earlier ones as donors
        c, d, m = copies[lo:lo + 4096], donors[lo:lo + 4096], n_shared[lo:lo + 4096]
        take = np.arange(k)[None, :] < m[:, None]
        jitter = 1.0 + rng.normal(0.0, 2e-6, (len(c), k))
        mz[c] = np.where(take, (mz[d] * jitter).astype(np.float32), mz[c])
        rt[c] = rt[d] + rng.uniform(-1.0, 1.0, len(c)).astype(np.float32)
    frag_start = np.arange(n_psm, dtype=np.int64) * k
    return dict(window_start=window_start, window_stop=window_stop, rt=rt, frag_start=frag_start,
                frag_stop=frag_start + k, mz=mz.reshape(-1), n_windows=n_windows, k=k)
