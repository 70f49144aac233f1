"""Throughput of candidate selection on the ion-mobility bench run (see tools/bench_timstof.py).
Run on the GPU box from the repo root."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # synthetic data generators
import synthetic as syn
from alphadia_amd import _abi, runtime  # noqa: E402
from alphadia_amd.scoring import fragment_columns  # noqa: E402
from alphadia_amd.selection import CandidateSelectionConfig, gaussian_kernel  # noqa: E402

n_prec = int(os.environ.get("N_PREC", 20000))
case = syn.make_timstof_case(
    n_precursors=n_prec, n_cycles=int(os.environ.get("N_CYCLES", 300)), config_id=4, per_precursor=1, n_ms2_frames=8,
    windows_per_frame=3, scan_max_index=int(os.environ.get("SCAN_MAX", 256)), n_tof=int(os.environ.get("N_TOF", 200000)),
    events_per_push=float(os.environ.get("EVENTS_PER_PUSH", 25.0)), mz_lo=400.0, mz_hi=1000.0, frag_mz_lo=200.0,
    frag_mz_hi=1000.0, tof_mz_lo=195.0, tof_mz_hi=1010.0, planted_fraction=0.3,
)
case.dia.has_mobility = True
cfg = CandidateSelectionConfig()
cfg.update(dict(rt_tolerance=float(os.environ.get("RT_TOL", 15.0)), mobility_tolerance=0.1, candidate_count=3,
                peak_len_rt=3.0, sigma_scale_rt=0.5, peak_len_mobility=0.02))
kern = gaussian_kernel(case.dia, cfg.peak_len_rt, cfg.sigma_scale_rt, cfg.kernel_size, cfg.peak_len_mobility,
                       cfg.sigma_scale_mobility)
pdf = case.library.precursor_df.sort_values("precursor_idx").reset_index(drop=True)
iso = pdf[[c for c in pdf.columns if c.startswith("i_")]].values


def pack(df, iso_rows):
    return _abi.pack_precursors(df.precursor_idx.values, df.flat_frag_start_idx.values, df.flat_frag_stop_idx.values,
                                df.charge.values, df.rt_library.values, df.mobility_library.values,
                                df.mz_library.values, iso_rows)


pm = pack(pdf, iso)
ctx = runtime.get_context(0)
ctx.stage_run(case.dia)
cols = fragment_columns(case.library.fragment_df, "mz_library")
ctx.stage_fragments(*cols)
got = ctx.select_candidates(pm, cfg, kern)
t0 = time.perf_counter()
got = ctx.select_candidates(pm, cfg, kern)
wall = time.perf_counter() - t0
k_ms = ctx.select_time_ms()
found = got["score"] > 0
res = {
    "workload": f"timsTOF-style run ({case.dia.push_indices.size/1e6:.1f}M events, {int(case.dia.scan_max_index)} scans, {int(os.environ.get('N_CYCLES', 300))} cycles), {n_prec} "
                f"precursors, rt tolerance {cfg.rt_tolerance} s, mobility tolerance {cfg.mobility_tolerance}, kernel "
                f"{kern.shape[0]}x{kern.shape[1]}",
    "candidates_found": int(found.sum()),
    "tile_scans": int(np.median((got["scan_stop"] - got["scan_start"])[found])) if found.any() else 0,
    "kernel_ms": k_ms, "precursors_per_s_kernel": n_prec / (k_ms * 1e-3), "host_call_ms": wall * 1e3,
}
if not os.environ.get("ADH_BENCH_NO_CPU"):
    from oracle import oracle

    sample = min(n_prec, 1500)
    pm_s = pack(pdf.iloc[:sample], iso[:sample])
    th = min(64, os.cpu_count() or 1)
    t0 = time.perf_counter()
    exp = oracle.select_timstof(case.dia, cols, pm_s, cfg, kern, n_threads=th)
    dt = time.perf_counter() - t0
    n_rows = sample * cfg.candidate_count
    res["cpu_oracle"] = {"precursors_per_s": sample / dt, "threads": th, "sample": sample,
                         "boxes_identical_to_gpu": bool(all(np.array_equal(got[c][:n_rows], exp[c]) for c in got if c != "score"))}
print(json.dumps(res))
