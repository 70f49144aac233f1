"""BASELINE configs[3]: the ion-mobility (timsTOF-style) path at the specified shape.

    N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 python tools/bench_timstof.py

Run "B" of SURVEY.md section 8(d): 1 MS1 + 8 diaPASEF frames per cycle, candidates of S in [17, 39]
scans x F in [7, 29] cycles centred where the precursor is isolated (what candidate selection
delivers), 30 % of the target precursors planted.  Reports, like bench.py, the host -> host step
(adh_score_candidates), the resident step, the kernel roofline and the CPU oracle on the same
inputs.  No torch.  Defaults are a reduced run that finishes in a minute.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # synthetic data generators
import synthetic as syn
from alphadia_amd import runtime  # noqa: E402
from alphadia_amd.distributed import slice_soa  # noqa: E402
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled  # noqa: E402
from bench import cpu_quota_cores  # noqa: E402

n_prec = int(os.environ.get("N_PREC", 20000))
n_cycles = int(os.environ.get("N_CYCLES", 300))
t0 = time.time()
case = syn.make_timstof_case(
    n_precursors=n_prec, n_cycles=n_cycles, config_id=4, per_precursor=3, n_ms2_frames=8, windows_per_frame=3,
    scan_max_index=int(os.environ.get("SCAN_MAX", 256)), n_tof=int(os.environ.get("N_TOF", 200000)),
    events_per_push=float(os.environ.get("EVENTS_PER_PUSH", 25.0)), mz_lo=400.0, mz_hi=1000.0, frag_mz_lo=200.0,
    frag_mz_hi=1000.0, tof_mz_lo=195.0, tof_mz_hi=1010.0, planted_fraction=float(os.environ.get("PLANTED", 0.3)),
    h_range=(3, 14), hs_range=(9, 19), candidates_on_window=True,
    sorted_noise=bool(int(os.environ.get("SORTED_NOISE", "1"))), threads=min(32, os.cpu_count() or 8),
)
dia = case.dia
print(f"generated {dia.push_indices.size/1e6:.1f}M events in {time.time()-t0:.1f}s", file=sys.stderr, flush=True)
cfg = CandidateScoringConfig()
cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True, experimental_xic=True))
cfgj = cfg.to_jitclass()
ctx = runtime.get_context(0)
soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library", pool=ctx.pinned)
n = len(soa["precursor_idx"])
t0 = time.time()
ctx.stage_run(dia)
cols = fragment_columns(case.library.fragment_df, "mz_library")
ctx.stage_fragments(*cols)
t_stage = time.time() - t0
packed = pack_assembled(soa)

# ---- host -> host steps
steps, warm = int(os.environ.get("STEPS", 5)), 4
for _ in range(warm):
    host = ctx.score_host(packed, cfgj, reuse_buffers=True)
ctx.kernel_time_ms(reset=True)
t0 = time.perf_counter()
for _ in range(steps):
    host = ctx.score_host(packed, cfgj, reuse_buffers=True)
h2h_ms = (time.perf_counter() - t0) / steps * 1e3
g_ms, f_ms, launches = ctx.kernel_time_ms(reset=True)
g_ms, f_ms = g_ms * launches / steps, f_ms * launches / steps
valid = host["valid"].astype(bool).copy()
matched = ctx.device_tables_to_host(names=["stat_matched_peaks"])["stat_matched_peaks"][:n]

# ---- resident steps (table + plan in HBM, tables stay in HBM)
ctx.upload_candidates(packed)
view, st = ctx.device_tables(), ctx.stream_handle()
for it in range(steps + 1):
    if it == 1:
        ctx.synchronize()
        t0 = time.perf_counter()
    ctx.zero_device_tables(st)
    ctx.score_uploaded(cfgj, view, st)
ctx.synchronize()
res_ms = (time.perf_counter() - t0) / steps * 1e3

# ---- algorithmic bytes, SURVEY.md section 8(d) with the run-B probe term
L, S_max = dia.cycle_len, int(dia.scan_max_index)
mzv = dia.mz_values
indptr = dia.tof_indptr
lib_start, lib_stop = soa["frag_start_idx"].astype(np.int64), soa["frag_stop_idx"].astype(np.int64)
K = np.minimum(lib_stop - lib_start, 12)
fmz = case.library.fragment_df["mz_library"].values.astype(np.float64)


def window_bytes(mz, tol_ppm):
    lo = np.searchsorted(mzv, mz * (1 - tol_ppm * 1e-6))
    hi = np.searchsorted(mzv, mz * (1 + tol_ppm * 1e-6))
    return 8 * (hi - lo) + 6 * (indptr[hi] - indptr[lo])


per = np.zeros(n, dtype=np.int64)
for k in range(int((lib_stop - lib_start).max())):
    has = lib_start + k < lib_stop
    per[has] += window_bytes(fmz[(lib_start + k)[has]], 15.0)
for i in range(3):
    per += window_bytes(soa["precursor_mz"].astype(np.float64) + i * 1.0033548350700006 / soa["charge"], 10.0)
F = (soa["frame_stop"] - 1) // L - (soa["frame_start"] - 1) // L
S = soa["scan_stop"] - soa["scan_start"]
per += 4 * (F * S * 2)  # the two push queries (fragment frames, MS1 frame), one push per (cycle, scan) each
per += 18 * (lib_stop - lib_start) + 64 + (46 * 4 + K * 38 + 6)
alg_bytes = float(per.sum())
kernel_ms = g_ms + f_ms

# ---- second yardstick: the bytes the candidates' (TOF bin, cycle range) ranges actually hold.  The run-B
# formula above charges whole TOF bins (the reference merge-joins a bin from its first event), ~100x what
# these kernels read through their indices, so its "fraction" says nothing about them.  Here, per candidate:
# events of every (window, TOF bin) inside the candidate's frames x 6 B (push + intensity), two index words +
# the bin's tof_indptr and m/z per (window, bin), two m/z look-ups per window, the library slice, the plan
# record and the 646-byte output row.  Exact on a sample of candidates, scaled to all of them.
def touched_bytes(sample_rows):
    push = dia.push_indices
    out = np.zeros(len(sample_rows), dtype=np.int64)
    for j, i in enumerate(sample_rows):
        p_lo = np.uint32(int(soa["frame_start"][i]) * S_max)
        p_hi = np.uint32(int(soa["frame_stop"][i]) * S_max)
        mzs = [fmz[a] for a in range(lib_start[i], lib_stop[i])]
        tol = [15.0] * len(mzs)
        mzs += [float(soa["precursor_mz"][i]) + k * 1.0033548350700006 / float(soa["charge"][i]) for k in range(3)]
        tol += [10.0] * 3
        b = 128 + 32 * (lib_stop[i] - lib_start[i]) + 646
        for m, t in zip(mzs, tol):
            lo = np.searchsorted(mzv, m * (1 - t * 1e-6))
            hi = np.searchsorted(mzv, m * (1 + t * 1e-6))
            b += 8 + (hi - lo) * (8 + 8 + 8)
            for tof in range(lo, hi):
                a0, a1 = indptr[tof], indptr[tof + 1]
                e0 = a0 + np.searchsorted(push[a0:a1], p_lo)
                e1 = a0 + np.searchsorted(push[a0:a1], p_hi)
                b += 6 * int(e1 - e0)
        out[j] = b
    return out


rows = np.linspace(0, n - 1, int(os.environ.get("TOUCHED_SAMPLE", 400))).astype(np.int64)
touched = float(touched_bytes(rows).mean()) * n
traffic = gather_traffic = None
tfile = os.environ.get("ADH_IM_TRAFFIC_JSON")  # {"fetch_bytes_per_pass": ..., "write_bytes_per_pass": ...} from tools/profile_r3_im.sh
if tfile and os.path.exists(tfile):
    tj = json.load(open(tfile))
    if tj.get("candidates") == n:
        traffic = tj.get("hbm_bytes_per_pass")
        gather_traffic = tj.get("gather_kernel_hbm_bytes_per_pass")

result = {
    "workload": f"BASELINE configs[3]: timsTOF-style synthetic run, {dia.push_indices.size/1e6:.0f}M events, {S_max} scans, "
                f"{L} frames/cycle, {n_cycles} cycles; {n_prec} precursors x 3 candidates, "
                f"S in [{int(S.min())},{int(S.max())}], F in [{int(F.min())},{int(F.max())}]",
    "metric": "precursors scored/sec", "value": n_prec / (h2h_ms * 1e-3), "unit": "precursors/s", "ms_per_step": h2h_ms,
    "timed_region": "host candidate SoA -> host OutputPsmDF SoA (adh_score_candidates)",
    "resident": {"ms_per_step": res_ms, "value": n_prec / (res_ms * 1e-3)},
    "candidates": n, "valid_fraction": float(valid.mean()), "mean_matched_events": float(matched.mean()),
    "stage_seconds": t_stage,
    "roofline": {
        "bound": "hbm", "achieved": touched / (kernel_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
        "frac": touched / (kernel_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic,
        "kernel": "adh_gather_im_kernel + adh_feature_im_tile4_kernel (four candidates per wavefront, with the order kernels) + adh_feature_im_profiles_kernel", "kernel_ms": kernel_ms,
        "gather_kernel_ms": g_ms, "feature_kernel_ms": f_ms,
        # the gather kernel against the lines it makes HBM fill (PMC traffic of that kernel / its duration)
        "gather_kernel_traffic": gather_traffic,
        "gather_kernel_line_fill_rate_GBps": (gather_traffic / (g_ms * 1e-3) / 1e9) if gather_traffic else None,
        "yardstick": "bytes the candidates' (window, TOF bin, frame range) ranges hold: events x 6 B + index words + "
                     "library slice + plan record + 646 B output row (exact on a strided sample of candidates)",
        "touched_bytes_per_candidate": touched / n,
        "survey_8d_run_b": {
            "algorithmic_bytes_per_candidate": alg_bytes / n,
            "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / 8000.0,
            "note": "SURVEY 8(d) run-B formula: the reference merge-joins every TOF bin of a window from the start of "
                    "the bin (bruker_jit.py:415-502), so whole bins count; the kernels read the candidate's cycles of "
                    "a bin through their indices, which is why this fraction exceeds 1 and is not the one reported",
        },
    },
}
if not os.environ.get("ADH_BENCH_NO_CPU"):
    from oracle import oracle

    quota = cpu_quota_cores()
    cores = int(min(os.cpu_count() or 1, 2 * quota if quota else 64))
    sample = min(n, int(os.environ.get("CPU_SAMPLE", 100_000)))  # (>= 100 000 candidates)
    sub = slice_soa(soa, 0, sample)
    oracle.score_timstof(dia, cols, pack_assembled(slice_soa(soa, 0, 500)), cfgj, n_threads=cores)
    t0 = time.perf_counter()
    exp = oracle.score_timstof(dia, cols, pack_assembled(sub), cfgj, n_threads=cores, with_stats=True)
    cdt = time.perf_counter() - t0
    result["cpu_baseline"] = {
        "value": len(np.unique(sub["precursor_idx"])) / cdt, "unit": "precursors/s", "cores": cores,
        "cpu_quota_cores": quota, "kind": "port", "sample": f"first {sample} candidates, one pass",
        "valid_and_matched_events_identical_to_gpu": bool(
            np.array_equal(exp["valid"].astype(bool), valid[:sample])
            and np.array_equal(exp["stat_matched_peaks"], matched[:sample])),
    }
    result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
# ---- candidate selection on the same run and library (SURVEY.md section 8 f-1 on the ion-mobility layout;
# tools/bench_select_timstof.py is the stand-alone form)
if not os.environ.get("ADH_BENCH_NO_SELECT"):
    from alphadia_amd import _abi  # noqa: E402
    from alphadia_amd.selection import CandidateSelectionConfig, gaussian_kernel  # noqa: E402

    dia.has_mobility = True
    scfg = CandidateSelectionConfig()
    scfg.update(dict(rt_tolerance=float(os.environ.get("RT_TOL", 15.0)), mobility_tolerance=0.1, candidate_count=3,
                     peak_len_rt=3.0, sigma_scale_rt=0.5, peak_len_mobility=0.02))
    kern = gaussian_kernel(dia, scfg.peak_len_rt, scfg.sigma_scale_rt, scfg.kernel_size, scfg.peak_len_mobility,
                           scfg.sigma_scale_mobility)
    pdf = case.library.precursor_df.sort_values("precursor_idx").reset_index(drop=True)
    iso = pdf[[c for c in pdf.columns if c.startswith("i_")]].values

    def pack(df, iso_rows):
        return _abi.pack_precursors(df.precursor_idx.values, df.flat_frag_start_idx.values, df.flat_frag_stop_idx.values,
                                    df.charge.values, df.rt_library.values, df.mobility_library.values,
                                    df.mz_library.values, iso_rows)

    pm = pack(pdf, iso)
    got = ctx.select_candidates(pm, scfg, kern)
    t0 = time.perf_counter()
    got = ctx.select_candidates(pm, scfg, kern)
    wall = time.perf_counter() - t0
    k_ms = ctx.select_time_ms()
    found = got["score"] > 0
    sel = {
        "workload": f"{len(pdf)} precursors, rt tolerance {scfg.rt_tolerance} s, mobility tolerance {scfg.mobility_tolerance}, "
                    f"kernel {kern.shape[0]}x{kern.shape[1]}, 3 candidates",
        "candidates_found": int(found.sum()),
        "tile_scans": int(np.median((got["scan_stop"] - got["scan_start"])[found])) if found.any() else 0,
        "kernel_ms": k_ms, "precursors_per_s_kernel": len(pdf) / (k_ms * 1e-3), "host_call_ms": wall * 1e3,
    }

    def selection_touched(sample_rows):
        """Per precursor: the events of every (window, TOF bin) inside the frames of its rt tolerance and the scans of
        its mobility tolerance x 6 B (push + intensity), per (window, bin) the two index words + tof_indptr + m/z, two
        m/z look-ups per window, the library slice, the precursor record and the candidate rows."""
        push = dia.push_indices
        L = int(dia.cycle.shape[1])
        rt_frames = dia.rt_values
        mob = dia.mobility_values
        out = np.zeros(len(sample_rows), dtype=np.int64)
        for j, i in enumerate(sample_rows):
            rt, im = float(pdf.rt_library.values[i]), float(pdf.mobility_library.values[i])
            f0 = int(np.searchsorted(rt_frames, rt - scfg.rt_tolerance))
            f1 = int(np.searchsorted(rt_frames, rt + scfg.rt_tolerance))
            sc = np.flatnonzero(np.abs(mob - im) <= scfg.mobility_tolerance)
            s0, s1 = (int(sc.min()), int(sc.max()) + 1) if len(sc) else (0, 1)
            a, b = int(pdf.flat_frag_start_idx.values[i]), int(pdf.flat_frag_stop_idx.values[i])
            mzs = [(float(m), float(scfg.fragment_mz_tolerance)) for m in fmz[a:b]]
            mzs += [(float(pdf.mz_library.values[i]) + k * 1.0033548350700006 / float(pdf.charge.values[i]),
                     float(scfg.precursor_mz_tolerance)) for k in range(int(scfg.top_k_precursors))]
            nb = 128 + 32 * (b - a) + 3 * 60
            p_lo, p_hi = np.uint32(f0 * S_max), np.uint32(f1 * S_max)
            for m, t in mzs:
                lo = np.searchsorted(mzv, m * (1 - t * 1e-6))
                hi = np.searchsorted(mzv, m * (1 + t * 1e-6))
                nb += 8 + (hi - lo) * 24
                for tof in range(lo, hi):
                    a0, a1 = indptr[tof], indptr[tof + 1]
                    e0 = a0 + np.searchsorted(push[a0:a1], p_lo)
                    e1 = a0 + np.searchsorted(push[a0:a1], p_hi)
                    if e1 > e0:
                        scan = push[e0:e1] % np.uint32(S_max)
                        nb += 6 * int(((scan >= s0) & (scan < s1)).sum())
            out[j] = nb
        return out

    try:
        rows_sel = np.linspace(0, len(pdf) - 1, int(os.environ.get("TOUCHED_SAMPLE_SEL", 60))).astype(np.int64)
        t_sel = float(selection_touched(rows_sel).mean()) * len(pdf)
        sel["roofline"] = {"bound": "hbm", "achieved": t_sel / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": t_sel / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel_ms": k_ms,
                           "kernel": "adh_select_gather_im_kernel + adh_select_smooth_im_kernel (+ limits, plan)",
                           "touched_bytes_per_precursor": t_sel / len(pdf),
                           "yardstick": "events inside the windows' TOF bins, the rt tolerance and the mobility tolerance x 6 B "
                                        f"+ index words + library slice + outputs; exact on a sample of {len(rows_sel)} precursors"}
        if tfile and os.path.exists(tfile):  # (FETCH_SIZE / WRITE_SIZE passes of tools/profile_r6.sh)
            tj = json.load(open(tfile))
            if tj.get("selection_hbm_bytes_per_pass") and tj.get("candidates") == n:
                sel["roofline"]["traffic"] = tj["selection_hbm_bytes_per_pass"]
    except Exception as exc:  # the yardstick must not take the leg down
        sel["roofline"] = {"skipped": f"{type(exc).__name__}: {exc}"[:200]}
    if not os.environ.get("ADH_BENCH_NO_CPU"):
        from oracle import oracle

        sample = min(len(pdf), 600)
        th = min(64, os.cpu_count() or 1)
        t0 = time.perf_counter()
        exp = oracle.select_timstof(dia, cols, pack(pdf.iloc[:sample], iso[:sample]), scfg, kern, n_threads=th)
        dt = time.perf_counter() - t0
        n_rows = sample * scfg.candidate_count
        sel["cpu_oracle"] = {"precursors_per_s": sample / dt, "threads": th, "sample": sample,
                             "boxes_identical_to_gpu": bool(all(np.array_equal(got[c][:n_rows], exp[c]) for c in got if c != "score"))}
    result["selection"] = sel
print(json.dumps(result))
