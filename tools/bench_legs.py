"""The legs of bench.py beside the headline: fragment competition,
BASELINE configs[4] (3-plex multiplex requantification) and configs[3] (ion mobility).  Every leg
reports HIP-event kernel time, a roofline object against the bytes its inputs hold and the CPU
oracle on the same inputs; bench.py puts them under ``extras`` of its JSON line.  Each leg is also a
command of its own:

    python tools/bench_legs.py fragcomp | multiplex | timstof

The oracle is used here as `cpu_baseline` only (the thing timed beside the GPU) and as the checker of
the GPU result; nothing below is on the product path.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0


def _log(*a):
    print(*a, file=sys.stderr, flush=True)


def _leg_traffic(key: str):
    """HBM bytes per pass of a leg from the committed PMC passes (profiles/legs_traffic.json, tools/profile_r6.sh)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "legs_traffic.json")))[key]["hbm_bytes_per_pass"]
    except Exception:
        return None


def _cpu_threads() -> int:
    from bench import cpu_quota_cores

    quota = cpu_quota_cores()
    return int(max(1, min(os.cpu_count() or 1, quota if quota else 64)))


# --------------------------------------------------------------------------------------------
def fragcomp_leg(ctx, sizes=(100_000, 1_000_000), reps: int = 5) -> dict:
    """Fragment competition at the size FDR hands over (alphadia/fdr/fdr.py:146-163: the PSMs below the
    heuristic FDR of a 1e6-precursor search, 60 DIA windows, K = 12 observed fragments each):
    `adh_fragcomp` against `_compete_for_fragments` as restated in the oracle.

    Bytes: what the inputs hold and the rule has to look at once - per PSM rt, fragment range and flag
    (4 + 16 + 1 B) and its K fragment masses, plus both fragment lists of every (PSM, RT neighbour)
    pair whose overlap is counted (2 * K * 4 B).  The reference additionally reads rt[j] of the whole
    window for every i (n_w^2 * 4 B, reported as `reference_rt_scan_bytes`): the GPU finds the
    neighbours by binary search in an RT-sorted copy instead."""
    import synthetic as syn
    from oracle import oracle

    threads = _cpu_threads()
    out = {}
    for n in sizes:
        t = syn.make_competition_table(int(n), seed=7)
        args = (t["window_start"], t["window_stop"], t["rt"], t["frag_start"], t["frag_stop"], t["mz"], 3, 15)
        got = ctx.fragcomp(*args)
        ms, wall = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            got = ctx.fragcomp(*args)
            wall.append((time.perf_counter() - t0) * 1e3)
            st = ctx.fragcomp_stats()
            ms.append(st["kernel_ms"])
        kernel_ms = float(np.median(ms))
        t0 = time.perf_counter()
        exp = oracle.fragcomp(*args, n_threads=threads)
        cpu_s = time.perf_counter() - t0
        k = int(t["k"])
        # the DataFrame operator on the same table (FragmentCompetition.__call__, fragcomp.py:204-299: candidate keys,
        # fragment ranges, DIA windows, processing order - competition_plan - then the call above, then the frame)
        op_ms = plan_ms = None
        try:
            import pandas as pd
            from alphadia_amd.fragcomp import FragmentCompetition, competition_plan

            rng = np.random.default_rng(5)
            cyc = syn.make_cycle(n_ms2=int(t["n_windows"]), mz_lo=400.0, mz_hi=1000.0)
            win = np.repeat(np.arange(int(t["n_windows"])), (t["window_stop"] - t["window_start"]))
            lo_w, hi_w = cyc[0, 1:, 0, 0], cyc[0, 1:, 0, 1]
            mz_obs = (lo_w[win] + (hi_w[win] - lo_w[win]) * rng.random(int(n))).astype(np.float32)
            pidx = rng.permutation(int(n)).astype(np.uint32)
            psm_df = pd.DataFrame({"precursor_idx": pidx, "rank": np.zeros(int(n), np.uint8), "mz_observed": mz_obs,
                                   "rt_observed": t["rt"], "proba": rng.random(int(n)).astype(np.float32)})
            frag_df = pd.DataFrame({"precursor_idx": np.repeat(pidx, k), "rank": np.zeros(int(n) * k, np.uint8),
                                    "mz_observed": t["mz"]})
            fc = FragmentCompetition()
            fc(psm_df, frag_df, cyc)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                kept = fc(psm_df, frag_df, cyc)
                ts.append((time.perf_counter() - t0) * 1e3)
            op_ms = float(np.median(ts))
            t0 = time.perf_counter()
            competition_plan(psm_df["precursor_idx"].values, psm_df["rank"].values, psm_df["mz_observed"].values,
                             psm_df["proba"].values, frag_df["precursor_idx"].values, frag_df["rank"].values, cyc)
            plan_ms = (time.perf_counter() - t0) * 1e3
            del kept
        except Exception as exc:
            _log(f"[bench] fragcomp operator timing skipped: {type(exc).__name__}: {exc}")
        sizes_w = (t["window_stop"] - t["window_start"]).astype(np.float64)
        touched = float(n) * (21 + 4 * k) + float(st["pairs"]) * 2 * k * 4
        achieved = touched / (kernel_ms * 1e-3) / 1e9
        out[str(n)] = {
            "psms": int(n), "windows": int(t["n_windows"]), "fragments_per_psm": k,
            "removed": int(len(exp) - exp.sum()), "identical_to_cpu": bool(np.array_equal(got, exp)),
            "kernel_ms": kernel_ms, "host_to_host_ms": float(np.median(wall)),
            "operator_ms": op_ms, "operator_plan_ms": plan_ms,
            "neighbour_pairs": int(st["pairs"]), "waiting_psms": int(st["waiting"]), "resolve_rounds": int(st["rounds"]),
            "serial_fallback": bool(st["serial"]),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": _leg_traffic(f"fragment_competition_{n}"),
                         "kernel": "adh_fc_edges_kernel (+ sort, ranges, resolve, final)",
                         "touched_bytes": touched,
                         "reference_rt_scan_bytes": float((sizes_w ** 2).sum() * 4)},
            "cpu_baseline": {"value": n / cpu_s, "unit": "PSMs/s", "seconds": cpu_s, "cores": threads, "kind": "port",
                             "sample": f"the same {n} PSMs, one pass, one window per OpenMP thread (static, as pjit)"},
            "psms_per_s": n / (kernel_ms * 1e-3),
            "gpu_over_cpu_kernel": cpu_s * 1e3 / kernel_ms,
        }
        _log(f"[bench] fragcomp {n}: kernels {kernel_ms:.2f} ms, host->host {np.median(wall):.1f} ms, operator "
             f"{op_ms if op_ms is None else round(op_ms, 1)} ms (plan {plan_ms if plan_ms is None else round(plan_ms, 1)}), cpu {cpu_s:.2f} s "
             f"({threads} threads), identical={out[str(n)]['identical_to_cpu']}")
    return out


# --------------------------------------------------------------------------------------------
def multiplex_leg(ctx, n_groups: int = 75_000, n_cycles: int = 4800, steps: int = 5, cpu_seconds: float = 6.0,
                  threads: int | None = None) -> dict:
    """BASELINE configs[4]: 75 000 elution groups x channels {0, 4, 8, 12} = 300 000 precursors against the
    2 h run, scored the way MultiplexingRequantificationHandler does
    (multiplexing_requantification_handler.py:95-140): best candidate of every group copied to all
    channels, ``CandidateScoringConfig()`` class defaults (top_k_isotopes 4, quant_all False, 15 / 15 ppm)
    + score_grouped, exclude_shared_ions, reference_channel 0.  Timed region as the headline: host
    candidate SoA -> host OutputPsmDF SoA."""
    import synthetic as syn
    from alphadia_amd.distributed import slice_soa
    from alphadia_amd.scoring import (CandidateScoringConfig, assemble_candidates, fragment_columns,
                                      multiplex_candidates, pack_assembled)
    from bench import algorithmic_bytes
    from oracle import oracle

    threads = threads or (os.cpu_count() or 8)
    t0 = time.time()
    mc = syn.make_multiplex_case(n_groups, n_cycles, threads=threads)
    gen_s = time.time() - t0
    cfg = CandidateScoringConfig()
    cfg.update(dict(score_grouped=True, exclude_shared_ions=True, reference_channel=int(mc.channels[0]),
                    experimental_xic=True))
    cfgj = cfg.to_jitclass()
    pdf = mc.library.precursor_df
    multiplexed = multiplex_candidates(mc.psm_df, pdf, channels=list(mc.channels))
    multiplexed["rank"] = 0
    soa = assemble_candidates(multiplexed, pdf.sort_values(by="precursor_idx"), "mz_library", score_grouped=True,
                              reference_channel=cfg.reference_channel, pool=ctx.pinned)
    n = len(soa["precursor_idx"])
    t0 = time.time()
    ctx.stage_run(mc.dia)
    cols = fragment_columns(mc.library.fragment_df, "mz_library")
    ctx.stage_fragments(*cols)
    stage_s = time.time() - t0
    packed = pack_assembled(soa)
    for _ in range(4):
        host = ctx.score_host(packed, cfgj, reuse_buffers=True)
    ctx.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        host = ctx.score_host(packed, cfgj, reuse_buffers=True)
    h2h_ms = (time.perf_counter() - t0) / steps * 1e3
    g_ms, f_ms, launches = ctx.kernel_time_ms(reset=True)
    g_ms, f_ms = g_ms * launches / steps, f_ms * launches / steps
    kernel_ms = g_ms + f_ms
    valid = host["valid"][:n].astype(bool).copy()
    feats = host["features"][:n].copy()
    matched = ctx.device_tables_to_host(names=["stat_matched_peaks"])["stat_matched_peaks"][:n]
    # resident: table + plan in HBM, tables stay in HBM
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
    ctx.kernel_time_ms(reset=True)
    lib_len = soa["frag_stop_idx"].astype(np.int64) - soa["frag_start_idx"].astype(np.int64)
    # exclude_shared_ions leaves the y-ions: K of the formula = fragments that can be selected
    card = mc.library.fragment_df["cardinality"].values
    csum = np.concatenate([[0], np.cumsum(card <= 1)])
    usable = (csum[soa["frag_stop_idx"].astype(np.int64)] - csum[soa["frag_start_idx"].astype(np.int64)]).astype(np.int64)
    alg = float(algorithmic_bytes(mc.dia, soa, cfgj, matched, lib_len, usable_fragments=usable).sum())
    achieved = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    n_prec = len(np.unique(soa["precursor_idx"]))
    traffic = traffic_src = None
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "legs_traffic.json"))).get("multiplex_configs4")
        if tr and int(tr.get("candidates", -1)) == int(n):
            traffic, traffic_src = tr["hbm_bytes_per_pass"], {"file": "profiles/legs_traffic.json",
                                                              "measured_at_commit": tr.get("git_head")}
    except Exception:
        pass
    result = {
        "workload": f"BASELINE configs[4]: {n_groups} elution groups x channels {list(mc.channels)} = {n_prec} precursors "
                    f"(one candidate each, score groups of {len(mc.channels)}, reference channel {mc.channels[0]}) vs "
                    f"{n_cycles} cycles x 61 spectra; class-default CandidateScoringConfig "
                    f"(top_k_isotopes {cfg.top_k_isotopes}, quant_all {cfg.quant_all})",
        "metric": "precursors scored/sec", "value": n_prec / (h2h_ms * 1e-3), "unit": "precursors/s",
        "ms_per_step": h2h_ms, "timed_region": "host candidate SoA -> host OutputPsmDF SoA (adh_score_candidates)",
        "resident": {"ms_per_step": res_ms, "value": n_prec / (res_ms * 1e-3)},
        "candidates": int(n), "valid_fraction": float(valid.mean()), "generation_seconds": gen_s, "stage_seconds": stage_s,
        "kernel_ms": kernel_ms,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel_ms": kernel_ms, "gather_kernel_ms": g_ms, "feature_kernel_ms": f_ms,
                     "algorithmic_bytes_per_candidate": alg / max(n, 1)},
    }
    # CPU oracle on a bounded sample (whole score groups)
    th = _cpu_threads()
    probe = min(n, 400 * th)
    while probe < n and soa["score_group_idx"][probe] == soa["score_group_idx"][probe - 1]:
        probe += 1
    pk = pack_assembled(slice_soa(soa, 0, probe))
    oracle.score(mc.dia, cols, pk, cfgj, n_threads=th)
    t0 = time.perf_counter()
    oracle.score(mc.dia, cols, pk, cfgj, n_threads=th, reuse=oracle.score.last_buffers)
    rate = probe / (time.perf_counter() - t0)
    sample = int(min(n, max(2000, rate * cpu_seconds)))
    while sample < n and soa["score_group_idx"][sample] == soa["score_group_idx"][sample - 1]:
        sample += 1
    sub = slice_soa(soa, 0, sample)
    pk = pack_assembled(sub)
    oracle.score(mc.dia, cols, pk, cfgj, n_threads=th)
    t0 = time.perf_counter()
    exp = oracle.score(mc.dia, cols, pk, cfgj, n_threads=th, reuse=oracle.score.last_buffers)
    dt = time.perf_counter() - t0
    ev = exp["valid"].astype(bool)
    same = bool(np.array_equal(ev, valid[:sample]))
    max_rel = None
    if same and ev.any():
        keep = [f for f in range(46) if f not in (8, 9, 41, 42, 45)]
        fe, fg = exp["features"][ev][:, keep].astype(np.float64), feats[:sample][ev][:, keep].astype(np.float64)
        d = np.abs(fe - fg) / np.maximum(np.maximum(np.abs(fe), np.abs(fg)), 1e-6)
        max_rel = float(np.nanmax(np.where(np.isnan(fe) & np.isnan(fg), 0.0, d)))
    cpu_prec = len(np.unique(sub["precursor_idx"]))
    result["cpu_baseline"] = {"value": cpu_prec / dt, "unit": "precursors/s", "cores": th, "kind": "port",
                              "sample": f"first {sample} candidates (whole score groups) of the same table, one pass of {dt:.2f} s",
                              "valid_identical_to_gpu": same, "max_rel_feature_diff_vs_gpu": max_rel}
    result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
    _log(f"[bench] multiplex: host->host {h2h_ms:.2f} ms, resident {res_ms:.2f} ms, kernels {kernel_ms:.2f} ms "
         f"(gather {g_ms:.2f} + features {f_ms:.2f}), frac {achieved / HBM_PEAK_GBPS:.3f}, valid {valid.mean():.2f}, "
         f"cpu {cpu_prec / dt:,.0f} precursors/s, same valid {same}")
    return result


# --------------------------------------------------------------------------------------------
def transfer_requant_leg(ctx, n_prec: int = 100_000, n_cycles: int = 4800, steps: int = 5, cpu_seconds: float = 5.0,
                         threads: int | None = None) -> dict:
    """The third scoring call site: transfer-library requantification scores the identified candidates again with
    ``top_k_fragments = 9999`` against a library that carries every predicted fragment
    (transfer_library_requantification_handler.py:117-124 -> quantify_candidates, extraction_handler.py:488-507:
    the handler's scoring configuration with only top_k_fragments replaced).  Here: 20-40 fragments per precursor, three
    candidates each, the 2 h run.  Candidates that keep more than 12 fragments are outside the fused kernel
    (adh_fused.hip: a 16-lane group holds 12 fragments + 4 isotopes) and take the two-kernel path
    (adh_gather_kernel -> scratch in HBM -> the wide register kernels of adh_features_fast.hip, 32 or 64 lanes per
    candidate; before round 5 the generic LDS kernel, 5 x slower)."""
    import synthetic as syn
    from alphadia_amd.distributed import slice_soa
    from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled
    from bench import algorithmic_bytes
    from oracle import oracle

    threads = threads or (os.cpu_count() or 8)
    case = syn.make_case(n_prec, n_cycles, config_id=2, per_precursor=3, threads=threads, k_fragments=(20, 40))
    cfg = CandidateScoringConfig()
    cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1, precursor_mz_tolerance=10,
                    fragment_mz_tolerance=15, exclude_shared_ions=True, quant_window=3, quant_all=True,
                    experimental_xic=True, top_k_fragments=9999))
    cfgj = cfg.to_jitclass()
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library", pool=ctx.pinned)
    n = len(soa["precursor_idx"])
    ctx.stage_run(case.dia)
    cols = fragment_columns(case.library.fragment_df, "mz_library")
    ctx.stage_fragments(*cols)
    packed = pack_assembled(soa)
    for _ in range(3):
        host = ctx.score_host(packed, cfgj, reuse_buffers=True)
    ctx.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        host = ctx.score_host(packed, cfgj, reuse_buffers=True)
    h2h_ms = (time.perf_counter() - t0) / steps * 1e3
    g_ms, f_ms, launches = ctx.kernel_time_ms(reset=True)
    g_ms, f_ms = g_ms * launches / steps, f_ms * launches / steps
    kernel_ms = g_ms + f_ms
    valid = host["valid"][:n].astype(bool).copy()
    feats = host["features"][:n].copy()
    width = int(host["fragment_mz_observed"].shape[1])
    filled = float((host["fragment_mz_library"][:n] > 0).sum()) / max(int(valid.sum()), 1)
    matched = ctx.device_tables_to_host(names=["stat_matched_peaks"])["stat_matched_peaks"][:n]
    lib_len = soa["frag_stop_idx"].astype(np.int64) - soa["frag_start_idx"].astype(np.int64)
    alg = float(algorithmic_bytes(case.dia, soa, cfgj, matched, lib_len).sum())
    achieved = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # CPU oracle on a sample of the same table
    sample = min(n, int(os.environ.get("CPU_SAMPLE", 100_000)))  # (>= 100 000 candidates)
    sub = pack_assembled(slice_soa(soa, 0, sample))
    th = _cpu_threads()
    oracle.score(case.dia, cols, sub, cfgj, n_threads=th)
    reps, dt = 0, 0.0
    while dt < cpu_seconds or reps == 0:
        t0 = time.perf_counter()
        exp = oracle.score(case.dia, cols, sub, cfgj, n_threads=th, reuse=oracle.score.last_buffers)
        dt += time.perf_counter() - t0
        reps += 1
    same_valid = bool(np.array_equal(exp["valid"].astype(bool), valid[:sample]))
    max_rel = None
    if same_valid and exp["valid"].any():
        v = exp["valid"].astype(bool)
        keep = [f for f in range(46) if f not in (8, 9, 41, 42, 45)]
        fe, fg = exp["features"][v][:, keep].astype(np.float64), feats[:sample][v][:, keep].astype(np.float64)
        d = np.abs(fe - fg) / np.maximum(np.maximum(np.abs(fe), np.abs(fg)), 1e-6)
        max_rel = float(np.nanmax(np.where(np.isnan(fe) & np.isnan(fg), 0.0, d)))
    n_p = len(np.unique(soa["precursor_idx"]))
    res = {
        "workload": f"transfer-library requantification: {n_p} precursors x 3 candidates, library of 20-40 fragments per "
                    f"precursor, top_k_fragments 9999 (tables {width} columns wide, {filled:.1f} filled per valid candidate) vs "
                    f"{n_cycles} cycles x 61 spectra",
        "metric": "precursors scored/sec", "value": n_p / (h2h_ms * 1e-3), "unit": "precursors/s", "ms_per_step": h2h_ms,
        "candidates": int(n), "valid_fraction": float(valid.mean()), "kernel_ms": kernel_ms,
        "kernel_ns_per_candidate": kernel_ms * 1e6 / max(n, 1),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": _leg_traffic("transfer_requant"), "kernel_ms": kernel_ms, "gather_kernel_ms": g_ms,
                     "feature_kernel_ms": f_ms, "algorithmic_bytes_per_candidate": alg / max(n, 1),
                     "kernel": "adh_gather_kernel + adh_feature_wide_kernel<heavy bodies>, <light bodies> (the two-kernel path: more than 12 fragments per candidate; the wide register kernels hold 17 ... 64)"},
        "cpu_baseline": {"value": (sample / 3.0) / (dt / reps), "unit": "precursors/s", "cores": th, "kind": "port",
                         "sample": f"first {sample} candidates, {reps} x {dt / reps:.2f} s, {th} OpenMP threads",
                         "valid_identical_to_gpu": same_valid, "max_rel_feature_diff_vs_gpu": max_rel},
    }
    _log(f"[bench] transfer requant: host->host {h2h_ms:.2f} ms per {n} candidates, kernels {kernel_ms:.2f} ms "
         f"({res['kernel_ns_per_candidate']:.1f} ns per candidate), frac {res['roofline']['frac']:.3f}, same valid {same_valid}")
    return res


def operator_leg(case, cfg, reps: int = 3) -> dict:
    """The plug-in operator itself: ``HipCandidateScoring.__call__`` from the candidate
    DataFrame to ``(features_df, fragments_df)`` on the headline table - what every call site of the reference
    enters through (scoring.py:582-661) - with the wall time of its stages."""
    from alphadia_amd.scoring import HipCandidateScoring

    scorer = HipCandidateScoring(dia_data=case.dia, precursors_flat=case.library.precursor_df,
                                 fragments_flat=case.library.fragment_df, rt_column="rt_library",
                                 mobility_column="mobility_library", precursor_mz_column="mz_library",
                                 fragment_mz_column="mz_library", config=cfg, device=None)
    runs = []
    rows = (0, 0)
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        features_df, fragments_df = scorer(case.candidates_df)
        wall = (time.perf_counter() - t0) * 1e3
        runs.append(dict(scorer.last_timings, wall_ms=wall))
        rows = (len(features_df), len(fragments_df))
        cols = (features_df.shape[1], fragments_df.shape[1])
        del features_df, fragments_df
    best = min(runs[1:], key=lambda r: r["wall_ms"])
    out = {"candidates": int(len(case.candidates_df)), "features_rows": rows[0], "fragments_rows": rows[1],
           "features_columns": cols[0], "fragments_columns": cols[1],
           "operator_ms": best["wall_ms"], "stages_ms": {k: v for k, v in best.items() if k != "wall_ms"},
           "first_call_ms": runs[0]["wall_ms"], "all_calls_ms": [r["wall_ms"] for r in runs],
           "region": "candidates_df -> assemble (lexsort, library lookup) -> adh_score_candidates -> features_df + "
                     "fragments_df (valid rows / filled fragment slots gathered column by column)"}
    _log(f"[bench] operator: {out['operator_ms']:.0f} ms for {out['candidates']} candidates "
         f"({', '.join(f'{k} {v:.0f}' for k, v in out['stages_ms'].items())}); first call {out['first_call_ms']:.0f} ms")
    return out


# --------------------------------------------------------------------------------------------
def timstof_leg(full_size: bool = True, timeout: float = 600.0) -> dict:
    """BASELINE configs[3] through tools/bench_timstof.py in a process of its own (its run, its 3 GB index
    and its scratch slab do not pile on top of the headline's)."""
    env = dict(os.environ)
    if full_size:
        env.update(N_PREC="200000", N_CYCLES="2000", SCAN_MAX="918", N_TOF="400000", EVENTS_PER_PUSH="30")
    traffic = os.path.join(ROOT, "profiles", "timstof_traffic.json")
    if os.path.exists(traffic):
        env["ADH_IM_TRAFFIC_JSON"] = traffic
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_timstof.py")], env=env, capture_output=True,
                       text=True, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"bench_timstof.py failed ({p.returncode}): {p.stderr[-400:]}")
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    res["leg_wall_seconds"] = time.time() - t0
    _log(f"[bench] timstof: host->host {res['ms_per_step']:.2f} ms, resident {res['resident']['ms_per_step']:.2f} ms, "
         f"kernels {res['roofline']['kernel_ms']:.2f} ms, leg took {res['leg_wall_seconds']:.0f} s")
    return res


# --------------------------------------------------------------------------------------------
def _tool_leg(script: str, args=(), env_extra=None, timeout: float = 600.0) -> dict:
    """One of the stand-alone benches under tools/ in a process of its own; its last JSON line."""
    env = dict(os.environ)
    env.update(env_extra or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args], env=env, capture_output=True, text=True,
                       timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"{script} failed ({p.returncode}): {p.stderr[-400:]}")
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    res["leg_wall_seconds"] = time.time() - t0
    return res


def selection_leg(n_prec: int = 100_000, timeout: float = 600.0) -> dict:
    """Candidate selection (selection.py:620-660) through tools/bench_select.py: 100 000
    precursors against the 2 h run, rt tolerance 60 s, 3 candidates each; kernel time, and the CPU oracle on a sample
    of the same precursors (boxes compared)."""
    res = _tool_leg("bench_select.py", env_extra={"N_PREC": str(n_prec)}, timeout=timeout)
    _log(f"[bench] selection: kernels {res['kernel_ms']:.2f} ms per {n_prec} precursors "
         f"({res['precursors_per_s_kernel'] / 1e6:.1f} M precursors/s), leg took {res['leg_wall_seconds']:.0f} s")
    return res


def fdr_leg(timeout: float = 600.0) -> dict:
    """The FDR stage on the device at the size of the headline table through
    tools/bench_fdr.py: classifier training / inference, q-values, best row per group.  Without its CPU leg (the
    plain-PyTorch oracle needs an `import torch`, minutes on a cold box): profiles/r01_fdr_bench.json holds it."""
    res = _tool_leg("bench_fdr.py", args=("--cpu-steps", "0"), env_extra={"ADH_FDR_NUMPY_INIT": "1"}, timeout=timeout)
    _log(f"[bench] fdr: fit {res['fit_kernels_ms']:.0f} ms of kernels ({res['fit_wall_s']:.2f} s wall), predict "
         f"{res['predict_kernel_ms']:.1f} ms, q-values {res['q_values_wall_ms']:.1f} ms, leg took {res['leg_wall_seconds']:.0f} s")
    return res


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "fragcomp"
    if which == "selection":
        print(json.dumps(selection_leg()))
    elif which == "fdr":
        print(json.dumps(fdr_leg()))
    elif which == "timstof":
        print(json.dumps(timstof_leg(full_size=not os.environ.get("REDUCED"))))
    else:
        from alphadia_amd import runtime

        ctx = runtime.get_context(0)
        if which == "fragcomp":
            sizes = tuple(int(x) for x in os.environ.get("FC_SIZES", "100000,1000000").split(","))
            print(json.dumps(fragcomp_leg(ctx, sizes=sizes)))
        elif which == "transfer":
            print(json.dumps(transfer_requant_leg(ctx, n_prec=int(os.environ.get("N_PREC", 100000)))))
        elif which == "multiplex":
            print(json.dumps(multiplex_leg(ctx, n_groups=int(os.environ.get("N_GROUPS", 75000)),
                                           n_cycles=int(os.environ.get("N_CYCLES", 4800)))))
        else:
            raise SystemExit(f"unknown leg {which}")
