"""Benchmark of the candidate-scoring hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): precursors scored / s on the 1e6-precursor predicted library (3 candidates
per precursor = 3e6 candidates) against the 2 h synthetic DIA run (4800 cycles x 61 spectra, 4.9e8
peaks).  N = 1 scores the whole library on one GPU; N > 1 shards THE SAME library by contiguous
score-group ranges over the ranks (strong scaling) and reassembles the computed tables in HBM with
ONE RCCL all-gather over xGMI per step.

A "step" is one pass of the hot path over the rank's candidate shard in the region SURVEY.md
section 8(d) / BASELINE.md section 2 define (the reference's region is CandidateScoring.__call__
between assembling the SoA and collecting DataFrames, search/scoring/scoring.py:634-643):

    host candidate SoA -> H2D -> plan (on the device) -> gather + feature kernels
                       -> [all-gather of the computed tables] -> D2H -> host OutputPsmDF SoA

through ONE C-ABI call (adh_score_candidates: chunked, H2D / kernels / D2H overlap on three
streams).  The run and the fragment library are staged in HBM once (reported, excluded).  No torch
anywhere in this file: device memory, streams and the communicator live behind the C ABI.

The JSON line also carries
  resident      - the same step with the candidate table and its plan already resident in HBM and
                  the tables left in HBM (kernels + zero fill only)
  roofline      - algorithmic bytes (SURVEY.md section 8d formula) / kernel time per step (HIP
                  events on the launch stream) vs the 8 TB/s HBM peak
  cpu_baseline  - the CPU oracle (oracle/, a C++ restatement of the reference's Numba path) timed on
                  this host's cores over the same host -> host region on a bounded sample of the
                  same candidates (rank 0, N = 1 only), at 1 thread and at the fastest thread count.
                  Checker code is used here only as the thing timed beside the GPU.
  small_batches - host -> host latency at the batch sizes of the reference's optimisation loop
                  (optimization_lock.py:72-108: 8 000 elution groups, doubling)
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))  # the synthetic data generators live with the tests

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(dia, soa, cfg, matched_peaks, lib_slice_len, usable_fragments=None):
    """bytes the reference algorithm has to touch per candidate.

    B = sum_probes [4 * (ceil(log2 P_s) + 1)] + 8 * matched_peaks + 18 * K_lib + 64
        + (46 * 4 + K * 38 + 6)
    with one probe per (fragment, observation, cycle) in MS2 and per (isotope, cycle)
    in MS1 and P_s the number of peaks of the probed spectrum.  Auxiliary index reads
    of this implementation are not counted.  ``usable_fragments``: fragments of the library slice that can be
    selected at all (``exclude_shared_ions`` drops cardinality > 1); default = the whole slice.
    """
    L = dia.cycle_len
    n = len(soa["precursor_idx"])
    counts = (dia.peak_stop_idx_list - dia.peak_start_idx_list).astype(np.int64)
    steps = (np.ceil(np.log2(np.maximum(counts, 1))).astype(np.int64) + 1) * 4  # bytes per probe
    steps_2d = steps.reshape(-1, L)  # [cycle, position]
    csum = np.concatenate([np.zeros((1, L), np.int64), np.cumsum(steps_2d, axis=0)], axis=0)
    c0 = soa["frame_start"] // L
    c1 = soa["frame_stop"] // L
    K = np.minimum(lib_slice_len if usable_fragments is None else usable_fragments,
                   int(cfg.top_k_fragments)).astype(np.int64)
    I = min(int(cfg.top_k_isotopes), soa["isotope_intensity"].shape[1])
    # MS1 probes: position 0 of every cycle in the window
    ms1 = (csum[c1, 0] - csum[c0, 0]) * I
    # MS2 probes: every overlapping window position
    iso_hi = soa["precursor_mz"].astype(np.float64) + (I - 1) * 1.0033548350700006 / soa["charge"]
    q_lo = soa["precursor_mz"].astype(np.float64) - 0.5
    q_hi = iso_hi + 0.5
    ms2 = np.zeros(n, dtype=np.int64)
    lo_w, hi_w = dia.cycle[0, :, 0, 0], dia.cycle[0, :, 0, 1]
    for pos in range(1, L):
        ov = (q_lo <= hi_w[pos]) & (q_hi >= lo_w[pos])
        if ov.any():
            ms2[ov] += (csum[c1[ov], pos] - csum[c0[ov], pos]) * K[ov]
    per = ms1 + ms2 + 8 * matched_peaks.astype(np.int64) + 18 * lib_slice_len + 64 + (46 * 4 + K * 38 + 6)
    return per


def cpu_quota_cores() -> float | None:
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def pinned_shard(ctx, soa: dict, a: int, b: int) -> dict:
    """Rows [a, b) of the assembled SoA with the uploaded columns in page-locked memory."""
    n = len(soa["precursor_idx"])
    out = {}
    for k, v in soa.items():
        if isinstance(v, np.ndarray) and v.shape[:1] == (n,):
            out[k] = ctx.pinned.take("cand:" + k, v[a:b])
        else:
            out[k] = v
    return out


def shared_case(n_prec: int, n_cycles: int, threads: int, rank: int, local_rank: int, world: int):
    """The synthetic workload.  With several ranks on a node the 4.9e8-peak run is generated ONCE (local rank 0,
    all host threads) and the other ranks map its arrays from /dev/shm - N generations side by side would share
    the host's CPU quota and hold N x 3.9 GB of peaks, and the first minutes of an 8-GPU run would go there.
    The library and the candidate table are cheap and deterministic: every rank builds its own."""
    import synthetic as syn

    if world == 1:
        return syn.make_case(n_prec, n_cycles, config_id=2, per_precursor=3, threads=threads)
    from alphadia_amd.runtime import _launch_nonce

    tag = f"adh_bench_{os.getuid()}_{_launch_nonce().hex()[:16]}_{n_prec}_{n_cycles}"
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", tag)
    arrays = ("rt_values", "peak_start_idx_list", "peak_stop_idx_list", "mz_values", "intensity_values", "cycle")
    done = base + ".done"
    if local_rank == 0:
        case = syn.make_case(n_prec, n_cycles, config_id=2, per_precursor=3, threads=threads)
        for name in arrays:
            np.save(f"{base}.{name}.npy", getattr(case.dia, name))
        with open(done + ".tmp", "w") as f:
            f.write("ok")
        os.rename(done + ".tmp", done)
        return case
    # the run is the expensive part; library + candidates come from the same seeds
    light = syn.make_case(n_prec, n_cycles, config_id=2, per_precursor=3, threads=1, run=False)
    t_wait = time.time()
    while not os.path.exists(done):
        if time.time() - t_wait > 1800:
            raise SystemExit(f"rank {rank}: local rank 0 never published the synthetic run ({done})")
        time.sleep(0.2)
    dia = syn.AlphaRawArrays(**{name: np.load(f"{base}.{name}.npy", mmap_mode="r") for name in arrays})
    return syn.SyntheticCase(dia, light.library, light.candidates_df, light.apex_cycle)


def release_shared_case(local_rank: int, world: int, n_prec: int, n_cycles: int):
    """Unlink the mapped run once every rank has staged it (the mappings stay valid until they are dropped)."""
    if world == 1 or local_rank != 0:
        return
    import glob

    from alphadia_amd.runtime import _launch_nonce

    tag = f"adh_bench_{os.getuid()}_{_launch_nonce().hex()[:16]}_{n_prec}_{n_cycles}"
    for base in ("/dev/shm", "/tmp"):
        for f in glob.glob(os.path.join(base, tag + ".*")):
            try:
                os.unlink(f)
            except OSError:
                pass


PRIME = 4  # untimed calls before the warm-up steps (see below)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precursors", type=int, default=1_000_000,
                    help="library size (total, sharded over the ranks); BASELINE headline = 1e6")
    ap.add_argument("--cycles", type=int, default=4800)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs beside the headline (small batches, selection, fragment competition, "
                         "configs[3] ion mobility, configs[4] multiplex)")
    ap.add_argument("--extras-seconds", type=float, default=400.0,
                    help="wall-clock budget of the legs under `extras`; a leg that would start past it is skipped")
    ap.add_argument("--pageable", action="store_true", help="host buffers in pageable memory (for comparison)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` for N > 1")

    import synthetic as syn
    from alphadia_amd import _abi, runtime
    from alphadia_amd.distributed import shard_bounds
    from alphadia_amd.scoring import (
        CandidateScoringConfig,
        assemble_candidates,
        fragment_columns,
        pack_assembled,
    )

    if runtime.device_count() <= local_rank:
        raise SystemExit(f"bench.py needs an MI355X for local rank {local_rank}")
    ctx = runtime.get_context(local_rank)

    # ---------------- workload: the 1e6-precursor library, sharded over the ranks ----------------
    n_prec_total = args.precursors
    t0 = time.time()
    threads = os.cpu_count() or 8
    case = shared_case(n_prec_total, args.cycles, threads, rank, local_rank, world)
    log(f"[bench] synthetic run: {case.dia.n_spectra} spectra, {case.dia.mz_values.size/1e6:.1f}M peaks, "
        f"{len(case.candidates_df)} candidates, ready in {time.time()-t0:.1f}s ({threads} threads"
        f"{', generated once per node and mapped by the other ranks' if world > 1 else ''})")
    cfg = CandidateScoringConfig()
    # ClassicExtractionHandler defaults (extraction_handler.py:370-376,400-409; default.yaml:158-199)
    cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1,
                    precursor_mz_tolerance=10, fragment_mz_tolerance=15, exclude_shared_ions=True,
                    quant_window=3, quant_all=True, experimental_xic=True, top_k_fragments=12))
    cfgj = cfg.to_jitclass()
    soa_all = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    n_all = len(soa_all["precursor_idx"])
    bounds = [shard_bounds(soa_all["score_group_idx"], r, world) for r in range(world)]
    a, b = bounds[rank]
    n_local = b - a
    max_rows = max(e - s for s, e in bounds)
    if args.pageable:
        from alphadia_amd.distributed import slice_soa

        soa = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in slice_soa(soa_all, a, b).items()}
    else:
        soa = pinned_shard(ctx, soa_all, a, b)
    n_prec_local = len(np.unique(soa["precursor_idx"]))

    # ---------------- staging (one-time, excluded from the metric) ----------------
    t0 = time.time()
    ctx.stage_run(case.dia)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
    t_stage = time.time() - t0
    log(f"[bench] staged run+library in {t_stage:.2f}s")
    with_comm = world > 1 or bool(os.environ.get("ADH_BENCH_FORCE_COMM"))  # (debug: the RCCL path on one GPU)
    ranks_seen = [1, 1]
    if with_comm:
        ctx.comm_init(rank, world, max_rows)
        # what RCCL itself says about the communicator (ncclCommCount), smallest and largest answer over the
        # ranks: a run on N GPUs shows here that N ranks met
        _, w_seen = ctx.comm_info()
        ranks_seen = [int(round(-ctx.all_reduce_max(-float(w_seen)))), int(round(ctx.all_reduce_max(float(w_seen))))]
    if world > 1:
        ctx.barrier()  # every rank holds its copy of the run in HBM: the files of the shared generation can go
        release_shared_case(local_rank, world, n_prec_total, args.cycles)

    packed = pack_assembled(soa)
    reuse = not args.pageable

    def step():
        return ctx.score_host(packed, cfgj, reuse_buffers=reuse)

    def fence():
        ctx.comm_wait()           # the all-gather of the last step
        ctx.device_synchronize()
        ctx.barrier()
        ctx.device_synchronize()

    # the first calls of a process are 2-3x slower whatever the kernels do (page-locked buffers touched for the
    # first time, copy engines and the PCIe link ramping up): PRIME untimed calls precede the W warm-up steps
    # the caller asked for, so that the record does not depend on how small W is
    for _ in range(PRIME):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    ctx.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        host = step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = ctx.all_reduce_max(elapsed)
    gather_ms, feature_ms, launches = ctx.kernel_time_ms(reset=True)
    per_step = launches / max(args.steps, 1)
    gather_ms, feature_ms = gather_ms * per_step, feature_ms * per_step
    kernel_ms = gather_ms + feature_ms

    # ---------------- results of the last step (sanity + roofline inputs) ----------------
    valid = host["valid"][:n_local].astype(bool)
    print(f"[bench] rank {rank}: {int(valid.sum())}/{n_local} candidates valid", file=sys.stderr, flush=True)
    matched = ctx.device_tables_to_host(names=["stat_matched_peaks"])["stat_matched_peaks"][:n_local]
    features_last = host["features"].copy() if world == 1 else None
    if with_comm:
        # every rank must now hold every rank's computed tables: its own slice must equal its host tables
        mine = ctx.gathered_tables(rank, rows=n_local)
        assert np.array_equal(mine["valid"], host["valid"]), "gathered tables differ from the local ones"
        assert np.array_equal(mine["features"], host["features"], equal_nan=True)
        peer = (rank + 1) % world
        got = ctx.gathered_tables(peer, rows=bounds[peer][1] - bounds[peer][0])
        assert got["valid"].any(), f"rank {rank}: the tables gathered from rank {peer} are empty"

    lib_len = (soa["frag_stop_idx"].astype(np.int64) - soa["frag_start_idx"].astype(np.int64))
    per_cand_bytes = algorithmic_bytes(case.dia, soa, cfgj, matched, lib_len)
    bytes_per_launch = float(per_cand_bytes.sum())
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    # ---------------- the same step with table and plan resident in HBM, outputs left in HBM ----------------
    ctx.upload_candidates(packed)
    view = ctx.device_tables()
    stream = ctx.stream_handle()
    resident_ms = None
    try:
        for _ in range(2):
            ctx.zero_device_tables(stream)
            ctx.score_uploaded(cfgj, view, stream)
        ctx.synchronize()
        reps_r = max(3, min(args.steps, 10))
        t0 = time.perf_counter()
        for _ in range(reps_r):
            ctx.zero_device_tables(stream)
            ctx.score_uploaded(cfgj, view, stream)
        ctx.synchronize()
        resident_ms = (time.perf_counter() - t0) / reps_r * 1e3
        ctx.kernel_time_ms(reset=True)
    except runtime.HipBackendError as exc:
        log(f"[bench] resident leg skipped: {exc}")

    # ---------------- the step through the compact entry point (what the plug-in operator calls) ----------------
    # adh_score_candidates_compact: valid rows and filled fragment slots leave the device as columns - 0.28 KB instead
    # of 0.45 KB per candidate on the link, which is what the padded step waits for.  Capacity arrays kept between the
    # calls (a search scores batch after batch); reported beside `value`, never as `value`.
    compact = None
    if world == 1 and not os.environ.get("ADH_BENCH_NO_COMPACT"):
        try:
            keep: dict = {}
            for _ in range(3):
                comp = ctx.score_host_compact(packed, cfgj, buffers=keep)
            reps_c = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(reps_c):
                comp = ctx.score_host_compact(packed, cfgj, buffers=keep)
            compact_ms = (time.perf_counter() - t0) / reps_c * 1e3
            rows_c = comp["row"]
            same_rows = bool(np.array_equal(rows_c, np.flatnonzero(valid)))
            same_feat = bool(np.array_equal(comp["features"].T, features_last[rows_c], equal_nan=True)) if same_rows else False
            wire = int(comp["features"].nbytes + 5 * len(rows_c) + 22 * len(comp["fragment_row"]))
            compact = {"ms_per_step": compact_ms, "value": n_prec_total / (compact_ms * 1e-3), "unit": "precursors/s",
                       "wire_bytes": wire, "rows": int(len(rows_c)), "fragment_slots": int(len(comp["fragment_row"])),
                       "same_rows_as_padded": same_rows, "same_features_as_padded": same_feat,
                       "region": "host candidate SoA -> adh_score_candidates_compact -> host columns of the valid rows and "
                                 "the filled fragment slots (capacity arrays reused between the calls)"}
            log(f"[bench] compact entry: {compact_ms:.2f} ms per step, {wire / 1e9:.2f} GB on the link, "
                f"rows equal {same_rows}, features equal {same_feat}")
            ctx.kernel_time_ms(reset=True)
        except runtime.HipBackendError as exc:
            log(f"[bench] compact leg skipped: {exc}")

    value = n_prec_total * args.steps / elapsed
    result = {
        "metric": "precursors scored/sec",
        "value": value,
        "unit": "precursors/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32/f64",
        "data": "synthetic",
        "config": {
            "workload": f"{n_prec_total}-precursor predicted library x 3 candidates vs "
                        f"{'2h ' if args.cycles == 4800 else ''}synthetic Thermo-style DIA run "
                        f"({args.cycles} cycles x 61 spectra); BASELINE configs[2] "
                        f"{'on one GPU' if world == 1 else f'sharded over {world} GPUs'}",
            "timed_region": "host candidate SoA -> H2D -> device plan -> gather + feature kernels -> "
                            + ("RCCL all-gather of the computed tables -> " if world > 1 else "")
                            + "D2H -> host OutputPsmDF SoA (one adh_score_candidates call per step)",
            "host_buffers": "pageable" if args.pageable else "page-locked (adh_host_alloc)",
            "precursors_total": n_prec_total,
            "candidates_total": int(n_all),
            "candidates_per_gpu": int(n_local),
            "cycles": args.cycles,
            "peaks": int(case.dia.mz_values.size),
            "parallelism": (f"score-group shards x{world}, computed tables all-gathered over RCCL "
                            f"(overlaps the D2H of the same step and the next step)") if world > 1 else "single GPU",
            "rccl_ranks_seen": ranks_seen[1] if ranks_seen[0] == ranks_seen[1] else ranks_seen,
            "valid_fraction": float(valid.mean()) if n_local else 0.0,
            "priming_steps_before_warmup": PRIME,
            "candidates_per_s": float(n_all * args.steps / elapsed),
            "stage_seconds": t_stage,
        },
        "compact": compact,
        "resident": None if resident_ms is None else {
            "ms_per_step": resident_ms,
            "value": n_prec_local / (resident_ms * 1e-3) * (world if world > 1 else 1),
            "unit": "precursors/s",
            "region": "candidate table + plan resident in HBM -> zero tables + kernels -> tables stay in HBM",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel": "adh_fused_kernel<FM_MIN, FM_MAX, observations> (gather + features in one kernel; every candidate of this "
                      "workload takes it) + the two-kernel fallback classes: the hot path, summed over the chunks of one step",
            "kernel_ms": kernel_ms,
            "gather_kernel_ms": gather_ms,
            "feature_kernel_ms": feature_ms,
            "launches_per_step": per_step,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "algorithmic_bytes_per_candidate": bytes_per_launch / max(n_local, 1),
        },
    }
    # HBM traffic: PMC counters need rocprofv3 passes of their own (tools/profile_r4.sh), so the line carries the
    # figure of the committed profile of this workload - and says so, with the commit it was measured at
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            if tr.get("candidates_per_gpu") == int(n_local):
                result["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
                result["roofline"]["traffic_source"] = {
                    "file": "profiles/pmc_traffic.json", "measured_at_commit": tr.get("git_head"),
                    "recipe": tr.get("recipe", "tools/profile_r4.sh"),
                    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run"}
        except Exception:
            pass

    extras = rank == 0 and world == 1 and not args.no_extras
    # ---------------- small batches: the optimisation loop of the reference ----------------
    if extras:
        try:
            table = {}
            from alphadia_amd.distributed import slice_soa

            for n_prec in (8_000, 16_000, 32_000, 64_000, 100_000):
                m = min(n_local, 3 * n_prec)
                sub = pack_assembled(slice_soa(soa, 0, m))
                ctx.score_host(sub, cfgj, reuse_buffers=reuse)
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.score_host(sub, cfgj, reuse_buffers=reuse)
                table[str(n_prec)] = (time.perf_counter() - t0) / reps * 1e3
                g, f, nl = ctx.kernel_time_ms(reset=True)
                table[str(n_prec) + "_kernel_ms"] = (g + f) * nl / (reps + 1)
            result["config"]["small_batch_host_to_host_ms"] = table
        except Exception as exc:  # the metric does not depend on this leg
            log(f"[bench] small-batch leg skipped: {exc}")

    # ---------------- the shards of an eight-GPU run, one after the other on this GPU ----------------
    # (no 8-GPU node has been available to any round: this is what each rank of `--gpus 8` would do per step, with the
    # communicator's table layout and all-gather enqueue inside the call - tools/bench_shard.py)
    if extras and not with_comm:
        try:
            from tools import bench_shard

            leg = bench_shard.shard_leg(ctx, soa, cfgj, ms_per_step=elapsed / args.steps * 1e3, ways=8, reps=7)
            result["shard_8way"] = {k: v for k, v in leg.items() if k != "shards"}
            result["shard_8way"]["shard_ms"] = [round(s["ms"], 3) for s in leg["shards"]]
            result["config"]["projected_scaling_8"] = leg["projected_scaling_8"]
            result["config"]["projected_scaling_8_contended"] = leg["projected_scaling_8_contended"]
            result["config"]["all_gather_bytes_per_rank"] = leg["all_gather"]["bytes_contributed_per_rank"]
            result["config"]["all_gather_ms_estimate"] = leg["all_gather"]["ms_at_70pct_of_link_rate"]
            result["config"]["max_shard_ms_8way"] = leg["max_shard_ms"]
            # the buffers of the full table come back for the legs below (the shards used smaller ones)
            ctx.score_host(packed, cfgj, reuse_buffers=reuse)
        except Exception as exc:  # the metric does not depend on this leg
            log(f"[bench] shard leg skipped: {type(exc).__name__}: {exc}")

    # ---------------- the step before scoring (candidate selection), reported next to the metric ----
    if extras:
        try:
            from alphadia_amd.selection import CandidateSelectionConfig, gaussian_kernel

            scfg = CandidateSelectionConfig()
            scfg.update(dict(rt_tolerance=60.0, candidate_count=3))
            kern = gaussian_kernel(case.dia, scfg.peak_len_rt, scfg.sigma_scale_rt, scfg.kernel_size)
            pdf = case.library.precursor_df.sort_values("precursor_idx").reset_index(drop=True)
            iso_cols = [c for c in pdf.columns if c.startswith("i_")]
            pm = _abi.pack_precursors(pdf.precursor_idx.values, pdf.flat_frag_start_idx.values,
                                      pdf.flat_frag_stop_idx.values, pdf.charge.values, pdf.rt_library.values,
                                      pdf.mobility_library.values, pdf.mz_library.values, pdf[iso_cols].values)
            ctx.select_candidates(pm, scfg, kern)
            sel = ctx.select_candidates(pm, scfg, kern)
            sel_ms = ctx.select_time_ms()
            result["config"]["selection_kernel_ms"] = sel_ms
            result["config"]["selection_precursors_per_s"] = len(pdf) / (sel_ms * 1e-3)
            result["config"]["selection_candidates_found"] = int((sel["score"] > 0).sum())
        except Exception as exc:  # the metric does not depend on this leg
            log(f"[bench] selection leg skipped: {exc}")

    # ---------------- CPU baseline: the oracle on this host's cores, same host -> host region ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from alphadia_amd.distributed import slice_soa
        from oracle import oracle

        ncpu = os.cpu_count() or 1
        quota = cpu_quota_cores()
        cols = fragment_columns(case.library.fragment_df, "mz_library")
        # pick the thread count that is fastest on this host.  The container may be capped by a CFS
        # quota well below the visible CPUs (the pool's GPU boxes: 16 cores' worth of 256 hardware
        # threads); more threads than about twice the quota only add throttling
        best = (0.0, 1)
        tried = {}
        cand_threads = {t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}
        if quota:
            cand_threads = {t for t in cand_threads if t <= 4 * quota} | {max(1, int(quota))}
        for th in sorted(cand_threads):
            probe = min(400 * th, n_local)
            pk = pack_assembled(slice_soa(soa, 0, probe))
            oracle.score(case.dia, cols, pk, cfgj, n_threads=th)
            t0 = time.perf_counter()
            oracle.score(case.dia, cols, pk, cfgj, n_threads=th, reuse=oracle.score.last_buffers)
            rate = probe / (time.perf_counter() - t0)
            tried[th] = rate
            log(f"[bench] cpu oracle {th:4d} threads: {rate:,.0f} candidates/s")
            if rate > best[0]:
                best = (rate, th)
        rate, cores = best
        # single-thread rate (the reference's pjit loop at thread_count = 1), short sample
        probe1 = min(1500, n_local)
        pk1 = pack_assembled(slice_soa(soa, 0, probe1))
        oracle.score(case.dia, cols, pk1, cfgj, n_threads=1)
        t0 = time.perf_counter()
        oracle.score(case.dia, cols, pk1, cfgj, n_threads=1, reuse=oracle.score.last_buffers)
        rate_1 = probe1 / (time.perf_counter() - t0)
        # sustained rate per thread count: the probes above are bursts of a fraction of a second, which a CFS quota
        # lets run ahead of the sustained rate; the record is the rate over seconds, at the quota's thread count and
        # at the fastest probed one, and `value` is the better of the two
        sustained = {}
        best_s = None
        counts = sorted({cores} | ({max(1, int(quota))} if quota else set()))
        budget = min(args.cpu_seconds, 10.0) * 2.0 / len(counts)
        for th in counts:
            sample_t = int(min(n_local, max(2000, tried.get(th, rate) * budget * 0.6)))
            sub_t = slice_soa(soa, 0, sample_t)
            pk_t = pack_assembled(sub_t)
            oracle.score(case.dia, cols, pk_t, cfgj, n_threads=th)  # touches the output pages
            reps_t, dt_t = 0, 0.0
            while dt_t < budget or reps_t == 0:  # repeat the sample for a stable rate
                t0 = time.perf_counter()
                exp_t = oracle.score(case.dia, cols, pk_t, cfgj, n_threads=th, reuse=oracle.score.last_buffers)
                dt_t += time.perf_counter() - t0
                reps_t += 1
            sustained[th] = sample_t * reps_t / dt_t
            log(f"[bench] cpu oracle {th:4d} threads sustained: {sustained[th]:,.0f} candidates/s ({reps_t} x {dt_t / reps_t:.2f} s)")
            if best_s is None or sustained[th] > sustained[best_s[0]]:
                best_s = (th, sample_t, sub_t, reps_t, dt_t / reps_t, {k: np.array(v, copy=True) for k, v in exp_t.items() if k in ("valid", "features")})
        cores, sample, sub, reps, dt, exp = best_s
        cpu_prec = len(np.unique(sub["precursor_idx"]))
        same_valid = bool(np.array_equal(exp["valid"].astype(bool), valid[:sample]))
        max_rel = max_ppm = None
        if same_valid and exp["valid"].any():
            fe = exp["features"][exp["valid"].astype(bool)]
            fg = features_last[:sample][valid[:sample]]
            # features 8, 9, 41, 42, 45 are mass errors in ppm (differences of nearly equal m/z): they are
            # held to an absolute bound, reported beside the relative one of the other 41 features
            ppm_cols = [8, 9, 41, 42, 45]
            keep = [f for f in range(46) if f not in ppm_cols]
            dp = np.abs(fe[:, ppm_cols].astype(np.float64) - fg[:, ppm_cols])
            dp = np.where(np.isnan(fe[:, ppm_cols]) & np.isnan(fg[:, ppm_cols]), 0.0, dp)
            max_ppm = float(np.nanmax(dp))
            fe, fg = fe[:, keep], fg[:, keep]
            d = np.abs(fe.astype(np.float64) - fg) / np.maximum(np.maximum(np.abs(fe), np.abs(fg)), 1e-6)
            d = np.where(np.isnan(fe) & np.isnan(fg), 0.0, d)
            max_rel = float(np.nanmax(d))
        result["cpu_baseline"] = {
            "value": cpu_prec / dt,
            "unit": "precursors/s",
            "cores": cores,
            "cpu_quota_cores": quota,
            "kind": "port",
            "region": "host candidate SoA -> host OutputPsmDF SoA (the region `value` is timed on)",
            "sample": f"first {sample} candidates ({cpu_prec} precursors) of the same batch, "
                      f"{reps} x {dt:.2f}s, {cores} OpenMP threads (the faster of the sustained runs; the host "
                      f"shows {ncpu} hardware threads, the container's CFS quota is "
                      f"{'unlimited' if not quota else f'{quota:g} cores'})",
            "precursors_per_s_1_thread": rate_1 * cpu_prec / sample,
            "sustained_candidates_per_s_by_threads": {str(k): v for k, v in sustained.items()},
            "burst_candidates_per_s_by_threads": {str(k): v for k, v in tried.items()},
            "valid_identical_to_gpu": same_valid,
            "max_rel_feature_diff_vs_gpu": max_rel,
            "max_abs_ppm_feature_diff_vs_gpu": max_ppm,
            "feature_diff_note": "relative difference over the 41 non-ppm features; the 5 mass-error features "
                                 "(columns 8, 9, 41, 42, 45) as an absolute difference in ppm",
        }
        result["config"]["gpu_over_cpu"] = value / (cpu_prec / dt)

    # ---------------- the other kernel families of the path, each with roofline + cpu_baseline ----------------
    if extras:
        from tools import bench_legs

        t_legs = time.time()
        legs = {}
        try:  # the operator on the headline table (the run and the library are still staged)
            legs["operator"] = bench_legs.operator_leg(case, cfg)
            result["operator_ms"] = legs["operator"]["operator_ms"]
        except Exception as exc:
            legs["operator"] = {"skipped": f"{type(exc).__name__}: {exc}"[:300]}
            log(f"[bench] leg operator skipped: {exc}")
        # free the headline's host arrays (3.9 GB of peaks + page-locked tables) before the other legs
        del host, features_last, packed, soa, soa_all, per_cand_bytes, matched
        case = None
        for name, fn in (
            ("fragment_competition", lambda: bench_legs.fragcomp_leg(ctx)),
            ("multiplex_configs4", lambda: bench_legs.multiplex_leg(ctx, threads=os.cpu_count() or 8,
                                                                    cpu_seconds=min(args.cpu_seconds, 6.0))),
            ("transfer_requant", lambda: bench_legs.transfer_requant_leg(ctx, threads=os.cpu_count() or 8,
                                                                         cpu_seconds=min(args.cpu_seconds, 4.0))),
            ("ion_mobility_configs3", lambda: bench_legs.timstof_leg(full_size=True)),
            ("candidate_selection", lambda: bench_legs.selection_leg()),
            ("fdr_stage", lambda: bench_legs.fdr_leg()),
        ):
            if time.time() - t_legs > args.extras_seconds:
                legs[name] = {"skipped": f"the legs' budget of {args.extras_seconds:.0f} s was spent"}
                continue
            try:
                t0 = time.time()
                legs[name] = fn()
                legs[name]["leg_seconds"] = time.time() - t0
            except Exception as exc:  # the metric does not depend on these legs
                legs[name] = {"skipped": f"{type(exc).__name__}: {exc}"[:300]}
                log(f"[bench] leg {name} skipped: {exc}")
        result["extras"] = legs
        # the legs' key numbers once more as flat keys (a reader that keeps only the top of the record sees them)
        def pick(leg, *path):
            v = legs.get(leg)
            for k in path:
                v = v.get(k) if isinstance(v, dict) else None
            return v

        result["config"]["legs"] = {
            "operator_ms": pick("operator", "operator_ms"),
            "operator_wire_bytes": pick("operator", "stages_ms", "wire_bytes"),
            "fragcomp_1e6_kernel_ms": pick("fragment_competition", "1000000", "kernel_ms"),
            "fragcomp_1e6_call_ms": pick("fragment_competition", "1000000", "host_to_host_ms"),
            "fragcomp_1e6_operator_ms": pick("fragment_competition", "1000000", "operator_ms"),
            "fragcomp_1e6_frac": pick("fragment_competition", "1000000", "roofline", "frac"),
            "configs4_ms": pick("multiplex_configs4", "ms_per_step"),
            "configs4_kernel_ms": pick("multiplex_configs4", "kernel_ms"),
            "configs4_frac": pick("multiplex_configs4", "roofline", "frac"),
            "transfer_requant_ms": pick("transfer_requant", "ms_per_step"),
            "transfer_requant_kernel_ns_per_candidate": pick("transfer_requant", "kernel_ns_per_candidate"),
            "transfer_requant_frac": pick("transfer_requant", "roofline", "frac"),
            "configs3_ms": pick("ion_mobility_configs3", "ms_per_step"),
            "configs3_kernel_ms": pick("ion_mobility_configs3", "roofline", "kernel_ms"),
            "configs3_frac": pick("ion_mobility_configs3", "roofline", "frac"),
            "configs3_selection_kernel_ms": pick("ion_mobility_configs3", "selection", "kernel_ms"),
            "configs3_selection_frac": pick("ion_mobility_configs3", "selection", "roofline", "frac"),
            "selection_kernel_ms": pick("candidate_selection", "kernel_ms"),
            "selection_frac": pick("candidate_selection", "roofline", "frac"),
            "fdr_fit_kernels_ms": pick("fdr_stage", "fit_kernels_ms"),
            "fdr_predict_kernel_ms": pick("fdr_stage", "predict_kernel_ms"),
        }

    if rank == 0:
        print(json.dumps(result))
    if with_comm:
        ctx.comm_wait()
        ctx.comm_destroy()


if __name__ == "__main__":
    main()
