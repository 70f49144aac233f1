"""Benchmark of the candidate-scoring hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): precursors scored / s.  A "step" is one pass of the hot
path over the rank's candidate batch: zero the output tables, run the scoring
kernel on the candidate table that is already resident in HBM and - for N > 1 -
reassemble the tables with ONE RCCL all-gather.  Workload at N = 1 is
BASELINE.json configs[1] (100k-precursor predicted library x 3 candidates vs the
2 h synthetic run T120); with N GPUs every rank keeps that amount of work
(weak scaling: N x 100k precursors, run and library replicated per GPU).

The JSON line also carries
  roofline     - algorithmic bytes (SURVEY.md section 8d formula) / average kernel
                 duration (HIP events on the launch stream) vs the 8 TB/s HBM peak
  cpu_baseline - the CPU oracle (oracle/, a C++ restatement of the reference's
                 Numba path) timed on this host's cores on a bounded sample of the
                 same candidates (rank 0, N = 1 only).  Checker code is used here
                 only as the thing timed beside the GPU, never in the GPU path.
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

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(dia, soa, cfg, matched_peaks, lib_slice_len):
    """bytes the reference algorithm has to touch per candidate.

    B = sum_probes [4 * (ceil(log2 P_s) + 1)] + 8 * matched_peaks + 18 * K_lib + 64
        + (46 * 4 + K * 38 + 6)
    with one probe per (fragment, observation, cycle) in MS2 and per (isotope, cycle)
    in MS1 and P_s the number of peaks of the probed spectrum.  Auxiliary index reads
    of this implementation are not counted.
    """
    L = dia.cycle_len
    n = len(soa["precursor_idx"])
    counts = (dia.peak_stop_idx_list - dia.peak_start_idx_list).astype(np.int64)
    steps = (np.ceil(np.log2(np.maximum(counts, 1))).astype(np.int64) + 1) * 4  # bytes per probe
    steps_2d = steps.reshape(-1, L)  # [cycle, position]
    csum = np.concatenate([np.zeros((1, L), np.int64), np.cumsum(steps_2d, axis=0)], axis=0)
    c0 = soa["frame_start"] // L
    c1 = soa["frame_stop"] // L
    K = np.minimum(lib_slice_len, int(cfg.top_k_fragments)).astype(np.int64)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precursors-per-gpu", type=int, default=100_000)
    ap.add_argument("--cycles", type=int, default=4800)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true",
                    help="debug: every rank uses cuda:0 (smoke-test the N>1 path on a 1-GPU box; "
                         "needs --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from alphadia_amd import runtime, synthetic as syn
    from alphadia_amd.distributed import DeviceTables, all_gather_tables, shard_bounds, slice_soa
    from alphadia_amd.scoring import (
        CandidateScoringConfig,
        assemble_candidates,
        fragment_columns,
        pack_assembled,
    )

    # ---------------- workload: configs[1] per GPU ----------------
    n_prec_total = args.precursors_per_gpu * world
    t0 = time.time()
    threads = max(1, (os.cpu_count() or 8) // max(world, 1))
    case = syn.make_case(n_prec_total, args.cycles, config_id=2, per_precursor=3, threads=threads)
    log(f"[bench] synthetic run: {case.dia.n_spectra} spectra, {case.dia.mz_values.size/1e6:.1f}M peaks, "
        f"{len(case.candidates_df)} candidates, generated in {time.time()-t0:.1f}s ({threads} threads)")
    if os.environ.get("ADH_BENCH_DEBUG"):
        print(f"[debug] rank {rank} mz checksum {float(case.dia.mz_values[::1000].astype(np.float64).sum())} "
              f"int checksum {float(case.dia.intensity_values[::1000].astype(np.float64).sum())} "
              f"cand checksum {int(case.candidates_df['frame_start'].sum())}", file=sys.stderr, flush=True)
    cfg = CandidateScoringConfig()
    # ClassicExtractionHandler defaults (extraction_handler.py:370-376,400-409; default.yaml:158-199)
    cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1,
                    precursor_mz_tolerance=10, fragment_mz_tolerance=15, exclude_shared_ions=True,
                    quant_window=3, quant_all=True, experimental_xic=True, top_k_fragments=12))
    cfgj = cfg.to_jitclass()
    soa_all = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    a, b = shard_bounds(soa_all["score_group_idx"], rank, world)
    soa = slice_soa(soa_all, a, b)
    n_local = b - a
    n_prec_local = len(np.unique(soa["precursor_idx"]))

    # ---------------- staging (one-time, excluded from the metric) ----------------
    ctx = runtime.get_context(local_rank)
    t0 = time.time()
    ctx.stage_run(case.dia)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
    t_stage = time.time() - t0
    t0 = time.time()
    ctx.upload_candidates(pack_assembled(soa))
    t_upload = time.time() - t0
    log(f"[bench] staged run+library in {t_stage:.2f}s, candidates uploaded in {t_upload:.3f}s")

    n_rows = -(-len(soa_all["precursor_idx"]) // world)  # pad to the largest shard
    # one explicit (non-default) torch stream carries the memset and the kernels; with N > 1 the
    # all-gather of step i (RCCL, its own stream) overlaps the kernels of step i+1 through two
    # table buffers (alphadia_amd.distributed.PipelinedGather)
    from alphadia_amd.distributed import PipelinedGather

    pg = PipelinedGather(n_rows, int(cfgj.top_k_fragments), device, world, with_stats=True)
    if os.environ.get("ADH_BENCH_NO_OVERLAP"):
        pg.overlap = False
    outs = [t.as_output(n_local) for t in pg.tables]
    work_stream = torch.cuda.Stream(device=device)
    stream = work_stream.cuda_stream

    def step():
        with torch.cuda.stream(work_stream):
            tables = pg.begin()
            tables.zero_()
            ctx.score_uploaded(cfgj, outs[pg.slot], stream)
            pg.end()

    def fence():
        with torch.cuda.stream(work_stream):
            out = pg.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return out

    for _ in range(args.warmup):
        step()
    fence()
    ctx.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gathered = fence()
    elapsed = time.perf_counter() - t0
    tables = pg.tables[pg.slot]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gather_ms, feature_ms, launches = ctx.kernel_time_ms(reset=True)
    kernel_ms = gather_ms + feature_ms

    # ---------------- results of the last step (sanity + roofline inputs) ----------------
    host = tables.to_host()
    valid = host["valid"][:n_local].astype(bool)
    print(f"[bench] rank {rank}: {int(valid.sum())}/{n_local} candidates valid", file=sys.stderr, flush=True)
    matched = host["stat_matched_peaks"][:n_local]
    if world > 1:
        # every rank must now hold every rank's tables: compare the valid counts
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        counts[rank] = int(valid.sum())
        dist.all_reduce(counts)
        got = [int(tables.to_host(gathered[r])["valid"].sum()) for r in range(world)]
        assert got == [int(c) for c in counts.tolist()], (got, counts.tolist())

    lib_len = (soa["frag_stop_idx"].astype(np.int64) - soa["frag_start_idx"].astype(np.int64))
    per_cand_bytes = algorithmic_bytes(case.dia, soa, cfgj, matched, lib_len)
    bytes_per_launch = float(per_cand_bytes.sum())
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    total_prec = n_prec_total
    value = total_prec * args.steps / elapsed
    result = {
        "metric": "precursors scored/sec",
        "value": value,
        "unit": "precursors/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32/f64",
        "data": "synthetic",
        "config": {
            "workload": f"configs[1] per GPU: {args.precursors_per_gpu // 1000}k-precursor predicted library x 3 "
                        f"candidates vs {'2h ' if args.cycles == 4800 else ''}synthetic Thermo-style DIA run "
                        f"({args.cycles} cycles x 61 spectra)",
            "precursors_total": total_prec,
            "candidates_total": int(len(soa_all["precursor_idx"])),
            "candidates_per_gpu": int(n_local),
            "cycles": args.cycles,
            "peaks": int(case.dia.mz_values.size),
            "parallelism": (f"candidate-sharded x{world}, tables all-gathered "
                            f"({'overlapped with the next step' if pg.overlap else 'synchronous'})")
            if world > 1 else "single GPU",
            "valid_fraction": float(valid.mean()) if n_local else 0.0,
            "candidates_per_s": float(len(soa_all["precursor_idx"]) * args.steps / elapsed),
            "gathered_bytes_per_candidate": pg.tables[0].wire_nbytes / max(n_rows, 1),
            "stage_seconds": t_stage,
            "candidate_upload_seconds": t_upload,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel": "adh_gather_kernel + feature kernels (adh_feature_fast_kernel<FM,NO>, adh_feature_kernel): the hot path, sum of both",
            "kernel_ms": kernel_ms,
            "gather_kernel_ms": gather_ms,
            "feature_kernel_ms": feature_ms,
            "launches": int(launches),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "algorithmic_bytes_per_candidate": bytes_per_launch / max(n_local, 1),
        },
    }
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            if tr.get("candidates_per_gpu") == int(n_local):
                result["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
        except Exception:
            pass

    # ---------------- the same batch through host buffers (PCIe both ways): reported, never `value` ----
    if rank == 0 and world == 1:
        try:
            packed_all = pack_assembled(soa)
            ctx.score_host(packed_all, cfgj)  # first call builds the plan
            t0 = time.perf_counter()
            reps_h = 3
            for _ in range(reps_h):
                ctx.score_host(packed_all, cfgj)
            th = (time.perf_counter() - t0) / reps_h
            result["config"]["host_buffers_ms_per_step"] = th * 1e3
            result["config"]["host_buffers_precursors_per_s"] = n_prec_local / th
        except Exception as exc:  # the metric does not depend on this leg
            log(f"[bench] host-buffer leg skipped: {exc}")

    # ---------------- the step before scoring (candidate selection), reported next to the metric ----
    if rank == 0 and world == 1:
        try:
            from alphadia_amd import _abi
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

    # ---------------- CPU baseline: the oracle on this host's cores ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle

        ncpu = os.cpu_count() or 1
        cols = fragment_columns(case.library.fragment_df, "mz_library")
        # pick the thread count that is fastest on this host (memory-latency bound: more
        # threads than memory channels x a few does not help)
        best = (0.0, 1)
        for th in sorted({t for t in (16, 32, 64, 128, ncpu) if t <= ncpu} | {min(8, ncpu)}):
            probe = min(400 * th, n_local)
            packed = pack_assembled(slice_soa(soa, 0, probe))
            oracle.score(case.dia, cols, packed, cfgj, n_threads=th)
            t0 = time.perf_counter()
            oracle.score(case.dia, cols, packed, cfgj, n_threads=th, reuse=oracle.score.last_buffers)
            rate = probe / (time.perf_counter() - t0)
            log(f"[bench] cpu oracle {th:4d} threads: {rate:,.0f} candidates/s")
            if rate > best[0]:
                best = (rate, th)
        rate, cores = best
        # single-thread rate (the reference's pjit loop at thread_count = 1), short sample
        probe1 = min(1500, n_local)
        packed1 = pack_assembled(slice_soa(soa, 0, probe1))
        oracle.score(case.dia, cols, packed1, cfgj, n_threads=1)
        t0 = time.perf_counter()
        oracle.score(case.dia, cols, packed1, cfgj, n_threads=1, reuse=oracle.score.last_buffers)
        rate_1 = probe1 / (time.perf_counter() - t0)
        all_rate = None
        sample = int(min(n_local, max(2000, rate * args.cpu_seconds)))
        sub = slice_soa(soa, 0, sample)
        packed = pack_assembled(sub)
        oracle.score(case.dia, cols, packed, cfgj, n_threads=cores)  # touches the output pages
        reps, dt = 0, 0.0
        while dt < min(args.cpu_seconds, 10.0) or reps == 0:  # repeat the sample for a stable rate
            t0 = time.perf_counter()
            exp = oracle.score(case.dia, cols, packed, cfgj, n_threads=cores, reuse=oracle.score.last_buffers)
            dt += time.perf_counter() - t0
            reps += 1
        dt /= reps
        cpu_prec = len(np.unique(sub["precursor_idx"]))
        same_valid = bool(np.array_equal(exp["valid"].astype(bool), valid[:sample]))
        fe, fg = exp["features"][exp["valid"].astype(bool)], host["features"][:sample][valid[:sample]] if same_valid else None
        max_rel = None
        if same_valid and fe.size:
            keep = [f for f in range(46) if f not in (8, 9, 41, 42, 45)]  # ppm errors: absolute metric
            fe, fg = fe[:, keep], fg[:, keep]
            d = np.abs(fe.astype(np.float64) - fg) / np.maximum(np.maximum(np.abs(fe), np.abs(fg)), 1e-6)
            d = np.where(np.isnan(fe) & np.isnan(fg), 0.0, d)
            max_rel = float(np.nanmax(d))
        result["cpu_baseline"] = {
            "value": cpu_prec / dt,
            "unit": "precursors/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {sample} candidates ({cpu_prec} precursors) of the same batch, "
                      f"{reps} x {dt:.2f}s, OpenMP over {cores} threads (fastest of the thread counts tried "
                      f"on this {ncpu}-thread host)",
            "candidates_per_s_1_thread": rate_1,
            "valid_identical_to_gpu": same_valid,
            "max_rel_feature_diff_vs_gpu": max_rel,
        }
        result["config"]["gpu_over_cpu"] = value / (cpu_prec / dt)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
