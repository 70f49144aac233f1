"""The shard of an eight-GPU run on one GPU: the 3 M-row headline table cut with ``shard_bounds(., r, 8)``, each of
the eight shards scored host -> host with a one-rank RCCL communicator attached (so the padded table layout, the
double-buffered slots and the all-gather enqueue of the multi-GPU path are inside the timed call).

``shard_leg`` is the `shard_8way` leg of bench.py; run as a script it prints the same record (and, with
ADH_DEBUG_TIMING=2 in the environment, the library's per-chunk copy-out timeline of every call)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def shard_leg(ctx, soa_all: dict, cfgj, ms_per_step: float | None = None, ways: int = 8, reps: int = 7,
              with_comm: bool = True) -> dict:
    """Host -> host time of every ``ways``-way shard of the assembled table on this GPU."""
    from alphadia_amd import _abi
    from alphadia_amd.distributed import shard_bounds, slice_soa
    from alphadia_amd.scoring import pack_assembled

    n_all = len(soa_all["precursor_idx"])
    bounds = [shard_bounds(soa_all["score_group_idx"], r, ways) for r in range(ways)]
    max_rows = max(b - a for a, b in bounds)
    attached = False
    if with_comm and not getattr(ctx, "_comm_attached", False):
        ctx.comm_init(0, 1, max_rows)
        attached = True
    from alphadia_amd import runtime

    # CONTENDED pass: eight ranks of one node share ONE CPU quota, so each rank's host team is
    # the eighth part of it (what adh_host_threads gives under LOCAL_WORLD_SIZE = 8) - the first pass, with this
    # process's whole budget, is the optimistic figure
    budget = runtime.host_threads(1 << 40)[1]
    contended_threads = max(1, budget // ways)

    def one_pass(host_threads: int | None, force_rebuild: bool = False):
        out = []
        old = os.environ.get("ADH_HOST_THREADS")
        if host_threads is not None:
            os.environ["ADH_HOST_THREADS"] = str(host_threads)
        if force_rebuild:
            os.environ["ADH_REBUILD_MIN_THREADS"] = "0"
        try:
            for r, (a, b) in enumerate(bounds):
                sub = slice_soa(soa_all, a, b)
                sub = {k: (ctx.pinned.take("shard:" + k, v) if isinstance(v, np.ndarray) and v.shape[:1] == (b - a,) else v)
                       for k, v in sub.items()}
                packed = pack_assembled(sub)
                for _ in range(3):
                    ctx.score_host(packed, cfgj, reuse_buffers=True)
                ctx.comm_wait()
                ctx.device_synchronize()
                ctx.kernel_time_ms(reset=True)
                d2h0 = ctx.d2h_bytes(reset=True)
                del d2h0
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    ctx.score_host(packed, cfgj, reuse_buffers=True)
                    ts.append((time.perf_counter() - t0) * 1e3)
                ctx.comm_wait()
                ctx.device_synchronize()
                g, f, nl = ctx.kernel_time_ms(reset=True)
                wire = ctx.d2h_bytes(reset=True) / reps
                out.append({"rank": r, "rows": int(b - a), "ms": float(np.median(ts)), "min_ms": float(min(ts)),
                            "kernel_ms": float((g + f) * nl / reps), "launches": float(nl / reps),
                            "d2h_bytes": float(wire), "d2h_ms_at_55GBps": float(wire / 55e9 * 1e3)})
        finally:
            os.environ.pop("ADH_REBUILD_MIN_THREADS", None)
            if host_threads is not None:
                if old is None:
                    os.environ.pop("ADH_HOST_THREADS", None)
                else:
                    os.environ["ADH_HOST_THREADS"] = old
        return out

    try:
        per_shard = one_pass(None)
        contended = one_pass(contended_threads)                       # (the library's policy: few threads -> the link carries all columns)
        contended_rb = one_pass(contended_threads, force_rebuild=True)  # (the same team made to rebuild the columns)
    finally:
        if attached:
            ctx.comm_wait()
            ctx.comm_destroy()
    worst = max(per_shard, key=lambda s: s["ms"])
    rec = {
        "ways": ways,
        "rows_total": int(n_all),
        "with_one_rank_communicator": bool(with_comm),
        "max_shard_ms": worst["ms"],
        "median_shard_ms": float(np.median([s["ms"] for s in per_shard])),
        "kernel_ms": worst["kernel_ms"],
        "d2h_ms": worst["d2h_ms_at_55GBps"],
        "shards": per_shard,
        "note": "every shard scored host -> host on ONE GPU, one after the other; max_shard_ms is what an "
                f"{ways}-GPU step would wait for before the all-gather costs anything (the gather is enqueued, one rank)",
    }
    worst_c = max(contended, key=lambda s: s["ms"])
    # the all-gather this leg does not time: every rank contributes its wire tables and receives the other ways - 1
    # shards; point-to-point xGMI (7 links per GPU, ~153 GB/s each in both directions = ~76 GB/s in): a direct
    # all-gather moves one shard per link, side by side
    shard_bytes = max(s["d2h_bytes"] for s in per_shard)
    rec["contended"] = {
        "host_threads_per_rank": int(contended_threads), "cpu_budget": int(budget),
        "max_shard_ms": worst_c["ms"], "median_shard_ms": float(np.median([s["ms"] for s in contended])),
        "shard_ms": [round(s["ms"], 3) for s in contended],
        "d2h_bytes": max(s["d2h_bytes"] for s in contended),
        "policy": "below 6 host threads per rank the device writes the id / library columns and the link carries all "
                  "646 bytes per candidate (host_rebuild_pays, adh_score_host.hip)",
        "max_shard_ms_if_the_team_rebuilds": max(s["ms"] for s in contended_rb),
    }
    rec["all_gather"] = {
        "bytes_contributed_per_rank": float(shard_bytes),
        "bytes_received_per_rank": float(shard_bytes * (ways - 1)),
        "ms_at_xgmi_link_rate_76GBps": float(shard_bytes / 76e9 * 1e3),
        "ms_at_70pct_of_link_rate": float(shard_bytes / (0.7 * 76e9) * 1e3),
        "note": "estimate, not measured: enqueued after a rank's last chunk, it overlaps that rank's remaining "
                "copy-out and the next step; exposed only if longer than those",
    }
    if ms_per_step:
        rec["ms_per_step_1gpu"] = float(ms_per_step)
        rec["projected_scaling_8"] = float(ms_per_step / worst["ms"])
        rec["projected_scaling_8_contended"] = float(ms_per_step / worst_c["ms"])
        rec["projected_scaling_8_contended_rebuild"] = float(ms_per_step / max(s["ms"] for s in contended_rb))
    return rec


def sweep(ctx, soa, cfgj, settings, sizes=(24_000, 48_000, 96_000, 192_000, 375_000, 750_000, 1_500_000), reps=7):
    """Developer sweep: host -> host ms of the first `size` rows under each environment setting (A/B in one process;
    the library reads its pipeline switches per call)."""
    from alphadia_amd.distributed import slice_soa
    from alphadia_amd.scoring import pack_assembled

    keys = ("ADH_CHUNK_PARTS", "ADH_CHUNK_MIN", "ADH_FIRST_CHUNK_DIV", "ADH_CHUNK", "ADH_H2D_BURST_LATE", "ADH_FUSED_STREAMS", "ADH_ANY_ORDER")
    for size in sizes:
        m = min(size, len(soa["precursor_idx"]))
        sub = slice_soa(soa, 0, m)
        sub = {k: (ctx.pinned.take("sweep:" + k, v) if isinstance(v, np.ndarray) and v.shape[:1] == (m,) else v)
               for k, v in sub.items()}
        packed = pack_assembled(sub)
        for name, env in settings:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            for _ in range(2):
                ctx.score_host(packed, cfgj, reuse_buffers=True)
            ctx.kernel_time_ms(reset=True)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                ctx.score_host(packed, cfgj, reuse_buffers=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            g, f, nl = ctx.kernel_time_ms(reset=True)
            print(f"[sweep] rows {m:8d} {name:28s}: median {np.median(ts):6.2f} ms  min {min(ts):6.2f}  "
                  f"kernels {(g + f) * nl / reps:5.2f} ms in {nl / reps:.0f} launches", file=sys.stderr, flush=True)
    for k in keys:
        os.environ.pop(k, None)


def main():
    import synthetic as syn
    from alphadia_amd import runtime
    from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled

    n_prec = int(os.environ.get("N_PREC", 1_000_000))
    case = syn.make_case(n_prec, 4800, config_id=2, per_precursor=3, threads=os.cpu_count())
    cfg = CandidateScoringConfig()
    cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1, precursor_mz_tolerance=10,
                    fragment_mz_tolerance=15, exclude_shared_ions=True, quant_window=3, quant_all=True,
                    experimental_xic=True, top_k_fragments=12))
    cfgj = cfg.to_jitclass()
    ctx = runtime.get_context(0)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library", pool=ctx.pinned)
    ctx.stage_run(case.dia)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
    packed = pack_assembled(soa)
    full = None
    if os.environ.get("SWEEP"):
        # SWEEP="name:KEY=V,KEY=V;name2:..."  (an empty setting = the defaults)
        settings = []
        for item in os.environ["SWEEP"].split(";"):
            name, _, kv = item.partition(":")
            settings.append((name, dict(x.split("=") for x in kv.split(",") if x)))
        for _ in range(5):
            ctx.score_host(packed, cfgj, reuse_buffers=True)
        sizes = [int(x) for x in os.environ.get("SIZES", "24000,48000,96000,192000,375000,750000,1500000,3000000").split(",")]
        sweep(ctx, soa, cfgj, settings, sizes=sizes)
        return
    if not os.environ.get("SHARD_ONLY"):
        for _ in range(5):
            ctx.score_host(packed, cfgj, reuse_buffers=True)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            ctx.score_host(packed, cfgj, reuse_buffers=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        full = float(np.median(ts))
        print(f"[shard] full table: {full:.2f} ms", file=sys.stderr, flush=True)
    rec = shard_leg(ctx, soa, cfgj, ms_per_step=full, ways=int(os.environ.get("WAYS", 8)),
                    reps=int(os.environ.get("REPS", 7)), with_comm=not os.environ.get("NO_COMM"))
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
