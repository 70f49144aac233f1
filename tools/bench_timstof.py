"""Throughput of the ion-mobility path (BASELINE config 4 shape at reduced scale).

Not the driver's bench (bench.py measures configs[1]); this records how the timsTOF-style
path performs next to the CPU oracle on the same inputs.  Run on the GPU box from the repo root.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from alphadia_amd import runtime, synthetic as syn  # noqa: E402
from alphadia_amd.distributed import DeviceTables, slice_soa  # noqa: E402
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled  # noqa: E402
from oracle import oracle  # noqa: E402

n_prec = int(os.environ.get("N_PREC", 20000))
t0 = time.time()
# BASELINE config 4 at full scale: N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30
case = syn.make_timstof_case(
    n_precursors=n_prec, n_cycles=int(os.environ.get("N_CYCLES", 300)), config_id=4, per_precursor=3, n_ms2_frames=8,
    windows_per_frame=3, scan_max_index=int(os.environ.get("SCAN_MAX", 256)), n_tof=int(os.environ.get("N_TOF", 200000)),
    events_per_push=float(os.environ.get("EVENTS_PER_PUSH", 25.0)), mz_lo=400.0,
    mz_hi=1000.0, frag_mz_lo=200.0, frag_mz_hi=1000.0, tof_mz_lo=195.0, tof_mz_hi=1010.0,
    planted_fraction=0.02,
)
print(f"generated {case.dia.push_indices.size/1e6:.1f}M events in {time.time()-t0:.1f}s", file=sys.stderr)
cfg = CandidateScoringConfig()
cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                experimental_xic=True))
cfgj = cfg.to_jitclass()
soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
n = len(soa["precursor_idx"])
ctx = runtime.get_context(0)
ctx.stage_run(case.dia)
cols = fragment_columns(case.library.fragment_df, "mz_library")
ctx.stage_fragments(*cols)
ctx.upload_candidates(pack_assembled(soa))
dev = torch.device("cuda", 0)
tables = DeviceTables(n, 12, dev, with_stats=True)
out = tables.as_output(n)
ws = torch.cuda.Stream(device=dev)
steps = 5
with torch.cuda.stream(ws):
    for it in range(steps + 1):
        if it == 1:
            torch.cuda.synchronize()
            ctx.kernel_time_ms(reset=True)
            t0 = time.perf_counter()
        tables.zero_()
        ctx.score_uploaded(cfgj, out, ws.cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
g_ms, f_ms, _ = ctx.kernel_time_ms(reset=True)
host = tables.to_host()
if os.environ.get("ADH_BENCH_NO_CPU"):
    print(json.dumps({"gather_kernel_ms": g_ms, "feature_kernel_ms": f_ms, "ms_per_step": dt * 1e3}))
    sys.exit(0)
cores = min(64, os.cpu_count() or 1)
sample = min(n, 6000)
sub = slice_soa(soa, 0, sample)
oracle.score_timstof(case.dia, cols, pack_assembled(slice_soa(soa, 0, 500)), cfgj, n_threads=cores)
t0 = time.perf_counter()
exp = oracle.score_timstof(case.dia, cols, pack_assembled(sub), cfgj, n_threads=cores, with_stats=True)
cdt = time.perf_counter() - t0
same = bool(np.array_equal(exp["valid"], host["valid"][:sample])
            and np.array_equal(exp["stat_matched_peaks"], host["stat_matched_peaks"][:sample]))
F = (soa["frame_stop"] - soa["frame_start"]) // case.dia.cycle_len
S = soa["scan_stop"] - soa["scan_start"]
print(json.dumps({
    "workload": f"timsTOF-style synthetic run: {case.dia.push_indices.size/1e6:.1f}M events, "
                f"{int(case.dia.scan_max_index)} scans, {int(case.dia.cycle.shape[1])} frames/cycle, "
                f"{int(os.environ.get('N_CYCLES', 300))} cycles; {n_prec} precursors x 3 candidates, "
                f"S in [{int(S.min())},{int(S.max())}], F in [{int(F.min())},{int(F.max())}]",
    "candidates": n, "precursors_per_s": n_prec / dt, "candidates_per_s": n / dt, "ms_per_step": dt * 1e3,
    "gather_kernel_ms": g_ms, "feature_kernel_ms": f_ms, "valid_fraction": float(host["valid"].mean()),
    "cpu_oracle": {"candidates_per_s": sample / cdt, "threads": cores, "sample": sample,
                   "valid_and_matched_peaks_identical_to_gpu": same},
}))
