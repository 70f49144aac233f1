import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))  # synthetic data generators
import synthetic as syn
from alphadia_amd import runtime
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled
case = syn.make_case(1_000_000, 4800, config_id=2, per_precursor=3, threads=os.cpu_count())
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
def run(tag, k):
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); ctx.score_host(packed, cfgj, reuse_buffers=True); ts.append((time.perf_counter()-t0)*1e3)
    print(tag, " ".join(f"{t:.1f}" for t in ts), flush=True)
run("cold   ", 8)
time.sleep(3.0)
run("after 3 s idle", 6)
time.sleep(0.3)
run("after 0.3 s idle", 4)
# kernels only for 0.5 s, then host->host
ctx.upload_candidates(packed); view = ctx.device_tables(); st = ctx.stream_handle()
time.sleep(3.0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    ctx.score_uploaded(cfgj, view, st); ctx.synchronize()
run("after 3 s idle + 0.5 s of kernels", 5)
