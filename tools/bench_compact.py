"""Developer timing of adh_score_candidates_compact next to adh_score_candidates on the headline table
(ADH_COPY_OUT_BLOCKS / ADH_HOST_THREADS sweeps in one process)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synthetic as syn  # noqa: E402
from alphadia_amd import runtime  # noqa: E402
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled  # noqa: E402

case = syn.make_case(int(os.environ.get("N_PREC", 1_000_000)), 4800, config_id=2, per_precursor=3, threads=os.cpu_count())
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
for _ in range(4):
    ctx.score_host(packed, cfgj, reuse_buffers=True)


def timed(fn, reps=6):
    fn()
    ctx.kernel_time_ms(reset=True)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    g, f, nl = ctx.kernel_time_ms(reset=True)
    return np.median(ts), min(ts), (g + f) * nl / reps


print("padded  : median %.2f min %.2f kernels %.2f" % timed(lambda: ctx.score_host(packed, cfgj, reuse_buffers=True)), flush=True)
settings = [{}] + [dict(x.split("=") for x in item.split(",") if x) for item in os.environ.get("SWEEP", "").split(";") if item]
for env in settings:
    for k in ("ADH_COPY_OUT_BLOCKS", "ADH_HOST_THREADS", "ADH_CHUNK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    print("compact %s: median %.2f min %.2f kernels %.2f" % ((env,) + timed(lambda: ctx.score_host_compact(packed, cfgj))), flush=True)
