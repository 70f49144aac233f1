"""Host -> host scoring step (adh_score_candidates) under different pipeline settings, A/B inside one
process: ADH_CHUNK sweep, repeated so that box noise (shared PCIe / host memory) shows up as spread."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # synthetic data generators
import synthetic as syn
from alphadia_amd import runtime  # noqa: E402
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled  # noqa: E402

N_PREC = int(os.environ.get("N_PREC", 1_000_000))
case = syn.make_case(N_PREC, 4800, config_id=2, per_precursor=3, threads=os.cpu_count())
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
for _ in range(3):
    ctx.score_host(packed, cfgj, reuse_buffers=True)
chunks = [int(x) for x in os.environ.get("CHUNKS", "131072,262144,524288,1048576").split(",")]
for rep in range(int(os.environ.get("REPS", 3))):
    for ch in chunks:
        os.environ["ADH_CHUNK"] = str(ch)
        ctx.score_host(packed, cfgj, reuse_buffers=True)
        ctx.kernel_time_ms(reset=True)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            ctx.score_host(packed, cfgj, reuse_buffers=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        g, f, n = ctx.kernel_time_ms(reset=True)
        print(f"rep {rep} chunk {ch:8d}: median {np.median(ts):6.2f} ms  min {min(ts):6.2f}  max {max(ts):6.2f}  "
              f"kernels/step {(g + f) * n / 8:6.2f} ms", flush=True)
