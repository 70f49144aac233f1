"""Scratch: where FragmentCompetition.__call__ spends its time on the 1e6-PSM table of the bench leg
(the C call adh_fragcomp_frames, the frame of the survivors, the candidate keys).  GPU box, repo root."""
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import synthetic as syn  # noqa: E402
from alphadia_amd import runtime  # noqa: E402
from alphadia_amd.fragcomp import FragmentCompetition, candidate_hash  # noqa: E402

n = int(os.environ.get("N_PSM", 1_000_000))
t = syn.make_competition_table(n, seed=7)
k = int(t["k"])
rng = np.random.default_rng(5)
cyc = syn.make_cycle(n_ms2=int(t["n_windows"]), mz_lo=400.0, mz_hi=1000.0)
win = np.repeat(np.arange(int(t["n_windows"])), (t["window_stop"] - t["window_start"]))
lo_w, hi_w = cyc[0, 1:, 0, 0], cyc[0, 1:, 0, 1]
mz_obs = (lo_w[win] + (hi_w[win] - lo_w[win]) * rng.random(n)).astype(np.float32)
pidx = rng.permutation(n).astype(np.uint32)
psm_df = pd.DataFrame({"precursor_idx": pidx, "rank": np.zeros(n, np.uint8), "mz_observed": mz_obs,
                       "rt_observed": t["rt"], "proba": rng.random(n).astype(np.float32)})
frag_df = pd.DataFrame({"precursor_idx": np.repeat(pidx, k), "rank": np.zeros(n * k, np.uint8), "mz_observed": t["mz"]})
ctx = runtime.get_context(None)
fc = FragmentCompetition()
fc(psm_df, frag_df, cyc)
for rep in range(3):
    t0 = time.perf_counter()
    cols = (psm_df["precursor_idx"].values, psm_df["rank"].values, psm_df["mz_observed"].values,
            psm_df["rt_observed"].values, psm_df["proba"].values, frag_df["precursor_idx"].values,
            frag_df["rank"].values, frag_df["mz_observed"].values)
    t1 = time.perf_counter()
    rows, valid = ctx.fragcomp_frames(*cols, cyc, 3, 15)
    t2 = time.perf_counter()
    kept = rows[valid]
    t3 = time.perf_counter()
    out = psm_df.iloc[kept].copy()
    t4 = time.perf_counter()
    out["_candidate_idx"] = candidate_hash(psm_df["precursor_idx"].values[kept], psm_df["rank"].values[kept])
    out["valid"] = True
    t5 = time.perf_counter()
    full = fc(psm_df, frag_df, cyc)
    t6 = time.perf_counter()
    print(f"columns {1e3*(t1-t0):.2f}  C call {1e3*(t2-t1):.2f}  rows[valid] {1e3*(t3-t2):.2f}  iloc.copy {1e3*(t4-t3):.2f}  "
          f"keys+flag {1e3*(t5-t4):.2f}  | operator {1e3*(t6-t5):.2f} ms  kept {len(kept)}")
