"""Developer check: ion-mobility golden inputs, HIP vs oracle, feature by feature."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from alphadia_amd import runtime
from alphadia_amd.scoring import assemble_candidates, fragment_columns, pack_assembled
from oracle import oracle
z, dia, fragment_df, precursor_df, cand, cfg = T._tims_case_from_golden()
soa = assemble_candidates(cand, precursor_df, "mz_library")
ctx = runtime.get_context(0)
got = T._hip_score_tims(ctx, dia, fragment_df, soa, cfg, with_stats=True)
exp = oracle.score_timstof(dia, fragment_columns(fragment_df, "mz_library"), pack_assembled(soa), cfg.to_jitclass(), with_stats=True)
v = exp["valid"].astype(bool)
print("valid equal", np.array_equal(got["valid"], exp["valid"]), v.sum())
gf, ef = got["features"][v], exp["features"][v]
for j in range(46):
    d = np.abs(gf[:, j].astype(np.float64) - ef[:, j]) / np.maximum(np.abs(ef[:, j]), 1e-6)
    if np.nanmax(d) > 1e-6 or np.isnan(gf[:, j]).sum() != np.isnan(ef[:, j]).sum():
        i = int(np.nanargmax(d))
        print(j, "max rel", np.nanmax(d), "rows off", int((d > 1e-6).sum()), "e.g.", gf[i, j], ef[i, j])
for name in ("fragment_mz_observed", "fragment_height", "fragment_intensity", "fragment_mass_error", "fragment_correlation"):
    a, b = got[name][v], exp[name][v]
    print(name, "max abs diff", np.nanmax(np.abs(a.astype(np.float64) - b)))
a, b = got["fragment_height"][v], exp["fragment_height"][v]
bad = np.abs(a.astype(np.float64) - b) > 1e-3 * np.maximum(np.abs(b), 1)
print("bad height positions per fragment slot:", bad.sum(axis=0))
nobs = ef[:, 17]
print("rows bad by n_obs:", {int(o): int(bad[nobs == o].any(axis=1).sum()) for o in np.unique(nobs)}, {int(o): int((nobs == o).sum()) for o in np.unique(nobs)})
i = int(np.flatnonzero(bad.any(axis=1))[0])
print("row", i, "nobs", nobs[i], "\n got", a[i], "\n exp", b[i])
print(" mz got", got["fragment_mz_observed"][v][i], "\n mz exp", exp["fragment_mz_observed"][v][i])
