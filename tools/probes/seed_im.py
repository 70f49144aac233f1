import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import synthetic as syn
from alphadia_amd import runtime
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled
from oracle import oracle
seed = int(os.environ.get("SEED", 165))
rng = np.random.default_rng(2000 + seed)
case = syn.make_timstof_case(
    n_precursors=int(rng.integers(60, 200)), n_cycles=int(rng.integers(25, 60)), config_id=500 + seed,
    per_precursor=int(rng.integers(1, 4)), n_ms2_frames=int(rng.integers(2, 7)),
    windows_per_frame=int(rng.integers(1, 4)), scan_max_index=int(rng.choice([48, 64, 96])),
    events_per_push=float(rng.choice([15.0, 40.0, 80.0])), planted_fraction=float(rng.uniform(0.2, 0.8)),
)
soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
upd = dict(
    top_k_fragments=int(rng.choice([5, 12, 16])), top_k_isotopes=int(rng.integers(1, 5)),
    precursor_mz_tolerance=float(rng.choice([10, 40])), fragment_mz_tolerance=float(rng.choice([15, 60])),
    quant_window=int(rng.integers(1, 5)), quant_all=bool(rng.integers(0, 2)),
    experimental_xic=bool(rng.integers(0, 2)),
)
print(upd)
cfg = CandidateScoringConfig(); cfg.update(upd)
ctx = runtime.get_context(0)
def run():
    ctx.stage_run(case.dia, force=True)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"), force=True)
    g = ctx.score_host(pack_assembled(soa), cfg.to_jitclass(), with_stats=True)
    return {k: np.array(v, copy=True) for k, v in g.items()}
got = run()
exp = oracle.score_timstof(case.dia, fragment_columns(case.library.fragment_df, "mz_library"), pack_assembled(soa), cfg.to_jitclass(), n_threads=8, with_stats=True)
v = exp["valid"].astype(bool)
print("valid equal", np.array_equal(got["valid"], exp["valid"]))
gf, ef = got["features"][v], exp["features"][v]
bad = np.argwhere(~np.isclose(gf, ef, rtol=1e-4, atol=0, equal_nan=True))
print("mismatches", bad[:20])
for r_, f_ in bad[:5]:
    print(r_, f_, gf[r_, f_], ef[r_, f_])
rows = np.flatnonzero(v)
if len(bad):
    r0 = rows[bad[0][0]]
    print("candidate row", r0, {k: soa[k][r0] for k in ("precursor_idx","rank","scan_start","scan_stop","frame_start","frame_stop","charge")})
    print("got features", got["features"][r0][[4,5,6,7,8,11,12,13,14,15,16]])
    print("exp features", exp["features"][r0][[4,5,6,7,8,11,12,13,14,15,16]])
for env in ({"ADH_DEBUG_IM": "8"}, {"ADH_IM_INDEX": "0"}):
    os.environ.update(env)
    g2 = run()
    for k in env: del os.environ[k]
    b2 = np.argwhere(~np.isclose(g2["features"][v], ef, rtol=1e-4, atol=0, equal_nan=True))
    print(env, "mismatches", len(b2))
