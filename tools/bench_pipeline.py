"""One search step on the GPU, stage by stage, at the size of the headline configuration: candidate
selection -> candidate scoring -> FDR (classifier + q-values + fragment competition) for a
100k-precursor library against the 2 h synthetic run.  Goes through the host mirrors of the
reference's operators (DataFrames in and out), reports wall time per stage and the HIP-event kernel
time inside it, and how many planted peptides come out at 1 % FDR.

    python tools/bench_pipeline.py            # GPU box, repo root; N_PREC / N_CYCLES to resize
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # synthetic data generators
import synthetic as syn  # noqa: E402
from alphadia_amd import fdr, runtime  # noqa: E402
from alphadia_amd.scoring import DEFAULT_FEATURE_COLUMNS, CandidateScoringConfig, HipCandidateScoring  # noqa: E402
from alphadia_amd.selection import CandidateSelectionConfig, HipCandidateSelection  # noqa: E402

n_prec = int(os.environ.get("N_PREC", 100000))
cycles = int(os.environ.get("N_CYCLES", 4800))
case = syn.make_case(n_prec, cycles, config_id=2, per_precursor=1, threads=os.cpu_count() or 8)
dia, pdf, fdf = case.dia, case.library.precursor_df.copy(), case.library.fragment_df
planted = case.apex_cycle >= 0
L = dia.cycle.shape[1]
rng = np.random.default_rng(1)
rt_apex = dia.rt_values[np.clip(case.apex_cycle, 0, None) * L]
pdf.loc[planted, "rt_library"] = (rt_apex[planted] + rng.normal(0, 10, planted.sum())).astype(np.float32)
names = dict(rt_column="rt_library", mobility_column="mobility_library", precursor_mz_column="mz_library",
             fragment_mz_column="mz_library")
ctx = runtime.get_context(0)
t0 = time.perf_counter()
ctx.stage_run(dia)
stage_s = time.perf_counter() - t0

scfg = CandidateSelectionConfig()
scfg.update(dict(rt_tolerance=60.0, candidate_count=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15))
selector = HipCandidateSelection(dia, pdf, fdf, scfg, fwhm_rt=scfg.peak_len_rt, fwhm_mobility=scfg.peak_len_mobility,
                                 **names)
selector()  # warm-up (library staging, first-call allocations)
t0 = time.perf_counter()
cands = selector()
select_s = time.perf_counter() - t0
select_kernel_ms = ctx.select_time_ms()

cfg = CandidateScoringConfig()
cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                experimental_xic=True))
scorer = HipCandidateScoring(dia_data=dia, precursors_flat=pdf, fragments_flat=fdf, config=cfg, device=0, **names)
scorer(cands.iloc[:1000], thread_count=8)
ctx.kernel_time_ms(reset=True)
t0 = time.perf_counter()
features_df, fragments_df = scorer(cands, thread_count=8)
score_s = time.perf_counter() - t0
g_ms, f_ms, *_ = ctx.kernel_time_ms(reset=True)

cols = [c for c in DEFAULT_FEATURE_COLUMNS if c in features_df.columns]
clf = fdr.HipBinaryClassifier(test_size=0.001, batch_size=5000, learning_rate=0.001, epochs=10,
                              experimental_hyperparameter_tuning=True, random_state=3)
t0 = time.perf_counter()
res = fdr.perform_fdr(clf, cols, features_df[features_df["decoy"] == 0].copy(),
                      features_df[features_df["decoy"] == 1].copy(), competitive=True, group_channels=True,
                      df_fragments=fragments_df, dia_cycle=dia.cycle, random_state=4)
fdr_s = time.perf_counter() - t0
hits = res[res["qval"] <= 0.01]
ids = hits[hits["decoy"] == 0]
found = planted[ids["precursor_idx"].to_numpy()]
print(json.dumps({
    "workload": f"{n_prec} precursors (half decoys, {int(planted.sum())} planted) vs {cycles} cycles x {L} spectra: "
                f"selection (rt +- {scfg.rt_tolerance} s, {scfg.candidate_count} candidates) -> scoring -> FDR",
    "stage_run_s": stage_s,
    "selection": {"wall_s": select_s, "kernel_ms": select_kernel_ms, "candidates": int(len(cands))},
    "scoring": {"wall_s": score_s, "gather_kernel_ms": g_ms, "feature_kernel_ms": f_ms, "rows": int(len(features_df)),
                "fragment_rows": int(len(fragments_df))},
    "fdr": {"wall_s": fdr_s, "classifier_fit_kernels_ms": clf.last_fit_ms, "classifier_predict_kernel_ms": clf.last_predict_ms,
            "batch_size": int(clf.batch_size), "rows_out": int(len(res))},
    "gpu_kernel_ms_total": select_kernel_ms + g_ms + f_ms + clf.last_fit_ms + clf.last_predict_ms,
    "targets_at_1pct": int(len(ids)), "planted_found": int(found.sum()), "not_planted": int((~found).sum()),
    "decoys_at_1pct": int((hits["decoy"] == 1).sum()),
}))
