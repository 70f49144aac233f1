import sys, time, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import oracle
from alphadia_amd import synthetic as syn
from alphadia_amd.scoring import CandidateScoringConfig, fragment_columns, pack_assembled, assemble_candidates
t=time.time(); case = syn.make_case(100000, 4800, config_id=2, per_precursor=3, threads=64); print('gen', time.time()-t, flush=True)
cfg = CandidateScoringConfig(); cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True, experimental_xic=True))
soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
cols = fragment_columns(case.library.fragment_df, "mz_library")
n = len(soa['precursor_idx'])
from alphadia_amd.distributed import slice_soa
for th in (1,8,32,64,128,256):
    m = min(n, 4000*th)
    sub = slice_soa(soa, 0, m)
    p = pack_assembled(sub)
    t=time.time(); out = oracle.score(case.dia, cols, p, cfg.to_jitclass(), n_threads=th); dt=time.time()-t
    t=time.time(); out = oracle.score(case.dia, cols, p, cfg.to_jitclass(), n_threads=th); dt2=time.time()-t
    print(th,'threads:', m, 'cands', m/dt, m/dt2, 'cand/s', flush=True)
