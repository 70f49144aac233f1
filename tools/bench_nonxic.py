"""The bench batch scored with experimental_xic = False (fragment_correlation: the dense K x F x K
contraction, done with MFMA in adh_feature_kernel).  Used by tools/profile_mfma.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # synthetic data generators
import torch  # noqa: E402

import synthetic as syn
from alphadia_amd import runtime  # noqa: E402
from torch_transport import DeviceTables  # noqa: E402  (tests/: the torch-side packed buffer)
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled  # noqa: E402

case = syn.make_case(int(os.environ.get("N_PREC", 50000)), 2400, config_id=2, per_precursor=3, threads=os.cpu_count() or 8)
cfg = CandidateScoringConfig()
cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True,
                experimental_xic=False, top_k_fragments=12))
cfgj = cfg.to_jitclass()
soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
n = len(soa["precursor_idx"])
ctx = runtime.get_context(0)
ctx.stage_run(case.dia)
ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
ctx.upload_candidates(pack_assembled(soa))
dev = torch.device("cuda", 0)
tables = DeviceTables(n, 12, dev)
out = tables.as_output(n)
ws = torch.cuda.Stream(device=dev)
with torch.cuda.stream(ws):
    for it in range(4):
        if it == 1:
            torch.cuda.synchronize()
            ctx.kernel_time_ms(reset=True)
        tables.zero_()
        ctx.score_uploaded(cfgj, out, ws.cuda_stream)
torch.cuda.synchronize()
g, f, _ = ctx.kernel_time_ms(reset=True)
print(f"{n} candidates, experimental_xic=False: gather {g:.3f} ms, features (adh_feature_kernel, MFMA Gram matrix) {f:.3f} ms, "
      f"valid {int(tables.to_host()['valid'].sum())}")
