"""Developer sweep: re-stage the bench workload under different environment switches
and print the HIP-event kernel times.  Usage (on the GPU box):
    python tools/sweep_env.py ADH_BLOCK_CYCLES=32,64,128 [ADH_DEBUG_STOP_PHASE=0,2]
"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # synthetic data generators


def main():
    import torch

    import synthetic as syn
from alphadia_amd import runtime
    from alphadia_amd.distributed import DeviceTables
    from alphadia_amd.scoring import (CandidateScoringConfig, assemble_candidates, fragment_columns,
                                      pack_assembled)

    axes = []
    for a in sys.argv[1:]:
        k, v = a.split("=")
        axes.append([(k, x) for x in v.split(",")])
    n_prec = int(os.environ.get("SWEEP_PRECURSORS", "100000"))
    cycles = int(os.environ.get("SWEEP_CYCLES", "4800"))
    case = syn.make_case(n_prec, cycles, config_id=2, per_precursor=3, threads=os.cpu_count() or 8)
    cfg = CandidateScoringConfig()
    cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1,
                    precursor_mz_tolerance=10, fragment_mz_tolerance=15, exclude_shared_ions=True,
                    quant_window=3, quant_all=True, experimental_xic=True, top_k_fragments=12))
    cfgj = cfg.to_jitclass()
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
    n = len(soa["precursor_idx"])
    device = torch.device("cuda", 0)
    ctx = runtime.get_context(0)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
    tables = DeviceTables(n, int(cfgj.top_k_fragments), device, with_stats=True)
    out = tables.as_output(n)
    ws = torch.cuda.Stream(device=device)
    ref = None
    for combo in itertools.product(*axes) if axes else [()]:
        for k, v in combo:
            if v == "":
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        t0 = time.time()
        ctx.stage_run(case.dia, force=True)
        ts = time.time() - t0
        ctx.upload_candidates(pack_assembled(soa))
        for it in range(7):
            if it == 2:
                torch.cuda.synchronize()
                ctx.kernel_time_ms(reset=True)
                t0 = time.perf_counter()
            with torch.cuda.stream(ws):
                tables.zero_()
                ctx.score_uploaded(cfgj, out, ws.cuda_stream)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 5 * 1e3
        g, f, _ = ctx.kernel_time_ms(reset=True)
        host = tables.to_host()
        chk = float(np.nan_to_num(host["features"]).astype(np.float64).sum())
        if ref is None:
            ref = chk
        print(" ".join(f"{k}={v}" for k, v in combo),
              f"stage {ts:.2f}s step {el:.3f} ms gather {g:.3f} ms features {f:.3f} ms "
              f"valid {int(host['valid'].sum())} checksum {'same' if chk == ref else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
