"""Cumulative kernel time of the two-kernel path up to each stop point (ADH_DEBUG_STOP_PHASE) on the transfer-library
requantification workload (20-40 fragments per precursor, top_k_fragments = 9999; tools/bench_legs.py transfer).
Generic feature kernel: 3 = tiles, template, presence, profiles; 4 = weight tables, centre of mass, precursor means;
5 = envelope, quantification, observation means; 6 = assemble part 1; 0 = everything."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import synthetic as syn
    from alphadia_amd import runtime
    from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled

    n_prec = int(os.environ.get("N_PREC", 100_000))
    case = syn.make_case(n_prec, 4800, config_id=2, per_precursor=3, threads=os.cpu_count(), k_fragments=(20, 40))
    cfg = CandidateScoringConfig()
    cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1, precursor_mz_tolerance=10,
                    fragment_mz_tolerance=15, exclude_shared_ions=True, quant_window=3, quant_all=True,
                    experimental_xic=True, top_k_fragments=9999))
    cfgj = cfg.to_jitclass()
    ctx = runtime.get_context(0)
    soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library", pool=ctx.pinned)
    ctx.stage_run(case.dia)
    ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
    packed = pack_assembled(soa)
    n = len(soa["precursor_idx"])
    for phase in [int(x) for x in os.environ.get("PHASES", "3 4 5 6 0").split()]:
        os.environ["ADH_DEBUG_STOP_PHASE"] = str(phase)
        for _ in range(2):
            ctx.score_host(packed, cfgj, reuse_buffers=True)
        ctx.kernel_time_ms(reset=True)
        steps = 3
        for _ in range(steps):
            ctx.score_host(packed, cfgj, reuse_buffers=True)
        g, f, nl = ctx.kernel_time_ms(reset=True)
        print(f"stop {phase}: gather {g * nl / steps:.2f} ms, features {f * nl / steps:.2f} ms "
              f"({f * nl / steps / n * 1e6:.1f} ns per candidate)", flush=True)
    os.environ.pop("ADH_DEBUG_STOP_PHASE", None)


if __name__ == "__main__":
    main()
