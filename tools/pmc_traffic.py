"""profiles/pmc_traffic.json from a PMC summary (tools/rocpd_summary.py output of the FETCH_SIZE and WRITE_SIZE
passes of tools/profile_r4.sh): HBM-side bytes of one scoring step of the headline bench = everything the
scoring kernels (adh_fused_kernel, and the two-kernel fallback: adh_gather_kernel, adh_feature*) moved, divided
by the number of passes over the candidate table the profiled command made.

    python tools/pmc_traffic.py gpurun_out/r03_pmc.csv <passes> <candidates_per_pass>

FETCH_SIZE counts read requests at 64 B each while a request fills a 128-byte line, for a wide coalesced stream
(MI355X_MICROARCH.md) and for the gather's pattern alike - one narrow load per lane, every load its own line
(tools/probes/fetch_probe.hip, profiles/r03_fetch_probe.txt: 64.0 B per load): the read side is doubled.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORING = ("adh_fused_kernel", "adh_gather_kernel", "adh_feature")


def main(path, passes, n_cand):
    per = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
    for line in open(path):
        parts = line.rstrip("\n").rsplit(",", 4)
        if len(parts) != 5 or parts[1] not in per:
            continue
        name = parts[0].replace("void ", "").split("(")[0]
        if name.startswith(SCORING):
            per[parts[1]][name] = per[parts[1]].get(name, 0.0) + float(parts[4]) * 1024.0 / passes  # KB summed -> bytes per pass
    fetch_counter, write = sum(per["FETCH_SIZE"].values()), sum(per["WRITE_SIZE"].values())
    fetch = 2.0 * fetch_counter
    out = {
        "candidates_per_gpu": n_cand,
        "hbm_bytes_per_launch": fetch + write,
        "hbm_bytes_per_candidate": (fetch + write) / n_cand,
        "fetch_bytes": fetch,
        "fetch_size_counter_bytes": fetch_counter,
        "write_bytes": write,
        "per_kernel_fetch_bytes": {k: 2.0 * v for k, v in sorted(per["FETCH_SIZE"].items())},
        "per_kernel_write_bytes": dict(sorted(per["WRITE_SIZE"].items())),
        "git_head": os.environ.get("GIT_HEAD", "unknown"),
        "recipe": os.environ.get("ADH_PROFILE_RECIPE", "tools/profile_r6.sh"),
        "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, {os.path.basename(path)}), KB -> bytes, summed "
                f"over all launches of the scoring kernels and divided by the {passes:g} passes over the candidate table the "
                "profiled command made (one pass = all chunks of one adh_score_candidates call, or one resident step). "
                "Read side = 2 x FETCH_SIZE: the counter tallies a request at 64 B, the request fills a 128-byte line - "
                "calibrated on a coalesced stream and on scattered 4/8/16-byte loads, one line each "
                "(profiles/r03_fetch_probe.txt).  WRITE_SIZE as reported: the output rows (449 B per candidate) in "
                "partial lines; no fused kernel spills registers any more (the launch that held the 28- and "
                "32-cycle bodies to three wavefronts per SIMD wrote 2.5 GB of scratch per step).",
    }
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]))
