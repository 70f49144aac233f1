"""profiles/pmc_traffic.json from a PMC summary (tools/rocpd_summary.py output of the FETCH_SIZE and
WRITE_SIZE passes of tools/profile_r2.sh): HBM bytes of one scoring step of the headline bench =
everything adh_gather_kernel and the feature kernels moved, divided by the number of passes over the
candidate table the profiled command made (each kernel name is launched once per chunk and pass).

    python tools/pmc_traffic.py gpurun_out/r02_pmc.csv <passes> <candidates_per_pass>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(path, passes, n_cand):
    per = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
    for line in open(path):
        parts = line.rstrip("\n").rsplit(",", 4)
        if len(parts) != 5 or parts[1] not in per:
            continue
        name = parts[0].replace("void ", "").split("(")[0]
        if name.startswith("adh_gather_kernel") or name.startswith("adh_feature"):
            per[parts[1]][name] = per[parts[1]].get(name, 0.0) + float(parts[4]) * 1024.0 / passes  # KB summed -> bytes per pass
    fetch, write = sum(per["FETCH_SIZE"].values()), sum(per["WRITE_SIZE"].values())
    feat = lambda d: sum(v for k, v in d.items() if k.startswith("adh_feature"))  # noqa: E731
    out = {
        "candidates_per_gpu": n_cand,
        "hbm_bytes_per_launch": fetch + write,
        "hbm_bytes_per_candidate": (fetch + write) / n_cand,
        "fetch_bytes": fetch,
        "write_bytes": write,
        "per_kernel_fetch_bytes": {"adh_gather_kernel": per["FETCH_SIZE"].get("adh_gather_kernel", 0.0),
                                   "feature kernels": feat(per["FETCH_SIZE"])},
        "per_kernel_write_bytes": {"adh_gather_kernel": per["WRITE_SIZE"].get("adh_gather_kernel", 0.0),
                                   "feature kernels": feat(per["WRITE_SIZE"])},
        "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, {os.path.basename(path)}), KB -> bytes, summed "
                f"over all launches of adh_gather_kernel and the feature kernels and divided by the {passes} passes over "
                "the candidate table the profiled command made (one step = all chunks of one adh_score_candidates call). "
                "WRITE_SIZE was calibrated exact on a streaming kernel; FETCH_SIZE reads half of a wide coalesced stream "
                "on gfx950 (MI355X_MICROARCH.md) and is uncorrected here for the gather's narrow random loads: the read "
                "side is a lower bound, at most 2x higher.",
    }
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:400])


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]))
