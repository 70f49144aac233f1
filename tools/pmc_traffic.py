"""profiles/pmc_traffic.json from a PMC summary (tools/rocpd_summary.py output of the FETCH_SIZE and
WRITE_SIZE passes of tools/profile_r1.sh): HBM bytes of one scoring step = per-launch averages of
adh_gather_kernel plus every feature kernel.  Usage: python tools/pmc_traffic.py gpurun_out/pmc.csv
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(path):
    per = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
    for line in open(path):
        parts = line.rstrip("\n").rsplit(",", 4)
        if len(parts) != 5 or parts[1] not in per:
            continue
        name = parts[0].replace("void ", "").split("(")[0]
        if name.startswith("adh_gather_kernel") or name.startswith("adh_feature"):
            per[parts[1]][name] = float(parts[3]) * 1024.0  # KB per launch -> bytes
    fetch, write = sum(per["FETCH_SIZE"].values()), sum(per["WRITE_SIZE"].values())
    feat = lambda d: sum(v for k, v in d.items() if k.startswith("adh_feature"))  # noqa: E731
    n_feat = sum(1 for k in per["FETCH_SIZE"] if k.startswith("adh_feature"))
    out = {
        "candidates_per_gpu": 300000,
        "hbm_bytes_per_launch": fetch + write,
        "fetch_bytes": fetch,
        "write_bytes": write,
        "per_kernel_fetch_bytes": {"adh_gather_kernel": per["FETCH_SIZE"].get("adh_gather_kernel", 0.0),
                                   f"feature kernels ({n_feat} launches)": feat(per["FETCH_SIZE"])},
        "per_kernel_write_bytes": {"adh_gather_kernel": per["WRITE_SIZE"].get("adh_gather_kernel", 0.0),
                                   f"feature kernels ({n_feat} launches)": feat(per["WRITE_SIZE"])},
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/r01_final_pmc.csv), KB -> bytes, "
                "summed over adh_gather_kernel and the register feature kernels of one scoring step (transposed run "
                "layout). WRITE_SIZE was calibrated exact on a streaming kernel; FETCH_SIZE reads half of a wide "
                "coalesced stream on gfx950 (MI355X_MICROARCH.md) and is uncorrected here for the gather's narrow "
                "random loads: the read side is a lower bound, at most 2x higher.",
    }
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:300])


if __name__ == "__main__":
    main(sys.argv[1])
