"""Derived figures from the PMC summaries of tools/profile_r6.sh (tools/rocpd_summary.py output).

    python tools/pmc_derive.py valu  <pmc.csv> <kernel_stats.csv>     -> profiles/r06_valu_busy.json
    python tools/pmc_derive.py mfma  <pmc.csv> <kernel_stats.csv> <out.json> <kernel substring> <flops per MOPS unit>
    python tools/pmc_derive.py legs  <out.json> key=pmc.csv:passes[:kernel substrings,...] ...

valu: how busy the vector ALUs are under a kernel.  SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count in units of four
clocks summed over the wavefronts of a dispatch (checked: SQ_WAVE_CYCLES x 4 = resident wavefronts x kernel duration
x clock, and SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU within 2 %: one issue slot per instruction).  Their RATIO is the share
of its lifetime a wavefront spends issuing VALU instructions - not the utilisation of a SIMD, which holds 2-3 such
wavefronts: VALU busy per SIMD = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x kernel clocks), with the kernel clocks from
SQ_BUSY_CYCLES (summed over the 32 shader engines) or from the traced duration.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SE, N_SIMD = 32, 1024  # MI355X: 8 XCDs x 4 shader engines; 256 CUs x 4 SIMDs


def read_pmc(path):
    per = {}
    for line in open(path):
        parts = line.rstrip("\n").rsplit(",", 4)
        if len(parts) != 5 or parts[1] in ("counter",):
            continue
        try:
            per.setdefault(parts[0], {})[parts[1]] = (float(parts[2]), float(parts[3]), float(parts[4]))
        except ValueError:
            pass
    return per


def read_stats(path):
    st = {}
    for line in open(path):
        parts = line.rstrip("\n").rsplit(",", 6)
        if len(parts) == 7 and parts[1].isdigit():
            st[parts[0]] = (int(parts[1]), float(parts[2]), float(parts[3]))  # calls, total ns, avg ns
    return st


def short(name):
    return name.replace("void ", "").split("(")[0]


def valu(pmc_path, stats_path):
    pmc, st = read_pmc(pmc_path), read_stats(stats_path)
    out = {}
    for k, c in pmc.items():
        if "adh_fused_kernel" not in k or "SQ_ACTIVE_INST_VALU" not in c:
            continue
        disp, act, _ = c["SQ_ACTIVE_INST_VALU"]
        wave_cyc = c["SQ_WAVE_CYCLES"][1]
        busy = c["SQ_BUSY_CYCLES"][1] / N_SE  # clocks of one dispatch
        waves = c["SQ_WAVES"][1]
        avg_ns = next((v[2] for n, v in st.items() if short(n) == short(k)), None)
        rec = {
            "dispatches": disp,
            "wavefronts_per_dispatch": waves,
            "valu_instructions_per_wavefront": c["SQ_INSTS_VALU"][1] / waves,
            "wavefront_share_issuing_valu": act / wave_cyc,
            "resident_wavefronts_per_simd": wave_cyc * 4 / busy / N_SIMD,
            "valu_busy_per_simd": act * 4 / (busy * N_SIMD),
            "wait_any_share_of_wavefront": c.get("SQ_WAIT_ANY", (0, 0, 0))[1] / wave_cyc,
            "lds_bank_conflict_per_lds_cycle": (c["SQ_LDS_BANK_CONFLICT"][1] / c["SQ_ACTIVE_INST_LDS"][1])
            if "SQ_LDS_BANK_CONFLICT" in c and c["SQ_ACTIVE_INST_LDS"][1] else None,
            "kernel_clocks": busy,
            "avg_duration_us": None if avg_ns is None else avg_ns / 1e3,
            "clock_ghz_implied": None if avg_ns is None else busy / avg_ns,
        }
        out[short(k)] = rec
    res = {"kernels": out, "git_head": os.environ.get("GIT_HEAD", "unknown"), "recipe": os.environ.get("ADH_PROFILE_RECIPE", "tools/profile_r6.sh"),
           "note": __doc__.split("valu:")[1].strip()}
    json.dump(res, open(os.path.join(ROOT, "profiles", os.environ.get("ADH_PROFILE_ROUND", "r06") + "_valu_busy.json"), "w"), indent=1)
    for k, v in out.items():
        print(f"{k}: VALU busy per SIMD {v['valu_busy_per_simd']:.3f}, {v['resident_wavefronts_per_simd']:.2f} wavefronts per SIMD, "
              f"a wavefront issues VALU {v['wavefront_share_issuing_valu']:.3f} of its life, "
              f"{v['valu_instructions_per_wavefront']:.0f} VALU instructions per wavefront")


def mfma(pmc_path, stats_path, out_path, key, flops_per_unit):
    pmc, st = read_pmc(pmc_path), read_stats(stats_path)
    out = {}
    for k, c in pmc.items():
        if key not in k or "SQ_INSTS_VALU_MFMA_MOPS_F32" not in c:
            continue
        mops = c["SQ_INSTS_VALU_MFMA_MOPS_F32"][2]
        busy_mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0, 0))[2]
        busy = c.get("SQ_BUSY_CYCLES", (0, 0, 0))[2] / N_SE
        tot_ns = next((v[1] for n, v in st.items() if short(n) == short(k)), None)
        flops = mops * float(flops_per_unit)
        out[short(k)] = {
            "dispatches": c["SQ_INSTS_VALU_MFMA_MOPS_F32"][0], "mfma_mops_f32": mops, "flops": flops,
            "mfma_busy_cycles": busy_mfma, "kernel_clocks_sum": busy,
            "mfma_busy_share": busy_mfma / (busy * N_SIMD) if busy else None,  # (cycles summed over the SIMDs)
            "total_ms": None if tot_ns is None else tot_ns / 1e6,
            "achieved_tflops": None if not tot_ns else flops / tot_ns / 1e3,
            "share_of_f32_mfma_peak_157tf": None if not tot_ns else flops / tot_ns / 1e3 / 157.0,
        }
    res = {"kernels": out, "git_head": os.environ.get("GIT_HEAD", "unknown"), "recipe": os.environ.get("ADH_PROFILE_RECIPE", "tools/profile_r6.sh"),
           "note": "SQ_INSTS_VALU_MFMA_MOPS_F32 counts MFMA operations in units of 512 flops (MI355X_MICROARCH.md); "
                   "achieved = flops / summed kernel time of the traced run of the same command; the f32 MFMA peak is 157 TFLOP/s"}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(out)[:800])


def legs(out_path, specs):
    res = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for spec in specs:
        key, rest = spec.split("=", 1)
        bits = rest.split(":")
        path, passes = bits[0], float(bits[1])
        subs = bits[2].split(",") if len(bits) > 2 and bits[2] else ["adh_"]
        extra = dict(b.split("@") for b in bits[3].split(",")) if len(bits) > 3 else {}
        f = w = 0.0
        per = {}
        for k, c in read_pmc(path).items():
            if not any(sub in k for sub in subs):
                continue
            ff = 2.0 * c.get("FETCH_SIZE", (0, 0, 0))[2] * 1024.0 / passes
            ww = c.get("WRITE_SIZE", (0, 0, 0))[2] * 1024.0 / passes
            f, w = f + ff, w + ww
            per[short(k)] = per.get(short(k), 0.0) + ff + ww
        res[key] = {"hbm_bytes_per_pass": f + w, "fetch_bytes_per_pass": f, "write_bytes_per_pass": w, "passes": passes,
                    "per_kernel_bytes_per_pass": dict(sorted(per.items(), key=lambda kv: -kv[1])[:8]),
                    "git_head": os.environ.get("GIT_HEAD", "unknown"), "recipe": os.environ.get("ADH_PROFILE_RECIPE", "tools/profile_r6.sh"),
                    "note": "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes) of the leg's kernels / passes of the profiled command", **extra}
        print(key, f"{(f + w) / 1e9:.3f} GB per pass")
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "valu":
        valu(sys.argv[2], sys.argv[3])
    elif mode == "mfma":
        mfma(*sys.argv[2:7])
    else:
        legs(sys.argv[2], sys.argv[3:])
