#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -save-temps .s file.

usage: isa_count.py file.s kernel-substring [--marks]
Prints the instruction mix (VALU f64 / VALU other / SALU / LDS / VMEM / waits / branches) and the
resource lines of the kernel; with --marks also the running VALU count at every s_cmp / v_cmp
against a small literal (the ADH_DEBUG_STOP_PHASE sites), which splits the census into phases.
The register kernels are fully unrolled, so static counts are close to executed counts.
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    marks = "--marks" in sys.argv
    lines = open(path, errors="replace").read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.rstrip().endswith(("@" + l.split(":")[0],)) or (
            l.startswith("_Z") and key in l and ": " in l and "; @" in l
        ):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    cnt = dict(valu64=0, valu=0, salu=0, lds=0, vmem=0, smem=0, wait=0, branch=0, mfma=0, other=0)
    total = 0
    i = start + 1
    while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
        l = lines[i].strip()
        i += 1
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        total += 1
        if op.startswith("v_mfma"):
            cnt["mfma"] += 1
        elif op.startswith("v_"):
            if "f64" in op:
                cnt["valu64"] += 1
            else:
                cnt["valu"] += 1
        elif op.startswith("ds_"):
            cnt["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cnt["vmem"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            cnt["smem"] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
            cnt["wait"] += 1
        elif op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            cnt["branch"] += 1
        elif op.startswith("s_"):
            cnt["salu"] += 1
        else:
            cnt["other"] += 1
        if marks and re.match(r"s_cmp_(eq|lg)_[iu]32", op):
            m = re.search(r",\s*(\d+|0x[0-9a-f]+)\s*$", l)
            if m:
                print(f"  mark {l:48s} valu={cnt['valu']} valu64={cnt['valu64']} lds={cnt['lds']} vmem={cnt['vmem']} total={total}")
    print("kernel", lines[start].split(":")[0])
    print("total", total, cnt)
    # resource summary follows the function
    for l in lines[i : i + 80]:
        if any(k in l for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs", "TotalNumVgprs")):
            print(l.strip())


main()
