#!/bin/bash
# (tools/rocpd_summary.py reads a database as kernel statistics when its path holds "stats", as counters otherwise)
# Profiling recipe of the committed profiles/r04_* files (GPU box, from the repo root via gpurun).  Every PMC pass is a
# run of its own, never together with a trace domain other than --kernel-trace.
#   headline  : kernel-trace statistics + PMC passes of `bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras`
#               -> r04_final_kernel_stats.csv, r04_final_pmc.csv, pmc_traffic.json (stamped with the commit)
#   configs[3]: kernel trace + FETCH_SIZE / WRITE_SIZE of tools/bench_timstof.py at full size
#               -> r04_timstof_kernel_stats.csv, r04_timstof_pmc.csv, timstof_traffic.json
#   fragcomp / configs[4] / selection: kernel traces of the bench legs -> r04_fragcomp_kernel_stats.csv,
#               r04_multiplex_kernel_stats.csv, r04_selection_kernel_stats.csv
#   then the driver-style bench line with all legs -> r04_final_bench.json
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
HEAD_ID=${GIT_HEAD:-unknown}
mkdir -p $OUT
cd /tmp
rm -rf /tmp/p4_*
CMD="python $REPO/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d /tmp/p4_stats -o r4 -- $CMD > $OUT/p4_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d /tmp/p4_pmc1 -o r4 -- $CMD > $OUT/p4_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d /tmp/p4_pmc2 -o r4 -- $CMD > $OUT/p4_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p4_pmc3 -o r4 -- $CMD > $OUT/p4_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/p4_pmc4 -o r4 -- $CMD > $OUT/p4_pmc4.log 2>&1
python $REPO/tools/rocpd_summary.py /tmp/p4_stats/r4_results.db > $OUT/r04_final_kernel_stats.csv
for i in 1 2 3 4; do python $REPO/tools/rocpd_summary.py /tmp/p4_pmc$i/r4_results.db; done > $OUT/r04_final_pmc.csv
# passes over the candidate table of the profiled command: 4 priming + 3 warm-up + 5 steps + 2 + 5 resident = 19
( cd $REPO && GIT_HEAD=$HEAD_ID python tools/pmc_traffic.py $OUT/r04_final_pmc.csv 19 3000000 && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json )

# ---- configs[3]
export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=3 TOUCHED_SAMPLE=20
CMD="python $REPO/tools/bench_timstof.py"
rocprofv3 --kernel-trace --stats -d /tmp/p4_im_stats -o p -- $CMD > $OUT/p4_im.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p4_imf -o p -- $CMD > $OUT/p4_imf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p4_imw -o p -- $CMD > $OUT/p4_imw.log 2>&1
python $REPO/tools/rocpd_summary.py /tmp/p4_im_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r04_timstof_kernel_stats.csv
for d in f w; do python $REPO/tools/rocpd_summary.py /tmp/p4_im$d/p_results.db | grep "^#\|kernel,\|adh_gather_im\|adh_feature_im"; done > $OUT/r04_timstof_pmc.csv
python - <<PY
import json
passes = 4 + 3 + 3 + 1  # warm-up + host -> host + resident steps of the profiled command
f = w = gf = gw = 0.0
for line in open("$OUT/r04_timstof_pmc.csv"):
    parts = line.rstrip("\n").rsplit(",", 4)
    if len(parts) == 5 and parts[1] in ("FETCH_SIZE", "WRITE_SIZE"):
        v = float(parts[4]) * 1024.0 / passes
        gather = "adh_gather_im" in parts[0]
        if parts[1] == "FETCH_SIZE":
            f += 2.0 * v   # requests are tallied at 64 B, the lines they fill are 128 B (profiles/r03_fetch_probe.txt)
            gf += 2.0 * v if gather else 0.0
        else:
            w += v
            gw += v if gather else 0.0
json.dump({"candidates": 600000, "passes": passes, "fetch_bytes_per_pass": f, "write_bytes_per_pass": w, "hbm_bytes_per_pass": f + w,
           "gather_kernel_hbm_bytes_per_pass": gf + gw, "git_head": "$HEAD_ID", "recipe": "tools/profile_r4.sh"},
          open("$OUT/timstof_traffic.json", "w"), indent=1)
print(open("$OUT/timstof_traffic.json").read())
PY
unset N_PREC N_CYCLES SCAN_MAX N_TOF EVENTS_PER_PUSH ADH_BENCH_NO_CPU STEPS TOUCHED_SAMPLE

# ---- fragment competition, configs[4]
rocprofv3 --kernel-trace --stats -d /tmp/p4_fc_stats -o p -- python $REPO/tools/bench_legs.py fragcomp > $OUT/p4_fc.log 2>&1
python $REPO/tools/rocpd_summary.py /tmp/p4_fc_stats/p_results.db | grep -v "rocclr" > $OUT/r04_fragcomp_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p4_mx_stats -o p -- python $REPO/tools/bench_legs.py multiplex > $OUT/p4_mx.log 2>&1
python $REPO/tools/rocpd_summary.py /tmp/p4_mx_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r04_multiplex_kernel_stats.csv
# ---- candidate selection on the AlphaRaw run (the ion-mobility selection is part of the configs[3] trace above)
ADH_BENCH_NO_CPU=1 rocprofv3 --kernel-trace --stats -d /tmp/p4_sel_stats -o p -- python $REPO/tools/bench_select.py > $OUT/p4_sel.log 2>&1
python $REPO/tools/rocpd_summary.py /tmp/p4_sel_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r04_selection_kernel_stats.csv
rm -rf /tmp/p4_*

# ---- the bench line itself (driver style), with the traffic files of this run in place
cd $REPO
cp $OUT/timstof_traffic.json profiles/timstof_traffic.json
python bench.py --steps 20 --warmup 5 > $OUT/r04_final_bench.json 2> $OUT/r04_final_bench.log
tail -1 $OUT/r04_final_bench.json | cut -c1-400
grep -v "at::native\|rocprim\|rocclr" $OUT/r04_final_kernel_stats.csv | sed 's/(DevRun[^)]*)//; s/void //' | cut -c1-130 | head -14
head -8 $OUT/r04_timstof_kernel_stats.csv | cut -c1-130
head -12 $OUT/r04_fragcomp_kernel_stats.csv | cut -c1-130
head -6 $OUT/r04_selection_kernel_stats.csv | cut -c1-130
