#!/bin/bash
# Profiles of the ion-mobility path (BASELINE configs[3] at full size) committed as profiles/r03_timstof_*:
# kernel-trace statistics, the FETCH_SIZE / WRITE_SIZE passes (own runs), then the bench record with the
# measured traffic.  gpurun, from the repo root.
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=3
rm -rf /tmp/im_stats /tmp/im_f /tmp/im_w
CMD="python $REPO/tools/bench_timstof.py"
rocprofv3 --kernel-trace --stats -d /tmp/im_stats -o p -- $CMD > /tmp/im_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/im_f -o p -- $CMD > /tmp/im_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/im_w -o p -- $CMD > /tmp/im_w.log 2>&1
python $REPO/tools/rocpd_summary.py /tmp/im_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r03_timstof_kernel_stats.csv
for d in f w; do python $REPO/tools/rocpd_summary.py /tmp/im_$d/p_results.db | grep "^#\|kernel,\|adh_gather_im\|adh_feature_im"; done > $OUT/r03_timstof_pmc.csv
# passes over the candidate table of the profiled command: 4 warm-up + STEPS host -> host + (STEPS + 1) resident
python - <<PY
import json
passes = 4 + 3 + 3 + 1
f = w = gf = gw = 0.0
for line in open("$OUT/r03_timstof_pmc.csv"):
    parts = line.rstrip("\n").rsplit(",", 4)
    if len(parts) == 5 and parts[1] in ("FETCH_SIZE", "WRITE_SIZE"):
        v = float(parts[4]) * 1024.0 / passes
        gather = parts[0].startswith("adh_gather_im")
        if parts[1] == "FETCH_SIZE":
            f += 2.0 * v   # requests are tallied at 64 B, the lines they fill are 128 B (tools/probes/fetch_probe.hip)
            gf += 2.0 * v if gather else 0.0
        else:
            w += v
            gw += v if gather else 0.0
json.dump({"candidates": 600000, "passes": passes, "fetch_bytes_per_pass": f, "write_bytes_per_pass": w, "hbm_bytes_per_pass": f + w,
           "gather_kernel_hbm_bytes_per_pass": gf + gw},
          open("$OUT/r03_timstof_traffic.json", "w"), indent=1)
print(open("$OUT/r03_timstof_traffic.json").read())
PY
cd $REPO
unset ADH_BENCH_NO_CPU
STEPS=5 ADH_IM_TRAFFIC_JSON=$OUT/r03_timstof_traffic.json python tools/bench_timstof.py > $OUT/r03_timstof_full_bench.json 2> /dev/null
head -6 $OUT/r03_timstof_kernel_stats.csv | cut -c1-160
cut -c1-1200 $OUT/r03_timstof_full_bench.json
