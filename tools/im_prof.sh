#!/bin/bash
# configs[3] at full size under rocprofv3: per-kernel times (and, with PMC=1, issue / wait counters) of the ion-mobility
# scoring kernels.  NAME=tag names the outputs (gpurun_out/im_prof_<tag>_*); extra environment goes through as is.
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
NAME=${NAME:-x}
mkdir -p $OUT
export N_PREC=${N_PREC:-200000} N_CYCLES=${N_CYCLES:-2000} SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=${STEPS:-3} TOUCHED_SAMPLE=20 TOUCHED_SAMPLE_SEL=10
cd /tmp
S="python $REPO/tools/rocpd_summary.py"
CMD="python $REPO/tools/bench_timstof.py"
rm -rf /tmp/imp_stats /tmp/imp_pmc1 /tmp/imp_pmc2
rocprofv3 --kernel-trace --stats -d /tmp/imp_stats -o p -- $CMD > $OUT/im_prof_${NAME}.log 2>&1
$S /tmp/imp_stats/p_results.db | grep -v "rocprim\|rocclr\|at::native" > $OUT/im_prof_${NAME}_kernel_stats.csv
grep "adh_gather_im\|adh_feature_im" $OUT/im_prof_${NAME}_kernel_stats.csv | sed 's/(DevTims[^)]*)//; s/void //' | awk -F, '{printf "%-90s calls %s total_ms %.2f avg_us %.1f\n", substr($1,1,90), $(NF-5), $(NF-4)/1e6, $(NF-3)/1e3}'
if [ -n "$PMC" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d /tmp/imp_pmc1 -o p -- $CMD > $OUT/im_prof_${NAME}_pmc1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/imp_pmc2 -o p -- $CMD > $OUT/im_prof_${NAME}_pmc2.log 2>&1
  for i in 1 2; do $S /tmp/imp_pmc$i/p_results.db | grep "^#\|kernel,\|adh_gather_im\|adh_feature_im"; done > $OUT/im_prof_${NAME}_pmc.csv
fi
