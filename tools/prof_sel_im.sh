#!/bin/bash
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rm -rf $OUT/prof_stats_selim
N_PREC=${N_PREC:-50000} N_CYCLES=${N_CYCLES:-500} SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_selim -o r1 -- python $REPO/tools/bench_select_timstof.py > $OUT/prof_selim.log 2>&1
tail -1 $OUT/prof_selim.log | cut -c1-400
python $REPO/tools/rocpd_summary.py $OUT/prof_stats_selim/r1_results.db | grep "^adh_\|^void adh_" | sed 's/(.*),\([0-9]*,[0-9]*,[0-9]*,[0-9]*,[0-9]*,[0-9.]*\)$/,\1/' | head
rm -rf $OUT/prof_stats_selim
