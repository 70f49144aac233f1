#!/bin/bash
# scratch driver: per-kernel times of the ion-mobility selection under the smoothing kernel's ablation switches
# (ADH_DEBUG_SELECT_IM_ABL: 0 product, 1 no log, 2 no pass 2, 3 no taps, 4 the walk without the taps, 5 per-cell row walk instead of tap lists, 8 dense pass 1; any non-zero value runs the developer instantiation of the kernel)
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
for abl in ${ABLS:-0 1 2 3}; do
  rm -rf /tmp/selabl_stats
  ADH_DEBUG_SELECT_IM_ABL=$abl N_PREC=${N_PREC:-50000} N_CYCLES=${N_CYCLES:-500} SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 \
    rocprofv3 --kernel-trace --stats -d /tmp/selabl_stats -o r1 -- python $REPO/tools/bench_select_timstof.py > /tmp/selabl_stats.log 2>&1
  echo "abl $abl: $(grep "^{" /tmp/selabl_stats.log | tail -1 | cut -c1-300)"
  python $REPO/tools/rocpd_summary.py /tmp/selabl_stats/r1_results.db | grep "adh_select" | sed 's/(.*),\([0-9]*,[0-9]*,[0-9]*,[0-9]*,[0-9]*,[0-9.]*\)$/,\1/'
done 2>&1 | tee $OUT/sel_im_abl.txt
