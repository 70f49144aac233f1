#!/bin/bash
# PMC of the ion-mobility selection kernels (own runs, counters only): instruction mix, then wait reasons
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INST_CYCLES_SALU"; do
  rm -rf /tmp/sel_pmc
  N_PREC=${N_PREC:-50000} N_CYCLES=${N_CYCLES:-500} SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 rocprofv3 --pmc $pass -d /tmp/sel_pmc -o p -- python $REPO/tools/bench_select_timstof.py > /tmp/sel_pmc.log 2>&1
  python $REPO/tools/rocpd_summary.py /tmp/sel_pmc/p_results.db | grep "adh_select_score" | sed 's/(DevTims[^)]*)//' | cut -c1-160
done
