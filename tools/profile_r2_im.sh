#!/bin/bash
# Profiles of the ion-mobility path committed as profiles/r02_timstof_*: kernel-trace statistics and two
# PMC passes (own runs) of the reduced configs[3] bench, then the bench records (scoring + selection).
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rm -rf /tmp/im_stats /tmp/im_pmc_a /tmp/im_pmc_b
CMD="python $REPO/tools/bench_timstof.py"
ADH_BENCH_NO_CPU=1 rocprofv3 --kernel-trace --stats -d /tmp/im_stats -o p -- $CMD > /tmp/im_stats.log 2>&1
ADH_BENCH_NO_CPU=1 STEPS=2 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/im_pmc_a -o p -- $CMD > /tmp/im_pmc_a.log 2>&1
ADH_BENCH_NO_CPU=1 STEPS=2 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT FETCH_SIZE -d /tmp/im_pmc_b -o p -- $CMD > /tmp/im_pmc_b.log 2>&1
python $REPO/tools/rocpd_summary.py /tmp/im_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r02_timstof_kernel_stats.csv
for d in a b; do python $REPO/tools/rocpd_summary.py /tmp/im_pmc_$d/p_results.db | grep "^#\|kernel,\|adh_gather_im\|adh_feature_im\|adh_plan_rec_im"; done > $OUT/r02_timstof_pmc.csv
cd $REPO
python tools/bench_timstof.py > $OUT/r02_timstof_bench.json 2> /dev/null
python tools/bench_select_timstof.py 2> /dev/null | tail -1 > $OUT/r02_selection_timstof_bench.json
head -8 $OUT/r02_timstof_kernel_stats.csv | cut -c1-160
cut -c1-200 $OUT/r02_timstof_bench.json
cut -c1-300 $OUT/r02_selection_timstof_bench.json
