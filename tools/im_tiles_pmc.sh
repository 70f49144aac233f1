# scratch driver: counters of adh_gather_im_kernel on configs[3], tile layout on / off (one --pmc pass per run)
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
cd /tmp
export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=3 TOUCHED_SAMPLE=20
CMD="python $REPO/tools/bench_timstof.py"
for v in on off; do
  if [ $v = off ]; then export ADH_DEBUG_IM_NO_TILES=1; fi
  rm -rf /tmp/q_*
  rocprofv3 --pmc FETCH_SIZE -d /tmp/q_1 -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d /tmp/q_2 -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d /tmp/q_3 -o p -- $CMD > /dev/null 2>&1
  for i in 1 2 3; do python $REPO/tools/rocpd_summary.py /tmp/q_$i/p_results.db | grep "^#\|kernel,\|adh_gather_im"; done > $OUT/im_gather_pmc_$v.csv
done
cut -c1-40,150- $OUT/im_gather_pmc_on.csv
cut -c1-40,150- $OUT/im_gather_pmc_off.csv
