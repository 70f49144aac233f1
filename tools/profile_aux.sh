#!/bin/bash
# kernel-trace statistics of the secondary benches (selection, ion mobility); GPU box, repo root
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
for name in select timstof select_timstof; do
  rm -rf $OUT/prof_stats_$name
  ADH_BENCH_NO_CPU=1 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$name -o r1 -- python $REPO/tools/bench_$name.py > $OUT/prof_stats_$name.log 2>&1
  python $REPO/tools/rocpd_summary.py $OUT/prof_stats_$name/r1_results.db > $OUT/${name}_kernel_stats.csv
  grep -v "at::native\|rocprim\|rocclr" $OUT/${name}_kernel_stats.csv | cut -c1-150 | head -8
done
