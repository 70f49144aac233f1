#!/bin/bash
# kernel-trace statistics of the headline bench (gpurun, from the repo root); extra env is passed through
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rm -rf $OUT/prof_stats
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o r3 -- python $REPO/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof_stats.log 2>&1
python $REPO/tools/rocpd_summary.py $OUT/prof_stats/r3_results.db > $OUT/r3_kernel_stats.csv
rm -rf $OUT/prof_stats
grep -v "at::native\|rocprim\|rocclr" $OUT/r3_kernel_stats.csv | sed 's/(DevRun[^)]*)//; s/void //' | cut -c1-150 | head -30
tail -1 $OUT/prof_stats.log | cut -c1-200
