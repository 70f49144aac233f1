#!/bin/bash
# kernel-time summary of the bench workload (GPU box, from the repo root via gpurun)
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rm -rf $OUT/prof_stats
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o r1 -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
python $REPO/tools/rocpd_summary.py $OUT/prof_stats/r1_results.db | grep -v "at::native\|rocprim\|rocclr" | sed 's/(DevRun[^)]*)//; s/void //' | cut -c1-120 | head -16
