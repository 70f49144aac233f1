#!/bin/bash
# FDR stage: bench line + rocprofv3 kernel statistics (GPU box, from the repo root via gpurun)
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
python $REPO/tools/bench_fdr.py > $OUT/fdr_bench.log 2>&1
tail -1 $OUT/fdr_bench.log > $OUT/fdr_bench.json
rm -rf $OUT/prof_stats_fdr
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_fdr -o r1 -- python $REPO/tools/bench_fdr.py --epochs ${FDR_EPOCHS:-2} --cpu-steps 2 > $OUT/prof_stats_fdr.log 2>&1
python $REPO/tools/rocpd_summary.py $OUT/prof_stats_fdr/r1_results.db > $OUT/fdr_kernel_stats.csv
cut -c1-700 $OUT/fdr_bench.json
grep "^adh_mlp\|^fdr::" $OUT/fdr_kernel_stats.csv | sed 's/(.*),\([0-9]*,[0-9]*,[0-9]*,[0-9]*,[0-9]*,[0-9.]*\)$/,\1/' | head -12
