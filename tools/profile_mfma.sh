#!/bin/bash
# MFMA counters of the fragment_correlation contraction (GPU box, repo root)
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rm -rf $OUT/prof_mfma
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/prof_mfma -o r1 -- python $REPO/tools/bench_nonxic.py > $OUT/prof_mfma.log 2>&1
tail -1 $OUT/prof_mfma.log
python $REPO/tools/rocpd_summary.py $OUT/prof_mfma/r1_results.db | grep "adh_feature_kernel\|^kernel" | sed 's/(DevRun[^)]*)//' > $OUT/mfma_pmc.csv
cat $OUT/mfma_pmc.csv
