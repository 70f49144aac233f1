#!/bin/bash
# SQ counters of the scoring kernels on the headline bench (gpurun, from the repo root); env passes through
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf $OUT/prof_pmc1 $OUT/prof_pmc2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $OUT/prof_pmc1 -o r3 -- $CMD > $OUT/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d $OUT/prof_pmc2 -o r3 -- $CMD > $OUT/prof_pmc2.log 2>&1
for i in 1 2; do python $REPO/tools/rocpd_summary.py $OUT/prof_pmc$i/r3_results.db; done > $OUT/r3_pmc${TAG}.csv
rm -rf $OUT/prof_pmc1 $OUT/prof_pmc2
grep "fused_kernel" $OUT/r3_pmc${TAG}.csv | sed 's/(DevRun[^)]*)//; s/void //' | cut -c1-140
