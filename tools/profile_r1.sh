#!/bin/bash
# Round-1 profiling recipe (run on the GPU box from the repo root via gpurun).
set -x
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
# 1. ablation: kernel time when the kernel returns after phase p
for p in 1 2 3 4 5 6 0; do
  ADH_DEBUG_STOP_PHASE=$p python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read())[\"roofline\"]; print(\"stop_phase $p\", r[\"gather_kernel_ms\"], r[\"feature_kernel_ms\"])"
done > $OUT/ablation.txt 2>&1
cat $OUT/ablation.txt
# 2. kernel trace + stats
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o r1 -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
# 3. PMC passes (own runs, no trace domains)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $OUT/prof_pmc1 -o r1 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/prof_pmc2 -o r1 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc3 -o r1 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/prof_pmc4 -o r1 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
