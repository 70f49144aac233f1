#!/bin/bash
# Profiling recipe of the committed profiles/r03_* files (GPU box, from the repo root via gpurun):
# kernel-trace statistics of the headline bench, then the PMC passes (each in its own run, never together
# with a trace domain other than --kernel-trace), the FETCH_SIZE calibration probe, then the bench line itself.
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
for d in prof_stats prof_pmc1 prof_pmc2 prof_pmc3 prof_pmc4 prof_cal; do rm -rf $OUT/$d; done
CMD="python $REPO/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o r3 -- $CMD > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $OUT/prof_pmc1 -o r3 -- $CMD > $OUT/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d $OUT/prof_pmc2 -o r3 -- $CMD > $OUT/prof_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc3 -o r3 -- $CMD > $OUT/prof_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/prof_pmc4 -o r3 -- $CMD > $OUT/prof_pmc4.log 2>&1
python $REPO/tools/rocpd_summary.py $OUT/prof_stats/r3_results.db > $OUT/r03_kernel_stats.csv
for i in 1 2 3 4; do python $REPO/tools/rocpd_summary.py $OUT/prof_pmc$i/r3_results.db; done > $OUT/r03_pmc.csv
# calibration of FETCH_SIZE on the gather's access pattern (tools/probes/fetch_probe.hip)
( cd $REPO/tools/probes && hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe fetch_probe.hip 2>/dev/null )
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_cal -o cal -- /tmp/fetch_probe > $OUT/r03_fetch_probe.txt 2>&1
python $REPO/tools/rocpd_summary.py $OUT/prof_cal/cal_results.db >> $OUT/r03_fetch_probe.txt
for d in prof_stats prof_pmc1 prof_pmc2 prof_pmc3 prof_pmc4 prof_cal; do rm -rf $OUT/$d; done  # keep the summaries only (gpurun_out is capped at 64 MiB)
cd $REPO
python bench.py --steps 20 --warmup 5 > $OUT/r03_bench.json 2> $OUT/r03_bench.log
tail -1 $OUT/r03_bench.json | cut -c1-300
grep -v "at::native\|rocprim\|rocclr" $OUT/r03_kernel_stats.csv | sed 's/(DevRun[^)]*)//; s/void //' | cut -c1-120 | head -16
grep -i "stream16\|scattered\|asked" $OUT/r03_fetch_probe.txt | cut -c1-200
