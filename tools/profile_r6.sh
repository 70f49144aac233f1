#!/bin/bash
# Profiling recipe of the committed profiles/r06_* files (GPU box, from the repo root via gpurun; GIT_HEAD=<commit>).
# Every PMC pass is a run of its own, never together with a trace domain other than --kernel-trace.
#   headline  : kernel-trace statistics + PMC passes of `bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras`
#               -> r06_final_kernel_stats.csv, r06_final_pmc.csv, pmc_traffic.json, r06_valu_busy.json
#   MFMA      : SQ_INSTS_VALU_MFMA_MOPS_F32 / SQ_VALU_MFMA_BUSY_CYCLES of the ion-mobility profile kernel (configs[3])
#               and of the classifier fit (tools/bench_fdr.py) -> r06_mfma_pmc.csv, r06_mfma.json
#   legs      : FETCH_SIZE / WRITE_SIZE of configs[4], fragment competition (1e6 PSMs), candidate selection, transfer requantification
#               -> r06_legs_pmc.csv, legs_traffic.json; configs[3] as in round 4 -> timstof_traffic.json
#   then the driver-style bench line with all legs -> r06_final_bench.json
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
HEAD_ID=${GIT_HEAD:-unknown}
export GIT_HEAD=$HEAD_ID
mkdir -p $OUT
cd /tmp
rm -rf /tmp/p6_*
S="python $REPO/tools/rocpd_summary.py"
export ADH_BENCH_NO_COMPACT=1  # (the profiled passes: the padded step and the resident leg only, 19 passes as tools/pmc_traffic.py counts them)
CMD="python $REPO/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d /tmp/p6_stats -o r5 -- $CMD > $OUT/p6_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d /tmp/p6_pmc1 -o r5 -- $CMD > $OUT/p6_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d /tmp/p6_pmc2 -o r5 -- $CMD > $OUT/p6_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p6_pmc3 -o r5 -- $CMD > $OUT/p6_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/p6_pmc4 -o r5 -- $CMD > $OUT/p6_pmc4.log 2>&1
$S /tmp/p6_stats/r5_results.db > $OUT/r06_final_kernel_stats.csv
for i in 1 2 3 4; do $S /tmp/p6_pmc$i/r5_results.db; done > $OUT/r06_final_pmc.csv
# passes over the candidate table of the profiled command: 4 priming + 3 warm-up + 5 steps + 2 + 5 resident = 19
( cd $REPO && python tools/pmc_traffic.py $OUT/r06_final_pmc.csv 19 3000000 && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json )
( cd $REPO && python tools/pmc_derive.py valu $OUT/r06_final_pmc.csv $OUT/r06_final_kernel_stats.csv && cp profiles/r06_valu_busy.json $OUT/ )

# ---- configs[3]: kernel trace, FETCH / WRITE, MFMA counters
export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=3 TOUCHED_SAMPLE=20 TOUCHED_SAMPLE_SEL=10
CMD="python $REPO/tools/bench_timstof.py"
rocprofv3 --kernel-trace --stats -d /tmp/p6_im_stats -o p -- $CMD > $OUT/p6_im.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p6_imf -o p -- $CMD > $OUT/p6_imf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p6_imw -o p -- $CMD > $OUT/p6_imw.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/p6_imm -o p -- $CMD > $OUT/p6_imm.log 2>&1
$S /tmp/p6_im_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r06_timstof_kernel_stats.csv
for d in f w; do $S /tmp/p6_im$d/p_results.db | grep "^#\|kernel,\|adh_gather_im\|adh_feature_im\|adh_select"; done > $OUT/r06_timstof_pmc.csv
$S /tmp/p6_imm/p_results.db | grep "^#\|kernel,\|adh_feature_im" > $OUT/r06_mfma_pmc.csv
python - <<PY
import json
passes = 4 + 3 + 3 + 1  # warm-up + host -> host + resident steps of the profiled command
f = w = gf = gw = sf = sw = 0.0
for line in open("$OUT/r06_timstof_pmc.csv"):
    parts = line.rstrip("\n").rsplit(",", 4)
    if len(parts) == 5 and parts[1] in ("FETCH_SIZE", "WRITE_SIZE"):
        sel = "adh_select" in parts[0]
        v = float(parts[4]) * 1024.0 / (2 if sel else passes)  # (the selection ran twice)
        gather = "adh_gather_im" in parts[0]
        if parts[1] == "FETCH_SIZE":
            v *= 2.0   # requests are tallied at 64 B, the lines they fill are 128 B (profiles/r03_fetch_probe.txt)
            if sel: sf += v
            else: f += v; gf += v if gather else 0.0
        else:
            if sel: sw += v
            else: w += v; gw += v if gather else 0.0
json.dump({"candidates": 600000, "passes": passes, "fetch_bytes_per_pass": f, "write_bytes_per_pass": w, "hbm_bytes_per_pass": f + w,
           "gather_kernel_hbm_bytes_per_pass": gf + gw, "selection_hbm_bytes_per_pass": sf + sw,
           "git_head": "$HEAD_ID", "recipe": "tools/profile_r6.sh"},
          open("$OUT/timstof_traffic.json", "w"), indent=1)
print(open("$OUT/timstof_traffic.json").read())
PY
unset N_PREC N_CYCLES SCAN_MAX N_TOF EVENTS_PER_PUSH ADH_BENCH_NO_CPU STEPS TOUCHED_SAMPLE TOUCHED_SAMPLE_SEL

# ---- the classifier fit: MFMA counters + trace
CMD="python $REPO/tools/bench_fdr.py --cpu-steps 0"
ADH_FDR_NUMPY_INIT=1 rocprofv3 --kernel-trace --stats -d /tmp/p6_fdr_stats -o p -- $CMD > $OUT/p6_fdr.log 2>&1
ADH_FDR_NUMPY_INIT=1 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/p6_fdrm -o p -- $CMD > $OUT/p6_fdrm.log 2>&1
$S /tmp/p6_fdr_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r06_fdr_kernel_stats.csv
$S /tmp/p6_fdrm/p_results.db | grep "adh_mlp" >> $OUT/r06_mfma_pmc.csv
( cd $REPO && python tools/pmc_derive.py mfma $OUT/r06_mfma_pmc.csv $OUT/r06_timstof_kernel_stats.csv $OUT/r06_mfma_im.json adh_feature_im 512 \
  && python tools/pmc_derive.py mfma $OUT/r06_mfma_pmc.csv $OUT/r06_fdr_kernel_stats.csv $OUT/r06_mfma_fdr.json adh_mlp 512 )

# ---- legs: kernel traces + FETCH / WRITE of configs[4], fragment competition (1e6), candidate selection
rocprofv3 --kernel-trace --stats -d /tmp/p6_fc_stats -o p -- python $REPO/tools/bench_legs.py fragcomp > $OUT/p6_fc.log 2>&1
$S /tmp/p6_fc_stats/p_results.db | grep -v "rocclr" > $OUT/r06_fragcomp_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p6_mx_stats -o p -- python $REPO/tools/bench_legs.py multiplex > $OUT/p6_mx.log 2>&1
$S /tmp/p6_mx_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r06_multiplex_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p6_tr_stats -o p -- python $REPO/tools/bench_legs.py transfer > $OUT/p6_tr.log 2>&1
$S /tmp/p6_tr_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r06_transfer_kernel_stats.csv
ADH_BENCH_NO_CPU=1 rocprofv3 --kernel-trace --stats -d /tmp/p6_sel_stats -o p -- python $REPO/tools/bench_select.py > $OUT/p6_sel.log 2>&1
$S /tmp/p6_sel_stats/p_results.db | grep -v "rocprim\|rocclr" > $OUT/r06_selection_kernel_stats.csv
: > $OUT/r06_legs_pmc.csv
for c in FETCH_SIZE WRITE_SIZE; do
  FC_SIZES=1000000 rocprofv3 --pmc $c -d /tmp/p6_fc_$c -o p -- python $REPO/tools/bench_legs.py fragcomp > $OUT/p6_fc_$c.log 2>&1
  $S /tmp/p6_fc_$c/p_results.db | grep "adh_fc\|adh_fragcomp" > $OUT/p6_fc_$c.csv
  rocprofv3 --pmc $c -d /tmp/p6_mx_$c -o p -- python $REPO/tools/bench_legs.py multiplex > $OUT/p6_mx_$c.log 2>&1
  $S /tmp/p6_mx_$c/p_results.db | grep "adh_fused\|adh_gather_kernel\|adh_feature" > $OUT/p6_mx_$c.csv
  rocprofv3 --pmc $c -d /tmp/p6_tr_$c -o p -- python $REPO/tools/bench_legs.py transfer > $OUT/p6_tr_$c.log 2>&1
  $S /tmp/p6_tr_$c/p_results.db | grep "adh_gather_kernel\|adh_feature" > $OUT/p6_tr_$c.csv
  ADH_BENCH_NO_CPU=1 rocprofv3 --pmc $c -d /tmp/p6_sel_$c -o p -- python $REPO/tools/bench_select.py > $OUT/p6_sel_$c.log 2>&1
  $S /tmp/p6_sel_$c/p_results.db | grep "adh_select" > $OUT/p6_sel_$c.csv
done
cat $OUT/p6_fc_FETCH_SIZE.csv $OUT/p6_fc_WRITE_SIZE.csv > $OUT/p6_fc.csv
cat $OUT/p6_mx_FETCH_SIZE.csv $OUT/p6_mx_WRITE_SIZE.csv > $OUT/p6_mx.csv
cat $OUT/p6_sel_FETCH_SIZE.csv $OUT/p6_sel_WRITE_SIZE.csv > $OUT/p6_sel.csv
cat $OUT/p6_tr_FETCH_SIZE.csv $OUT/p6_tr_WRITE_SIZE.csv > $OUT/p6_tr.csv
cat $OUT/p6_fc.csv $OUT/p6_mx.csv $OUT/p6_sel.csv $OUT/p6_tr.csv > $OUT/r06_legs_pmc.csv
# passes: fragment competition 1 + 5 calls; configs[4] 4 + 5 host -> host + 6 resident; selection 1 + 3 calls;
# transfer requantification 3 + 5 host -> host calls
( cd $REPO && rm -f $OUT/legs_traffic.json && python tools/pmc_derive.py legs $OUT/legs_traffic.json \
    fragment_competition_1000000=$OUT/p6_fc.csv:6:adh_fc,adh_fragcomp multiplex_configs4=$OUT/p6_mx.csv:15:adh_:candidates@300000 \
    candidate_selection=$OUT/p6_sel.csv:4:adh_select transfer_requant=$OUT/p6_tr.csv:8:adh_:candidates@300000 )
rm -rf /tmp/p6_*

# ---- the bench line itself (driver style), with the traffic files of this run in place
cd $REPO
cp $OUT/timstof_traffic.json profiles/timstof_traffic.json
cp $OUT/legs_traffic.json profiles/legs_traffic.json
unset ADH_BENCH_NO_COMPACT
python bench.py --steps 20 --warmup 5 > $OUT/r06_final_bench.json 2> $OUT/r06_final_bench.log
tail -1 $OUT/r06_final_bench.json | cut -c1-400
grep -v "at::native\|rocprim\|rocclr" $OUT/r06_final_kernel_stats.csv | sed 's/(DevRun[^)]*)//; s/void //' | cut -c1-130 | head -10
cat $OUT/r06_valu_busy.json | head -40
cat $OUT/r06_mfma_im.json $OUT/r06_mfma_fdr.json | cut -c1-600
