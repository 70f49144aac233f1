#!/bin/bash
# A/B of the wide register kernels (transfer-library requantification leg) across library variants
# (alphadia_amd/libalphadia_hip_<name>.so, tools/build_variant.sh): kernel ms of the leg, per-kernel averages from a
# traced run, and the bits (the wide kernels against the generic kernel, the golden parity tests) per variant.
# PMC=1 adds the instruction / wait counters of the default build.
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out
for v in default "$@"; do
  if [ "$v" = default ]; then unset ADH_LIB_PATH; else export ADH_LIB_PATH=$REPO/alphadia_amd/libalphadia_hip_$v.so; fi
  CPU_SAMPLE=20000 python $REPO/tools/bench_legs.py transfer > $OUT/wide_ab_$v.json 2> $OUT/wide_ab_$v.log
  python - $OUT/wide_ab_$v.json $v <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf=r['roofline']
print(sys.argv[2], 'ms_per_step %.2f kernel_ms %.3f gather %.3f features %.3f frac %.4f' % (r['ms_per_step'], rf['kernel_ms'], rf['gather_kernel_ms'], rf['feature_kernel_ms'], rf['frac']))
PY
  rm -rf /tmp/wab_stats_$v; CPU_SAMPLE=2000 rocprofv3 --kernel-trace --stats -d /tmp/wab_stats_$v -o p -- python $REPO/tools/bench_legs.py transfer > $OUT/wide_ab_trace_$v.log 2>&1
  python $REPO/tools/rocpd_summary.py /tmp/wab_stats_$v/p_results.db | grep "wide_kernel\|adh_gather_kernel" | awk -F, '{n=split($0,a,","); print "   ", substr($1,1,40), a[n-5], a[n-3]}'
  ( cd $REPO && python -m pytest tests -q -m gpu -k "wide_register or golden_inputs or reference_goldens or transfer" 2>&1 | tail -1 )
done
if [ -n "$PMC" ]; then
  unset ADH_LIB_PATH
  [ "$PMC" != 1 ] && export ADH_LIB_PATH=$REPO/alphadia_amd/libalphadia_hip_$PMC.so
  rm -rf /tmp/wab_pmc1 /tmp/wab_pmc2
  CPU_SAMPLE=2000 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d /tmp/wab_pmc1 -o p -- python $REPO/tools/bench_legs.py transfer > $OUT/wide_ab_pmc1.log 2>&1
  CPU_SAMPLE=2000 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d /tmp/wab_pmc2 -o p -- python $REPO/tools/bench_legs.py transfer > $OUT/wide_ab_pmc2.log 2>&1
  for i in 1 2; do python $REPO/tools/rocpd_summary.py /tmp/wab_pmc$i/p_results.db | grep "wide_kernel\|adh_gather_kernel"; done > $OUT/wide_ab_pmc.csv
  cat $OUT/wide_ab_pmc.csv | awk -F, '{n=split($0,a,","); print substr($1,1,36), a[n-3], a[n-2], a[n-1]}'
fi
