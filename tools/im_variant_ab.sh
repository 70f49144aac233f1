#!/bin/bash
# configs[3] at full size, traced: per-kernel averages of the ion-mobility feature kernels for library variants
# (alphadia_amd/libalphadia_hip_<name>.so); the ion-mobility GPU tests run against each variant
export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=2 TOUCHED_SAMPLE=5 TOUCHED_SAMPLE_SEL=2 ADH_BENCH_NO_SELECT=1
for v in default "$@"; do
  if [ "$v" = default ]; then unset ADH_LIB_PATH; else export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_$v.so; fi
  rm -rf /tmp/imv_stats
  (cd /tmp; TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/imv_stats -o p -- python $OLDPWD/tools/bench_timstof.py > /tmp/imv_$v.json 2> /tmp/imv_$v.log)
  echo "$v: $(python tools/rocpd_summary.py /tmp/imv_stats/p_results.db | grep 'fused4\|tile4_kernel\|profiles_kernel' | awk -F, '{n=split($0,a,","); printf "%s %.0f us; ", substr($1,1,34), a[n-3]/1e3}')"
  python - /tmp/imv_$v.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=r['roofline']
print('   h2h %.2f gather %.3f features %.3f' % (r['ms_per_step'], rf['gather_kernel_ms'], rf['feature_kernel_ms']))
PY
  if [ "$v" != default ]; then python -m pytest tests -q -m gpu -k "timstof or ion_mob or im_" 2>&1 | tail -1; fi
done
