#!/bin/bash
# configs[3] at full size: the scoring gather of library variants (alphadia_amd/libalphadia_hip_<name>.so), whole and
# stopped behind fragment selection + window limits (ADH_DEBUG_IM=7); the ion-mobility GPU tests run against each variant
export N_PREC=${N_PREC:-200000} N_CYCLES=${N_CYCLES:-2000} SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=${STEPS:-5} TOUCHED_SAMPLE=5 TOUCHED_SAMPLE_SEL=2 ADH_BENCH_NO_SELECT=1
mkdir -p gpurun_out
for v in default "$@"; do
  if [ "$v" = default ]; then unset ADH_LIB_PATH; else export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_$v.so; fi
  for stop in 0 7; do
    ADH_DEBUG_IM=$stop python tools/bench_timstof.py > gpurun_out/im_gab_${v}_$stop.json 2> gpurun_out/im_gab_${v}_$stop.log
    python - gpurun_out/im_gab_${v}_$stop.json $v $stop <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'stop', sys.argv[3], 'h2h %.2f gather %.3f features %.3f valid %.3f' % (r['ms_per_step'], r['roofline']['gather_kernel_ms'], r['roofline']['feature_kernel_ms'], r['valid_fraction']))
PY
  done
  if [ "$v" != default ]; then python -m pytest tests -q -m gpu -k "timstof or ion_mob or im_ or selection" 2>&1 | tail -1; fi
done
