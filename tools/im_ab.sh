#!/bin/bash
# configs[3] at full size: the tile kernels four candidates per wavefront (default) against the one-candidate kernel
# (ADH_DEBUG_IM_TILE1=1); extra variants as "name VAR=value ..." lines in $IM_AB_VARIANTS
export N_PREC=${N_PREC:-200000} N_CYCLES=${N_CYCLES:-2000} SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=${STEPS:-5} TOUCHED_SAMPLE=20
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python tools/bench_timstof.py > gpurun_out/im_ab_$name.json 2> gpurun_out/im_ab_$name.log; }
run tile4 A=1
run tile1 ADH_DEBUG_IM_TILE1=1
if [ -n "$IM_AB_VARIANTS" ]; then
  while read -r line; do [ -n "$line" ] && run $line; done <<< "$IM_AB_VARIANTS"
fi
for f in gpurun_out/im_ab_*.json; do python - $f <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'h2h %.2f resident %.2f gather %.2f features %.2f valid %.3f' % (r['ms_per_step'], r['resident']['ms_per_step'], r['roofline']['gather_kernel_ms'], r['roofline']['feature_kernel_ms'], r['valid_fraction']))
PY
done
