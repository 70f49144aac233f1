#!/bin/bash
# what the occupancy of adh_gather_im_kernel is worth on configs[3]: ADH_DEBUG_IM_GATHER_LDS_PAD adds LDS per wavefront
export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=3 TOUCHED_SAMPLE=5 TOUCHED_SAMPLE_SEL=2 ADH_BENCH_NO_SELECT=1
for pad in 0 8000 20000 45000; do
  ADH_DEBUG_IM_GATHER_LDS_PAD=$pad python tools/bench_timstof.py > /tmp/gpad.json 2> /tmp/gpad.log
  python - /tmp/gpad.json $pad <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=r['roofline']
print('pad', sys.argv[2], 'gather %.3f ms features %.3f' % (rf['gather_kernel_ms'], rf['feature_kernel_ms']))
PY
done
