#!/bin/bash
# per-phase time of the ion-mobility feature kernel (ADH_DEBUG_IM stop points) on the reduced config-4 bench
for p in ${PHASES:-1 2 3 4 5 6 0}; do
  ADH_DEBUG_IM=$p ADH_BENCH_NO_CPU=1 python tools/bench_timstof.py 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('stop', $p, 'gather %.2f ms features %.2f ms' % (r['roofline']['gather_kernel_ms'], r['roofline']['feature_kernel_ms']))"
done
