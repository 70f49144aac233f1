#!/bin/bash
# cumulative time of the AlphaRaw feature kernels up to each stop point (ADH_DEBUG_STOP_PHASE) on the headline bench
for p in ${PHASES:-31 32 33 3 5 6 61 62 0}; do
  ADH_DEBUG_STOP_PHASE=$p ADH_BENCH_NO_CPU=1 python bench.py --steps 5 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stop', $p, 'gather %.2f ms features %.2f ms resident %.2f' % (d['roofline']['gather_kernel_ms'], d['roofline']['feature_kernel_ms'], d['resident']['ms_per_step']))"
done
