#!/bin/bash
# what is occupancy worth to the ion-mobility kernels?  extra LDS per block -> fewer resident waves
for pad in 0 14000 27000 54000; do
  ADH_DEBUG_IM_FEATURE_LDS_PAD=$pad ADH_BENCH_NO_CPU=1 python tools/bench_timstof.py 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('feature pad', $pad, 'gather %.2f ms features %.2f ms' % (r['roofline']['gather_kernel_ms'], r['roofline']['feature_kernel_ms']))"
done
for pad in 10000 30000; do
  ADH_DEBUG_IM_GATHER_LDS_PAD=$pad ADH_BENCH_NO_CPU=1 python tools/bench_timstof.py 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('gather pad', $pad, 'gather %.2f ms features %.2f ms' % (r['roofline']['gather_kernel_ms'], r['roofline']['feature_kernel_ms']))"
done
