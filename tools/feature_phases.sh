#!/bin/bash
# cumulative kernel time up to each stop point (ADH_DEBUG_STOP_PHASE) on the headline bench.
# Fused kernel: 2 = selection + gather, 31 = rows + template, 32 = tables, 33 = row sums + centre means,
# 3 = intensities, 5 = envelope + quantification, 6 = feature assembly, 61 = median, 62 = correlations, 0 = all
for p in ${PHASES:-2 31 32 33 3 5 6 61 62 0}; do
  ADH_DEBUG_STOP_PHASE=$p python bench.py --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stop', $p, 'gather %.2f ms features %.2f ms resident %.2f' % (d['roofline']['gather_kernel_ms'], d['roofline']['feature_kernel_ms'], d['resident']['ms_per_step']))"
done
