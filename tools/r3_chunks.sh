#!/bin/bash
# host -> host step and kernel time against the chunk size of the pipeline (ADH_CHUNK)
for c in ${CHUNKS:-262144 393216 524288 786432 1048576}; do
  ADH_CHUNK=$c python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('chunk', $c, 'h2h %.2f ms' % d['ms_per_step'], 'kernels %.2f ms' % d['roofline']['kernel_ms'], 'launches', d['roofline']['launches_per_step'])"
done
