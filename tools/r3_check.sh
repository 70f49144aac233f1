#!/bin/bash
# round-3 GPU check: reciprocal probe, the GPU suite, a short bench (gpurun, from the repo root)
mkdir -p gpurun_out
( cd tools/probes && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/recip_probe recip_probe.hip 2>/dev/null && /tmp/recip_probe ) > gpurun_out/recip_probe.txt 2>&1
cat gpurun_out/recip_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.log
tail -1 gpurun_out/r3_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'], d.get('resident'))"
