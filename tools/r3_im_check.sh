#!/bin/bash
# ion-mobility kernels: parity tests (goldens, fuzz seeds, index / tile variants), then phase timings of the
# reduced configs[3] run (gpurun, from the repo root)
mkdir -p gpurun_out
ADH_FUZZ_SEEDS_IM=${SEEDS:-30} timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pipeline_gpu.py -m gpu -x -q -k "timstof" 2>&1 | tail -8
PHASES="${PHASES:-1 3 4 5 6 0}" bash tools/im_phases.sh 2>&1 | tee gpurun_out/im_phases_now.txt
