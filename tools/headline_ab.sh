#!/bin/bash
# A/B of the fused kernel's compile-time switches on the headline workload: one bench.py run per library variant
# (alphadia_amd/libalphadia_hip_<name>.so, tools/build_variant.sh), kernel ms from the bench line; the variants must
# give the default's bits (the golden parity tests run against each).
mkdir -p gpurun_out
for v in default "$@"; do
  if [ "$v" = default ]; then unset ADH_LIB_PATH; else export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/headline_ab_$v.json 2> gpurun_out/headline_ab_$v.log
  python - gpurun_out/headline_ab_$v.json $v <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'ms_per_step %.2f kernel_ms %.3f frac %.4f resident %.2f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'], r.get('resident',{}).get('ms_per_step', float('nan'))))
PY
  if [ "$v" != default ]; then python -m pytest tests/test_gpu_parity.py -q -m gpu -k "golden_inputs or reference_goldens" 2>&1 | tail -1; fi
done
