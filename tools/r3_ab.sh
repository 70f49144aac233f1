#!/bin/bash
# same-box A/B of library builds: tools/r3_ab.sh NAME... (alphadia_amd/libalphadia_hip_NAME.so; "default" = the product build)
# env: PHASES (stop points), ADH_DEBUG_ONLY etc. pass through to the timed runs only
for v in "$@"; do
  echo "== $v"
  if [ "$v" = default ]; then unset ADH_LIB_PATH; else export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_$v.so; fi
  if [ -n "$AB_TESTS" ]; then ( unset ADH_DEBUG_ONLY; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ); fi
  PHASES="${PHASES:-2 0}" bash tools/feature_phases.sh
done
