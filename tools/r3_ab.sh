#!/bin/bash
# A/B of an experimental library build (ADH_LIB_PATH) against the default one: GPU suite + stop-phase times
echo "== default build"; PHASES="${PHASES:-22 2 0}" bash tools/feature_phases.sh
for v in "$@"; do
  lib=${v%%:*}; blk=${v##*:}
  echo "== $lib ADH_BLOCK_CYCLES=$blk"
  export ADH_LIB_PATH=$PWD/alphadia_amd/$lib ADH_BLOCK_CYCLES=$blk
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
  PHASES="${PHASES:-22 2 0}" bash tools/feature_phases.sh
done
