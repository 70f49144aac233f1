#!/bin/bash
# same-box A/B of environment settings for one library build: tools/r3_env_ab.sh LIBNAME "ENV=1 ENV2=2" "..." ...
lib=$1; shift
if [ "$lib" != default ]; then export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_$lib.so; fi
for e in "$@"; do
  echo "== $lib $e"
  env $e PHASES="${PHASES:-2 0}" bash tools/feature_phases.sh
done
