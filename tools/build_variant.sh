#!/bin/bash
# experimental build of the library beside the default one: tools/build_variant.sh NAME [hipcc flags]
# -> alphadia_amd/libalphadia_hip_NAME.so (git-ignored; selected at run time with ADH_LIB_PATH)
name=$1; shift
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$@" -o alphadia_amd/libalphadia_hip_$name.so alphadia_amd/csrc/adh_api.hip
