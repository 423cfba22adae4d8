#!/bin/bash
# tools/build_variant_fast.sh <name> <file.hip> <EXTRA flags...>: recompiles ONE kernel file with the flags and links it with the default
# build's other objects into tools/_libvdf_<name>.so (A/B runs inside one gpurun call: copy it over vid_dup_finder_lib_amd/libvdf_hip.so).
set -e
name=$1; src=$2; shift; shift
cd "$(dirname "$0")/../vid_dup_finder_lib_amd/csrc"
make -j8 -s
obj=_build/_variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -mllvm -amdgpu-mfma-vgpr-form "$@" -c $src -o $obj
others=$(ls _build/*.o | grep -v "_variant_" | grep -v "_build/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_libvdf_$name.so $obj $others -ldl -lpthread
rm -f $obj
