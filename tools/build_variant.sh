#!/bin/bash
# tools/build_variant.sh <name> <EXTRA flags...>: builds libvdf_hip.so with the flags into tools/_libvdf_<name>.so
# (A/B experiments inside one gpurun call; the default library is rebuilt afterwards with `make clean; make`).
set -e
name=$1; shift
cd "$(dirname "$0")/../vid_dup_finder_lib_amd/csrc"
make clean > /dev/null
make -j8 EXTRA="$*" 2>&1 | grep -E "error|warning: v" || true
cp ../libvdf_hip.so ../../tools/_libvdf_$name.so
