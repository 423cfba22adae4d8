#!/bin/bash
# A/B of library variants on the small-frame letterbox bench: bash tools/ab_small.sh <out_dir under gpurun_out> <variant> ...
O=gpurun_out/${1:-ab_small}; shift; mkdir -p $O
for v in "$@"; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  echo "== variant $v"
  timeout 300 python tools/bench_letterbox_small.py --child --clips 20000 --steps 20 --check ${CHECK:-0} 2>&1 | grep -E "plain|FAIL|Error"
done 2>&1 | tee $O/ab_small.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
