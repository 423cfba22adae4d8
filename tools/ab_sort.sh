#!/bin/bash
# rocPRIM's radix sort (tools/_libvdf_rocprimsort.so: round 5's sort_order.hip linked with today's other objects) against the hand-written one:
# the sorts alone, a one-shot caller's first call, and the legs that sort inside a search.   bash tools/ab_sort.sh <out_dir under gpurun_out>
O=gpurun_out/${1:-ab_sort}; mkdir -p $O
for v in rocprimsort default; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  echo "== variant $v ($(stat -c %s vid_dup_finder_lib_amd/libvdf_hip.so) bytes)"
  python tools/bench_sort.py 2>&1 | grep sort_order
  for i in 1 2 3; do python tools/first_call.py 2>&1 | grep "n="; done
done 2>&1 | tee $O/ab_sort.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
