#!/bin/bash
# four launches per radix pass (tools/_libvdf_fourlaunch.so) against one scatter launch per pass with look-back (default): the sorts alone,
# the reference search of the configs[4] shape, and the dense-duplicate search.   bash tools/ab_sort2.sh <out_dir under gpurun_out>
O=gpurun_out/${1:-ab_sort2}; mkdir -p $O
for v in fourlaunch default fourlaunch default; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  echo "== variant $v"
  python tools/bench_sort.py 2>&1 | grep sort_order
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --c5-cands 0 --cache-entries 0 --no-valu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['refs_c5_shape']; print('refs_c5_shape ms', round(r['ms'],3), 'min', round(r['ms_min'],3), 'unpinned', round(r['unpinned']['ms'],3))
t=d['dup_heavy']['timing']; print('dup_heavy total', round(t['total_ms'],3), 'download', round(t['download_ms'],3))"
done 2>&1 | tee $O/ab_sort2.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
