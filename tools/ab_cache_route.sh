#!/bin/bash
# bench.py's cache_ingest leg (10 M entries) with the path order on the device (default) and on the host (VDF_NO_DEVICE_PATH_ORDER, the route
# before round 6).   bash tools/ab_cache_route.sh <out_dir under gpurun_out>
O=gpurun_out/${1:-ab_cache}; mkdir -p $O
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --c5-cands 0 --dup-heavy 0 --no-valu --no-refs"
for rep in 1 2; do
  for v in device host; do
    if [ $v = host ]; then export VDF_NO_DEVICE_PATH_ORDER=1; else unset VDF_NO_DEVICE_PATH_ORDER; fi
    echo "== path order on the $v (rep $rep)"
    $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['cache_ingest']
print({k: round(v,1) for k,v in d['search_cache_entries'].items()}, 'host_ms', round(d['host_ms'],1), 'first_call', round(d['first_call_total_ms'],1), 'groups', d['match_groups'], d['planted_pairs'])"
  done
done 2>&1 | tee $O/ab_cache_route.txt
