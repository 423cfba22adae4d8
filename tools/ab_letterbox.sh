#!/bin/bash
# A/B of the letterbox path on the GPU box: library variants (tools/build_variant.sh) over the bench's shapes.
# Usage: bash tools/ab_letterbox.sh <out_dir under gpurun_out> <variant> [<variant> ...]   (the variant "default" must exist and is restored at the end)
O=gpurun_out/${1:-ab_lb}; shift; mkdir -p $O
run() { timeout 120 python tools/bench_letterbox.py --steps 5 "$@" 2>&1 | grep -E "detect|crop\[0\]" | sed "s/^/    /"; }
for rep in 1 2; do
for v in "$@"; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for ch in 1; do
    echo "== variant $v (rep $rep)"
    run --clips 1000 --w 1920 --h 1080 --bars 0 --side 0.125
    run --clips 1000 --w 1920 --h 1080 --bars 0.12
    run --clips 1000 --w 1920 --h 1080 --bars 0
    run --clips 1000 --w 1920 --h 1080 --bars 0 --black 0.001
    run --clips 1000 --w 1920 --h 1080 --bars 0.12 --side 0.125 --mix
    run --clips 2000 --w 1280 --h 720 --bars 0 --side 0.125
    if [ $ch = 1 ]; then
      run --clips 20000 --w 64 --h 64 --bars 0
      run --clips 20000 --w 64 --h 64 --bars 0.12
      run --clips 20000 --w 64 --h 64 --bars 0 --side 0.125
      run --clips 4000 --w 640 --h 360 --bars 0.12
    fi
  done
done
done 2>&1 | tee $O/ab_letterbox.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
