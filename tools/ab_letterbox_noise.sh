#!/bin/bash
# A/B of the letterbox path on noisy bars (16 + U{0..3}: what a lossy codec leaves of a black bar) beside the clean shapes.
# Usage: bash tools/ab_letterbox_noise.sh <out_dir under gpurun_out> <variant> [<variant> ...]   (tools/_libvdf_<variant>.so; "default" is restored at the end)
O=gpurun_out/${1:-ab_lbn}; shift; mkdir -p $O
run() { timeout 120 python tools/bench_letterbox.py --steps 5 "$@" 2>&1 | grep -E "detect|crop\[0\]" | sed "s/^/    /"; }
for rep in 1 2; do
for v in "$@"; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  echo "== variant $v (rep $rep)"
  for nz in 0 3; do
    echo "  -- noise $nz"
    run --clips 1000 --w 1920 --h 1080 --bars 0 --side 0.125 --noise $nz
    run --clips 1000 --w 1920 --h 1080 --bars 0.12 --noise $nz
    run --clips 1000 --w 1920 --h 1080 --bars 0.12 --side 0.125 --mix --noise $nz
    run --clips 2000 --w 1280 --h 720 --bars 0 --side 0.125 --noise $nz
    run --clips 4000 --w 640 --h 360 --bars 0.12 --noise $nz
    run --clips 20000 --w 64 --h 64 --bars 0.12 --side 0.125 --noise $nz
  done
  run --clips 1000 --w 1920 --h 1080 --bars 0
  run --clips 1000 --w 1920 --h 1080 --bars 0 --black 0.001
done
done 2>&1 | tee $O/ab_letterbox_noise.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
