#!/bin/bash
# letterbox detect A/B: column batches of 8 against 16 strips (library variants built with tools/build_variant.sh)
mkdir -p gpurun_out/r03lb
for v in nc8 nc16 nc8 nc16; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  echo "== $v" >> gpurun_out/r03lb/ab.txt
  for s in "1000 1920 1080 0.125 0" "2000 1280 720 0.125 0" "4000 640 360 0.125 0" "250 3840 2160 0.125 0" "1000 1920 1080 0 0.12" "1000 1920 1080 0 0" "20000 64 64 0 0" "20000 64 64 0.13 0" "1024 1920 1080 0 0.12" "1100 1920 1080 0 0.12"; do
    set -- $s
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars $5 --side $4 --steps 5 2>&1 | grep -v "amdgpu\|crop.0" >> gpurun_out/r03lb/ab.txt
  done
done
cp tools/_libvdf_nc8.so vid_dup_finder_lib_amd/libvdf_hip.so
cat gpurun_out/r03lb/ab.txt
