#!/bin/bash
# pillarboxed clips: general cropped kernels, whole-line (mode 0 at line-aligned pitches) against the cropped stream kernel (mode 5)
mkdir -p gpurun_out/r03pb
for m in 0 5; do
  export VDF_RESIZE_MODE=$m
  echo "== mode $m" >> gpurun_out/r03pb/ab.txt
  for s in "1000 1920 1080 0.125 0" "2000 1280 720 0.125 0" "4000 640 360 0.125 0" "250 3840 2160 0.125 0" "2000 1024 576 0.125 0" "1000 1536 864 0.125 0" "4000 854 480 0.12 0" "1000 1920 1080 0.125 0.12" "500 1366 768 0.125 0"; do
    set -- $s
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars $5 --side $4 --steps 5 2>&1 | grep "detect+crop" >> gpurun_out/r03pb/ab.txt
  done
done
cat gpurun_out/r03pb/ab.txt
