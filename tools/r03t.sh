#!/bin/bash
mkdir -p gpurun_out/r03t
for sz in "4000 640 360" "2000 1280 720" "2000 1024 576" "250 3840 2160" "500 2560 1440" "1000 1536 864" "4000 854 480"; do
  set -- $sz
  for bars in 0.12 0.0; do
    echo "== bars=$bars $2x$3" >> gpurun_out/r03t/lb.log
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars $bars --steps 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03t/lb.log
  done
done
cat gpurun_out/r03t/lb.log
