#!/bin/bash
mkdir -p gpurun_out/r03u
python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py tests/test_gpu_hash_queue.py -m gpu -q -x -k "letterbox or cropped or bars or queue" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r03u/tests.log
for sz in "20000 64 64" "4000 640 360" "2000 1280 720" "1000 1920 1080" "250 3840 2160" "4000 854 480" "4000 240 426"; do
  set -- $sz
  for bars in 0.12 0.0; do
    echo "== bars=$bars $2x$3" >> gpurun_out/r03u/lb.log
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars $bars --steps 5 2>&1 | grep -v "amdgpu.ids\|crop.0" >> gpurun_out/r03u/lb.log
  done
done
cat gpurun_out/r03u/tests.log gpurun_out/r03u/lb.log
