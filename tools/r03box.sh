#!/bin/bash
# boxes with side bars that share their column range through the per-wave kernel (one launch per range) against the gather kernel (VDF_NO_BOXSTREAM)
mkdir -p gpurun_out/r03box
VDF_FUZZ_SEEDS=40 python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py tests/test_gpu_hash_queue.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert|^FAILED" | head > gpurun_out/r03box/tests.log
cat gpurun_out/r03box/tests.log
for nb in 0 1; do
  if [ $nb = 1 ]; then export VDF_NO_BOXSTREAM=1; else unset VDF_NO_BOXSTREAM; fi
  for s in "1000 1920 1080 0.125 0 -" "2000 1280 720 0.125 0 -" "4000 854 480 0.12 0 -" "4000 640 360 0.125 0 -" "250 3840 2160 0.125 0 -" "1000 1920 1080 0.125 0.12 -" "1000 1920 1080 0.125 0.12 --mix" "2000 1280 720 0.125 0.12 --mix"; do
    set -- $s
    m=""; if [ "$6" = "--mix" ]; then m="--mix"; fi
    echo -n "no_boxstream=$nb $m " >> gpurun_out/r03box/ab.txt
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars $5 --side $4 $m --steps 5 2>&1 | grep "detect+crop" >> gpurun_out/r03box/ab.txt
  done
done
cat gpurun_out/r03box/ab.txt
