#!/bin/bash
# round 3: row-cropped per-wave kernel - parity + A/B
mkdir -p gpurun_out/r03s
python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py -m gpu -q -x -k "letterbox or cropped or soak or bars" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r03s/tests.log
for bars in 0.12 0.0; do
  for nw in 0 1; do
    for sz in "1920 1080" "1600 900"; do
      set -- $sz
      if [ $nw = 1 ]; then export VDF_NO_WAVESTREAM=1; else unset VDF_NO_WAVESTREAM; fi
      echo "== no_wavestream=$nw bars=$bars $1x$2" >> gpurun_out/r03s/ab.log
      python tools/bench_letterbox.py --clips 1000 --w $1 --h $2 --bars $bars --steps 5 >> gpurun_out/r03s/ab.log 2>&1
    done
  done
done
cat gpurun_out/r03s/tests.log gpurun_out/r03s/ab.log
