#!/bin/bash
mkdir -p gpurun_out/r03w
python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py tests/test_gpu_hash_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r03w/tests.log
cat gpurun_out/r03w/tests.log
for sz in "2000 768 432" "2000 1024 576" "1000 1536 864" "1000 1792 1008" "500 2048 1152" "500 1366 768" "1000 1600 900" "1000 1920 1080" "2000 896 504" "2000 1152 648"; do
  set -- $sz
  for nr in 0 1; do
    if [ $nr = 1 ]; then export VDF_NO_ROWCROP=1; else unset VDF_NO_ROWCROP; fi
    echo "== no_rowcrop=$nr $2x$3" >> gpurun_out/r03w/lb.log
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars 0.12 --steps 5 2>&1 | grep "detect+crop" >> gpurun_out/r03w/lb.log
  done
  unset VDF_NO_ROWCROP
  python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 5 2>&1 | grep clips >> gpurun_out/r03w/uncropped.log
done
cat gpurun_out/r03w/lb.log gpurun_out/r03w/uncropped.log
