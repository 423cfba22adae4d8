#!/bin/bash
# round 3: ROWCROP instantiations of the chunk / K-split stream kernels - parity, letterbox A/B, uncropped regression check
mkdir -p gpurun_out/r03v
python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py tests/test_gpu_hash_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r03v/tests.log
cat gpurun_out/r03v/tests.log
for sz in "4000 640 360" "2000 1280 720" "2000 1024 576" "250 3840 2160" "500 2560 1440" "1000 1536 864" "4000 854 480" "500 1366 768" "1000 1920 1080"; do
  set -- $sz
  for nr in 0 1; do
    if [ $nr = 1 ]; then export VDF_NO_ROWCROP=1; else unset VDF_NO_ROWCROP; fi
    echo "== no_rowcrop=$nr $2x$3" >> gpurun_out/r03v/lb.log
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars 0.12 --steps 5 2>&1 | grep "detect+crop" >> gpurun_out/r03v/lb.log
  done
  unset VDF_NO_ROWCROP
  python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 5 2>&1 | grep clips >> gpurun_out/r03v/uncropped.log
done
cat gpurun_out/r03v/lb.log gpurun_out/r03v/uncropped.log
