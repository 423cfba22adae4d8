#!/bin/bash
# three-wave per-wave streams (1936 .. 2368 columns): parity + A/B against what those widths took before (K-split / whole-line)
mkdir -p gpurun_out/r03nw3
python -m pytest tests/test_gpu_hash_parity.py tests/test_gpu_fuzz.py tests/test_gpu_letterbox.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert|^FAILED" | head > gpurun_out/r03nw3/tests.log
cat gpurun_out/r03nw3/tests.log
for sz in "500 2048 1152" "500 2048 1080" "300 2160 3840" "500 1936 1089" "500 2000 1125" "500 1950 1096" "400 2304 1296" "400 2352 1323"; do
  set -- $sz
  for nw in 0 1; do
    if [ $nw = 1 ]; then export VDF_NO_WAVESTREAM=1; else unset VDF_NO_WAVESTREAM; fi
    echo -n "no_wavestream=$nw " >> gpurun_out/r03nw3/ab.txt
    python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 5 2>&1 | grep clips >> gpurun_out/r03nw3/ab.txt
  done
done
unset VDF_NO_WAVESTREAM
for nr in 0 1; do
  if [ $nr = 1 ]; then export VDF_NO_ROWCROP=1; else unset VDF_NO_ROWCROP; fi
  export VDF_ROWCROP_ALL=1
  echo -n "no_rowcrop=$nr " >> gpurun_out/r03nw3/ab.txt
  python tools/bench_letterbox.py --clips 500 --w 2048 --h 1152 --bars 0.12 --steps 5 2>&1 | grep "detect+crop" >> gpurun_out/r03nw3/ab.txt
done
cat gpurun_out/r03nw3/ab.txt
