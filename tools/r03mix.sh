#!/bin/bash
# mixed batches (70 % no bars, 20 % top / bottom, 10 % side bars): per-clip split between the ROWCROP and the cropped stream kernel
# against the whole batch through the cropped stream kernel (VDF_NO_ROWCROP = what a batch with any side bar did before)
mkdir -p gpurun_out/r03mix
python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py tests/test_gpu_hash_queue.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert|^FAILED" | head > gpurun_out/r03mix/tests.log
cat gpurun_out/r03mix/tests.log
for sz in "1000 1920 1080" "2000 1280 720" "4000 640 360" "4000 854 480" "1000 1600 900" "250 3840 2160"; do
  set -- $sz
  for nr in 0 1; do
    if [ $nr = 1 ]; then export VDF_NO_ROWCROP=1; else unset VDF_NO_ROWCROP; fi
    echo -n "no_rowcrop=$nr " >> gpurun_out/r03mix/ab.txt
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars 0.12 --side 0.125 --mix --steps 5 2>&1 | grep "detect+crop" >> gpurun_out/r03mix/ab.txt
  done
done
cat gpurun_out/r03mix/ab.txt
