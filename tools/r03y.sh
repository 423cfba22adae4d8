#!/bin/bash
mkdir -p gpurun_out/r03y
VDF_FUZZ_SEEDS=60 python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py tests/test_gpu_hash_parity.py tests/test_gpu_hash_queue.py tests/test_golden.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert|^FAILED" | head -20 > gpurun_out/r03y/tests.log
cat gpurun_out/r03y/tests.log
python tools/sweep_wavestream_nw.py 462x260 500x282 512x288 528x297 576x324 640x360 720x405 768x432 854x480 896x504 960x540 1024x576 1152x648 1280x720 1312x738 1366x768 1440x810 1536x864 1600x900 1680x1050 1792x1008 1920x1080 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03y/nw_sweep.txt
for sz in "2000 768 432" "2000 1024 576" "1000 1536 864" "2000 1280 720" "4000 640 360" "2000 1152 648" "1000 1600 900"; do
  set -- $sz
  for nr in 0 1; do
    if [ $nr = 1 ]; then export VDF_NO_ROWCROP=1; else unset VDF_NO_ROWCROP; fi
    export VDF_ROWCROP_ALL=1
    echo "== no_rowcrop=$nr $2x$3" >> gpurun_out/r03y/lb.log
    python tools/bench_letterbox.py --clips $1 --w $2 --h $3 --bars 0.12 --steps 5 2>&1 | grep "detect+crop" >> gpurun_out/r03y/lb.log
  done
done
cat gpurun_out/r03y/lb.log
