#!/bin/bash
# round 3 GPU call: per-wave block streams with the re-pitching DMA modes - parity, soak and A/B (profiles/r03_wavestream_ab.txt, second block)
O=gpurun_out/r03p; mkdir -p $O
python -m pytest tests/test_gpu_hash_parity.py tests/test_golden.py tests/test_gpu_letterbox.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tee $O/pytest.log
python -m pytest tests/test_gpu_fuzz.py -k "soak or wide_frames or large_frames" -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tee -a $O/pytest.log
for rep in 1 2; do for e in VDF_NO_WAVESTREAM=1 VDF_X=0; do
  for shape in "2000 1366 768" "1500 1536 864" "1200 1792 1008" "1000 1916 1080" "1000 1920 1080"; do set -- $shape
    echo -n "$e: "; env $e timeout 60 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 10 2>/dev/null | grep clips
  done; done; done | tee $O/wavestream_modes_ab.txt
