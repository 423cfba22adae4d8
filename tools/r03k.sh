#!/bin/bash
# round 3 GPU call: per-wave block streams - parity, soak and A/B against the chunk kernel (profiles/r03_wavestream_ab.txt)
O=gpurun_out/r03k; mkdir -p $O
python -m pytest tests/test_gpu_hash_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
python -m pytest tests/test_gpu_fuzz.py -k "soak or wide_frames or large_frames" -m gpu -x -q 2>&1 | tail -3 | tee -a $O/pytest.log
for rep in 1 2; do for e in VDF_NO_WAVESTREAM=1 VDF_X=0; do
  for shape in "1000 1920 1080" "1500 1440 1080" "1500 1600 900" "1200 1680 1050" "2000 1360 768"; do set -- $shape
    echo -n "$e: "; env $e timeout 60 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 10 2>/dev/null | grep clips
  done; done; done | tee $O/wavestream_ab.txt
