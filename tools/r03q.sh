#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
for rep in 1 2; do for e in VDF_X=0 VDF_WAVESTREAM_NB3=1; do
  for shape in "2000 1280 720" "2500 1152 648" "3000 1056 594" "2500 1200 675"; do set -- $shape
    echo -n "$e: "; env $e timeout 60 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 10 2>/dev/null | grep clips
  done; done; done | tee $O/nb3_ab.txt
