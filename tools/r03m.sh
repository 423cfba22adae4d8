#!/bin/bash
# round 3 GPU call: forced chunk widths on the windowed / dense / reference legs (profiles/r03_refs_chunk_sweep.txt)
O=gpurun_out/r03m; mkdir -p $O
for c in 0 4096 8192 16384 32768; do
  VDF_MFMA_CHUNK_COLS=$c timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hash-clips 0 --c4-hashes 0 --no-valu --c5-cands 0 > $O/b_$c.json 2>/dev/null
  python - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.loads(open(f'gpurun_out/r03m/b_{c}.json').read().strip().splitlines()[-1])
print("chunk", c, "headline", round(d["roofline"]["kernel_ms"],2), "windowed kernel", round(d["windowed"]["kernel_ms"],3), "ms", round(d["windowed"]["ms"],3), "waste", round(d["windowed"]["waste_ratio"],3),
      "| dup", round(d["dup_heavy"]["ms_per_call"],2), "sparse", round(d["dup_heavy"]["sparse_same_windows"]["ms_per_call"],2), "| refs", round(d["refs_c5_shape"]["ms"],3), "kernel", round(d["refs_c5_shape"]["kernel_ms"],3))
PY
done | tee $O/chunk_sweep.txt
