#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
for cfg in "VDF_X=0" "VDF_MFMA_SELF_ROWS=256" "VDF_MFMA_SELF_ROWS=256 VDF_MFMA_CHUNK_COLS=8192" "VDF_MFMA_SELF_ROWS=256 VDF_MFMA_CHUNK_COLS=16384" "VDF_MFMA_SELF_ROWS=256 VDF_MFMA_CHUNK_COLS=4096"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hash-clips 0 --c4-hashes 0 --no-valu --c5-cands 0 --no-refs > $O/b.json 2>/dev/null
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03o/b.json').read().strip().splitlines()[-1])
print(sys.argv[1], "| headline", round(d["roofline"]["kernel_ms"],2), "| windowed kernel", round(d["windowed"]["kernel_ms"],3), "ms", round(d["windowed"]["ms"],3), "waste", round(d["windowed"]["waste_ratio"],3),
      "| dup", round(d["dup_heavy"]["ms_per_call"],2), "sparse", round(d["dup_heavy"]["sparse_same_windows"]["ms_per_call"],2))
PY
done | tee $O/self_rows_sweep.txt
