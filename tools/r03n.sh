#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests/test_gpu_search_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_multi_ctx.py tests/test_gpu_example_flow.py -k "not hash and not soak and not ten_million" -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --no-valu --dup-heavy 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03n/bench.json').read().strip().splitlines()[-1])
r=d["refs_c5_shape"]; print("refs", round(r["ms"],3), r["kernel_ms"], {k:round(v,3) for k,v in r["timing"].items()}, "unpinned", round(r["unpinned"]["ms"],3), "waste", r["waste_ratio"], "wgs", r["workgroups"])
c5=d["c5_end_to_end"]; print("c5", round(c5["ms_per_step"],2), {k:round(v,2) for k,v in c5["phases_ms"].items()}, c5["search_kernel_ms"])
PY
