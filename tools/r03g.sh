#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O
rocminfo 2>/dev/null | grep -i -E "Max Clock|Marketing|Compute Unit" | head -8 | tee $O/device.txt
python -m pytest tests/test_gpu_search_parity.py tests/test_gpu_dup_heavy.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_gpu_hash_parity.py tests/test_gpu_example_flow.py -k "not soak and not wide_frames" -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --no-valu --c5-cands 0 --dup-heavy 0 > $O/bench_refs.json 2> $O/bench_refs.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench_refs.json').read().strip().splitlines()[-1])
print(json.dumps(d["refs_c5_shape"], indent=1))
PY
