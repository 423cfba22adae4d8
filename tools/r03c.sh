#!/bin/bash
# round 3 GPU call: non-temporal variants of the frame streams (profiles/r03_hash_nt_ab.txt, first block), persistent-kernel occupancy
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_gpu_dup_heavy.py tests/test_gpu_hash_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
for rep in 1 2; do for v in default streamnt wident; do cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for shape in "100000 64 64" "100000 60 44" "1000 1920 1080" "4000 480 270" "250 3840 2160" "2000 1280 720" "1500 1536 864" "3000 1024 576" "3000 854 480" "20000 128 128"; do set -- $shape
    echo -n "$v: "; timeout 60 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 10 2>/dev/null | grep clips
  done; done; done | tee $O/hash_ab.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
for w in 2 3 4; do echo -n "wgs_per_cu=$w: "; VDF_HASH_WGS_PER_CU=$w timeout 60 python tools/bench_hash.py --clips 100000 --steps 20 2>/dev/null | grep clips; done | tee -a $O/hash_ab.txt
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --no-valu --c5-cands 0 --no-refs > $O/bench_dup.json 2> $O/bench_dup.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03c/bench_dup.json').read().strip().splitlines()[-1])
print("headline ms", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "suspects", d["suspects"])
x=d["dup_heavy"]; print(x["ms_per_call"], x["timing"], x["dense_over_sparse"], x["sparse_same_windows"]["ms_per_call"])
PY
