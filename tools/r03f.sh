#!/bin/bash
# round 3 GPU call: re-pitch A/B (profiles/r03_repitch_ab.txt), the resize-mode sweep (profiles/r03_resize_sweep.txt), dup_heavy
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests/test_gpu_dup_heavy.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for v in default repitch; do cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for shape in "1000 1920 1080" "4000 640 360" "2000 1280 720" "1200 1152 648" "3000 896 504"; do set -- $shape
    echo -n "$v: "; timeout 60 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 10 2>/dev/null | grep clips
  done; done; done | tee $O/repitch_ab.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
timeout 1500 python tools/sweep_resize_modes.py --modes 0,2,4,5,6 --mb 3000 > $O/resize_sweep.txt 2>&1
cat $O/resize_sweep.txt
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --no-valu --c5-cands 0 --no-refs > $O/bench_dup.json 2> $O/bench_dup.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f/bench_dup.json').read().strip().splitlines()[-1])
x=d["dup_heavy"]; print(x["ms_per_call"], x["timing"], x["dense_over_sparse"], x["sparse_same_windows"]["ms_per_call"])
PY
