#!/bin/bash
# round 3 GPU call: parity of the reworked hit path, the 64x64 kernel's load variants (profiles/r03_hash_nt_ab.txt, second block), dup_heavy
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_gpu_dup_heavy.py tests/test_gpu_search_parity.py tests/test_gpu_multi_ctx.py tests/test_gpu_fuzz.py -k "not hash and not stream_kernels" -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
cat $O/pytest.log
for rep in 1 2; do for v in default nt contig ntcontig; do cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for shape in "100000 64 64" "200000 32 48" "20000 128 128"; do set -- $shape
    echo -n "$v: "; timeout 60 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 20 2>/dev/null | grep clips
  done; done; done | tee $O/hash_ab.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
for w in 2 3 4; do echo -n "wgs_per_cu=$w: "; VDF_HASH_WGS_PER_CU=$w timeout 60 python tools/bench_hash.py --clips 100000 --steps 20 2>/dev/null | grep clips; done | tee -a $O/hash_ab.txt
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --no-valu --c5-cands 0 > $O/bench_dup.json 2> $O/bench_dup.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b/bench_dup.json').read().strip().splitlines()[-1])
print("headline ms", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
print(json.dumps(d["dup_heavy"], indent=1)[:3000])
print(json.dumps(d["refs_c5_shape"], indent=1)[:1500])
PY
