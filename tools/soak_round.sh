#!/bin/bash
# The unbounded differential sweeps and stresses on the shipped library, one line of verdict each (what profiles/r0N_diff_sweeps.txt records).
# Usage (GPU box, repo root): bash tools/soak_round.sh <out_dir under gpurun_out> [scale] [seed_base]     scale 1 ~ 4 GPU-minutes; seeds = seed_base + 61 .. 65
O=gpurun_out/${1:-soak}; S=${2:-1}; B=${3:-0}; mkdir -p $O
sha256sum vid_dup_finder_lib_amd/libvdf_hip.so | cut -c1-12 > $O/lib_sha.txt
run() { name=$1; shift; echo "== $name: $*"; timeout 3000 "$@" > $O/$name.log 2>&1; tail -1 $O/$name.log; }
{
run diff_sweep            python tools/diff_sweep.py --cases $((300 * S)) --seed $((B + 61))
run diff_sweep_search     python tools/diff_sweep_search.py --cases $((150 * S)) --seed $((B + 62))
run diff_sweep_letterbox  python tools/diff_sweep_letterbox.py --cases $((150 * S)) --seed $((B + 63))
run diff_sweep_lbhash     python tools/diff_sweep_letterbox_hash.py --cases $((80 * S)) --seed $((B + 64))
run diff_sweep_shards     python tools/diff_sweep_shards.py --cases $((100 * S)) --seed $((B + 65))
run stress_hash_queue     python tools/stress_hash_queue.py
for i in $(seq 1 $((3 * S))); do
  echo "== suite sweeps, repetition $i"
  python -m pytest tests/test_gpu_diff_sweep.py tests/test_gpu_letterbox_fused_small.py tests/test_gpu_radix_sort.py tests/test_gpu_path_order.py -m gpu -q 2>&1 | grep -E "passed|failed"
done
} 2>&1 | tee $O/soak.txt
