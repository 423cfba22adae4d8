#!/bin/bash
# Throughput of the batching queue (N2) by frame size, caller threads and batch size (tools/bench_hash_queue.cpp).  GPU box, repo root.
# Usage: bash tools/sweep_hash_queue.sh <out_dir under gpurun_out>
O=gpurun_out/${1:-queue}; mkdir -p $O
g++ -O2 -std=c++17 -pthread -o tools/bench_hash_queue tools/bench_hash_queue.cpp -Lvid_dup_finder_lib_amd -lvdf_hip -Wl,-rpath,$PWD/vid_dup_finder_lib_amd \
    -Wl,-rpath-link,/opt/rocm/lib -Wl,--allow-shlib-undefined || exit 1
{
nproc
for cfg in "1920 1080 8 8" "1920 1080 16 8" "1920 1080 32 16" "1920 1080 64 16" "1920 1080 64 32" "1280 720 32 16" "1280 720 64 32" "854 480 64 32" "640 360 64 32" "64 64 64 64"; do
  set -- $cfg
  timeout 120 tools/bench_hash_queue $1 $2 $3 $4 2000 0 3
done
timeout 120 tools/bench_hash_queue 1920 1080 32 16 2000 1 3
timeout 120 tools/bench_hash_queue 1280 720 64 32 2000 1 3
} 2>&1 | grep -v amdgpu.ids | tee $O/queue.txt
