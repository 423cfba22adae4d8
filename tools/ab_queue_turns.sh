#!/bin/bash
# Link turns (hash_host.cpp) against interleaved transfers (VDF_NO_LINK_TURNS=1 = before), alternating runs.  GPU box, repo root.
g++ -O2 -std=c++17 -pthread -o tools/bench_hash_queue tools/bench_hash_queue.cpp -Lvid_dup_finder_lib_amd -lvdf_hip -Wl,-rpath,$PWD/vid_dup_finder_lib_amd -Wl,-rpath-link,/opt/rocm/lib -Wl,--allow-shlib-undefined || exit 1
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), nproc $(nproc)"
for rep in 1 2; do
for cfg in "1920 1080 32 16 2000 1" "1920 1080 32 16 2000 0" "1920 1080 64 16 2000 1" "1280 720 64 32 2000 1" "1280 720 64 32 2000 0" "854 480 64 32 2000 0" "640 360 64 32 2000 1" "64 64 64 64 2000 0" "64 64 64 64 2000 1"; do
  echo -n "turns:    "; tools/bench_hash_queue $cfg 2 2>&1 | grep -v amdgpu | cut -d'|' -f1
  echo -n "no turns: "; VDF_NO_LINK_TURNS=1 tools/bench_hash_queue $cfg 2 2>&1 | grep -v amdgpu | cut -d'|' -f1
done; done
