#!/bin/bash
# Host (pageable) -> device hash path: staging threads x pinned chunk size, and the runtime's own pageable copy (DIRECT).
export HOST_PATH_SKIP_SEARCH=1
for t in 4 8 16 32; do for c in 8 32 64; do
  VDF_HOST_DIRECT=0 VDF_COPY_THREADS=$t VDF_HOST_CHUNK_MB=$c python tools/bench_host_path.py 2>/dev/null | grep -v amdgpu.ids
done; done
VDF_HOST_DIRECT=1 python tools/bench_host_path.py 2>/dev/null | grep -v amdgpu.ids
