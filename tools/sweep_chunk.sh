# A/B of prebuilt library variants (tools/_libvdf_<name>.so, see tools/build_variant.sh) x chunk widths, all inside one
# gpurun call (boxes differ by a few %).  VARIANTS="w4 w8" CHUNKS="4096 65536" REPS=2 bash tools/sweep_chunk.sh
for rep in $(seq 1 ${REPS:-2}); do
for v in ${VARIANTS:-w4 w8}; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for cc in ${CHUNKS:-4096 65536}; do
    echo -n "rep=$rep $v chunk=$cc "; VDF_MFMA_CHUNK_COLS=$cc timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hash-clips 0 --no-windowed | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['match_groups'])"
  done
done
done
