# A/B of prebuilt library variants on the large-frame hash path, inside one gpurun call.
# VARIANTS="base blk" bash tools/sweep_hash_sizes.sh
for v in ${VARIANTS:-base blk}; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for cfg in "20000 128 128" "5000 480 270" "2000 640 360" "1000 1280 720" "500 1920 1080" "2000 1920 1080"; do
    set -- $cfg
    echo -n "$v "; timeout 200 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 5
  done
done
