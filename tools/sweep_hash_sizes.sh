# A/B on the large-frame hash path inside one gpurun call: prebuilt library variants (VARIANTS, tools/_libvdf_<v>.so; "cur" =
# the library in the tree) x VDF_HASH_FRAME_PER_WAVE settings (MODES).
cp vid_dup_finder_lib_amd/libvdf_hip.so /tmp/_cur.so
for v in ${VARIANTS:-cur}; do
  if [ $v = cur ]; then cp /tmp/_cur.so vid_dup_finder_lib_amd/libvdf_hip.so; else cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so; fi
  for m in ${MODES:-0}; do
  for cfg in ${CFGS:-"20000 128 128" "5000 480 270" "2000 640 360" "1000 1280 720" "500 1920 1080" "2000 1920 1080"}; do
    set -- $cfg
    echo -n "$v fpw=$m "; VDF_HASH_FRAME_PER_WAVE=$m timeout 200 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 5
  done
  done
done
cp /tmp/_cur.so vid_dup_finder_lib_amd/libvdf_hip.so
