# A/B of two library builds on the hash kernels: tools/ab_hash.sh <variantA> <variantB>
mkdir -p gpurun_out/r02p
VARS="$*"
for rep in 1 2; do for v in $VARS; do cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for shape in "100000 64 64" "400000 16 16" "20000 128 128" "4000 480 270"; do set -- $shape
    echo -n "$v: "; timeout 60 python tools/bench_hash.py --clips $1 --w $2 --h $3 --steps 10 2>/dev/null | grep clips
  done; done; done | tee gpurun_out/r02p/hash_ab.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
