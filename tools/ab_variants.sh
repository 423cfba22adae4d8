# A/B of library builds made with tools/build_variant.sh <name> <flags>: bash tools/ab_variants.sh <name> <name> ... (each run under its own timeout)
mkdir -p gpurun_out/ab_variants
B="timeout 90 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --ten-million 0 --no-valu --no-refs"
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "pairs/s %.4g kernel_ms %.2f frac %.3f groups %s" % (d["value"], r["kernel_ms"], r["frac"], d["match_groups"]))'
run() { cp tools/_libvdf_$1.so vid_dup_finder_lib_amd/libvdf_hip.so; n=$1; shift; env "$@" $B 2>/dev/null | python -c "$P" "$n $*"; }
{
for v in "$@" default; do run $v VDF_SEARCH_BACKEND=mfma; done
} 2>&1 | tee gpurun_out/ab_variants/ab.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
