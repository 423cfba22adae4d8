mkdir -p gpurun_out/r02g
./tools/ubench_mfma_shapes 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02g/ubench_shapes.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --ten-million 0 --no-valu"
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "pairs/s %.4g kernel_ms %.2f frac %.3f alg %.3f groups %s early %.4f" % (d["value"], r["kernel_ms"], r["frac"], r["algorithmic_frac"], d["match_groups"], r["early_exit"]["pairs_fraction"]))'
run() { cp tools/_libvdf_$1.so vid_dup_finder_lib_amd/libvdf_hip.so; shift; env "$@" $B 2>/dev/null | python -c "$P" "$*"; }
{
run default VDF_MFMA_KERNEL=1
run default VDF_MFMA_KERNEL=2
run default VDF_MFMA_KERNEL=2 VDF_MFMA_PRUNE_STEP=13
run nocleanup VDF_MFMA_KERNEL=2
run nocleanup VDF_MFMA_KERNEL=2 VDF_MFMA_PRUNE_STEP=13
run default VDF_MFMA_KERNEL=1
run default VDF_MFMA_KERNEL=2
} | tee gpurun_out/r02g/ab.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
