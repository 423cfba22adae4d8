mkdir -p gpurun_out/r02k
B="timeout 90 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --ten-million 0 --no-valu"
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "pairs/s %.4g kernel_ms %.2f frac %.3f groups %s early %.4f" % (d["value"], r["kernel_ms"], r["frac"], d["match_groups"], r["early_exit"]["pairs_fraction"]))'
run() { cp tools/_libvdf_$1.so vid_dup_finder_lib_amd/libvdf_hip.so; n=$1; shift; env "$@" $B 2>/dev/null | python -c "$P" "$n $*"; }
{
run default VDF_MFMA_KERNEL=2
run default VDF_MFMA_KERNEL=1
run nocleanup VDF_MFMA_KERNEL=2
run default VDF_MFMA_KERNEL=1
run default VDF_MFMA_KERNEL=2
} 2>&1 | tee gpurun_out/r02k/ab.txt
cp tools/_libvdf_default.so vid_dup_finder_lib_amd/libvdf_hip.so
timeout 600 python -m pytest tests/test_gpu_search_parity.py tests/test_golden.py tests/test_gpu_multi_ctx.py -x -q -m gpu > gpurun_out/r02k/pytest.txt 2>&1; tail -4 gpurun_out/r02k/pytest.txt | head -3
