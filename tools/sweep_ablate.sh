# timing-only ablations of the MFMA search kernel (results are wrong when VDF_MFMA_ABLATE != 0): VARIANT=w8s4 bash tools/sweep_ablate.sh
# The variant must be built with -DVDF_BENCH_ABLATE (tools/build_variant.sh w8s4 -DVDF_BENCH_ABLATE): the shipped library
# neither reads VDF_MFMA_ABLATE nor contains the ablated kernels.
cp tools/_libvdf_${VARIANT:-w8s4}.so vid_dup_finder_lib_amd/libvdf_hip.so
for ab in ${ABLATES:-0 1 2 3 4 5 0}; do
  echo -n "ablate=$ab "; VDF_MFMA_ABLATE=$ab VDF_MFMA_CHUNK_COLS=${CHUNK:-65536} timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hash-clips 0 --no-windowed | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['match_groups'])"
done
