# early-exit step A/B for the MFMA search kernel (VDF_MFMA_PRUNE_STEP: 16 = off, -1/unset = from the tolerance)
for rep in 1 2; do
for v in ${VARIANTS:-pr6}; do
  cp tools/_libvdf_$v.so vid_dup_finder_lib_amd/libvdf_hip.so
  for st in ${STEPS:-16 13 12 11}; do
    echo -n "rep=$rep $v prune_step=$st "; VDF_MFMA_PRUNE_STEP=$st timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hash-clips 0 --no-windowed | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['match_groups'])"
  done
done
done
