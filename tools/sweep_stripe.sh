# XCD striping x chunk width A/B for the MFMA search kernel, one gpurun call
for rep in 1 2; do
for st in 0 1; do
  for cc in ${CHUNKS:-2048 4096 8192 16384}; do
    echo -n "rep=$rep stripe=$st chunk=$cc "; VDF_MFMA_XCD_STRIPE=$st VDF_MFMA_CHUNK_COLS=$cc timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hash-clips 0 --no-windowed | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['match_groups'])"
  done
done
done
