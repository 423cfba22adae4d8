# A/B of environment settings on the default library: tools/ab_env.sh "ENV1=.. ENV2=.." "ENV..." ...
mkdir -p gpurun_out/r02q
B="timeout 90 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --ten-million 0 --no-valu --no-refs"
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "pairs/s %.4g kernel_ms %.2f frac %.3f groups %s" % (d["value"], r["kernel_ms"], r["frac"], d["match_groups"]))'
for rep in 1 2; do for e in "$@"; do env $e $B 2>/dev/null | python -c "$P" "$e"; done; done | tee gpurun_out/r02q/ab_env.txt
