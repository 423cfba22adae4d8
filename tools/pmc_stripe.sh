set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stripe_pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --hash-clips 0 --no-windowed"
for cfg in "0 4096" "1 4096" "0 16384" "1 2048"; do
  set -- $cfg
  export VDF_MFMA_XCD_STRIPE=$1 VDF_MFMA_CHUNK_COLS=$2
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/s$1_c$2 -- $B1 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/*")):
    f=glob.glob(d+"/*/*counter_collection.csv")
    if not f: print(d,"no data"); continue
    acc=0.0
    for r in csv.DictReader(open(f[0])):
        if "hamming_mfma" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": acc+=float(r["Counter_Value"])
    print(d.split("/")[-1], "FETCH_SIZE x2 = %.1f GB per launch" % (acc*1024*2/1e9))
PY
