# PMC view of the search kernels: pipe busy vs active cycles, clock, waits.  Usage: bash tools/pmc_r02l.sh <outdir> [variants...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --hash-clips 0 --no-windowed --ten-million 0 --no-valu --no-refs"
for v in "$@"; do IFS=: read lib k <<< "$v"
  cp $R/tools/_libvdf_$lib.so $R/vid_dup_finder_lib_amd/libvdf_hip.so
  export VDF_MFMA_KERNEL=$k
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_${lib}_k$k -- $B1 > /dev/null 2>&1
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${lib}_k$k -- $B1 > /dev/null 2>&1
done
cp $R/tools/_libvdf_default.so $R/vid_dup_finder_lib_amd/libvdf_hip.so
python3 - <<PY
import csv,glob,collections,os
O="$O"
t={}
for d in sorted(glob.glob(O+"/kt_*")):
    fs=glob.glob(d+"/*/*_kernel_stats.csv")
    if not fs: continue
    for r in csv.reader(open(fs[0])):
        if "hamming_mfma" in r[0] or "resolve_cand" in r[0] or "expand_fp4" in r[0]:
            print(os.path.basename(d), r[0][:46], "calls", r[1], "avg_ns", r[3]); 
            if "hamming_mfma" in r[0]: t[os.path.basename(d)[3:]]=float(r[3])*1e-9
for d in sorted(glob.glob(O+"/pmc_*")):
    fs=glob.glob(d+"/*/*_counter_collection.csv")
    if not fs: print(d,"no data"); continue
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if "hamming_mfma" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    key=os.path.basename(d)[4:]
    cyc=agg["GRBM_GUI_ACTIVE"]/8
    print(key, "active cycles %.4g  mfma busy/SIMD %.4g  pipe utilisation %.3f  clock %.3f GHz (kernel-trace time)  wait_any/wave_cycles %.3f wait_inst %.3f  valu/mfma %.2f" % (
        cyc, agg["SQ_VALU_MFMA_BUSY_CYCLES"]/1024, agg["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc, cyc/t.get(key,1)/1e9, agg["SQ_WAIT_ANY"]/agg["SQ_WAVE_CYCLES"], agg["SQ_WAIT_INST_ANY"]/agg["SQ_WAVE_CYCLES"], agg["SQ_INSTS_VALU"]/agg["SQ_INSTS_MFMA"]))
PY
