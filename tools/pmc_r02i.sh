# PMC comparison of the two MFMA search kernels (and the cleanup-free stream): pipe busy cycles vs active cycles, waits.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02i; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --hash-clips 0 --no-windowed --ten-million 0 --no-valu"
for v in "default 1" "default 2" "nc_p4d13 2"; do set -- $v
  cp $R/tools/_libvdf_$1.so $R/vid_dup_finder_lib_amd/libvdf_hip.so
  export VDF_MFMA_KERNEL=$2
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_$1_k$2 -- $B1 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc2_$1_k$2 -- $B1 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1_k$2 -- $B1 > /dev/null 2>&1
done
cp $R/tools/_libvdf_default.so $R/vid_dup_finder_lib_amd/libvdf_hip.so
python3 - <<PY
import csv,glob,collections,os
O="$O"
for d in sorted(glob.glob(O+"/pmc*")):
    fs=glob.glob(d+"/*/*_counter_collection.csv")
    if not fs: print(d,"no data"); continue
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if "hamming_mfma" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(os.path.basename(d), dict(agg))
for d in sorted(glob.glob(O+"/kt_*")):
    fs=glob.glob(d+"/*/*_kernel_stats.csv")
    for r in csv.reader(open(fs[0])):
        if "hamming_mfma" in r[0]: print(os.path.basename(d), r[0][:40], r[1:5])
PY
