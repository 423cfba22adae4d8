#!/bin/bash
# Counters of the letterbox detect kernels on pillarboxed 1080p clips (run on the GPU box): bash tools/pmc_letterbox.sh <out_dir> [variant]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-pmc_lb}; mkdir -p $O
[ -n "${2:-}" ] && cp $R/tools/_libvdf_$2.so $R/vid_dup_finder_lib_amd/libvdf_hip.so
cd /tmp; export TMPDIR=/tmp
C="python3 $R/tools/bench_letterbox.py --clips 1000 --w 1920 --h 1080 --bars 0 --side 0.125 --steps 1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $C > $O/run.txt 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $C > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq1 -- $C > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- $C > /dev/null 2>&1
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum --output-format csv -d $O/tcc -- $C > /dev/null 2>&1
[ -n "${2:-}" ] && cp $R/tools/_libvdf_default.so $R/vid_dup_finder_lib_amd/libvdf_hip.so
python3 - <<P
import csv, glob, collections
for d in ("kt","fetch","sq1","sq2","tcc"):
    for f in glob.glob("$O/%s/*/*_kernel_stats.csv" % d):
        for r in csv.DictReader(open(f)):
            if "letterbox" in r["Name"]: print(d, r["Name"][:60], r["Calls"], r["AverageNs"])
    for f in glob.glob("$O/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "letterbox" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0][-40:]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k, cs in agg.items():
            print(d, k, {c: v / len(n[k]) for c, v in cs.items()})
P
