#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel stats + PMC passes for bench.py and the
# hash micro-bench.  Every profiler call has its own timeout (a TA_* counter pass once hung a whole call).
# Usage: tools/profile_round.sh <out_dir under gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
sha256sum $R/vid_dup_finder_lib_amd/libvdf_hip.so > $O/lib_sha256.txt   # which binary these counters belong to (bench.py: read_traffic)
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-queue --no-windowed --c4-hashes 0 --c5-cands 0 --cache-entries 0 --no-valu --no-refs"  # headline + dup_heavy + hash legs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench_under_profiler.json 2> /dev/null
B1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --c5-cands 0 --dup-heavy 0 --cache-entries 0 --no-valu --no-refs"
H1="python3 $R/tools/bench_hash.py --steps 1"
H2="python3 $R/tools/bench_hash.py --steps 1 --clips 1000 --w 1920 --h 1080"   # the bench's full_hd leg (linear-stream kernel)
H3="python3 $R/tools/bench_hash.py --steps 1 --clips 4000 --w 480 --h 270"     # the bench's pitch_480x270 leg
H4="python3 $R/tools/bench_hash.py --steps 1 --clips 250 --w 3840 --h 2160"    # the bench's uhd_3840x2160 leg (K-split kernel)
# the headline alone (3 timed + 2 warm-up launches of ONE kernel shape): its rocprofv3 average is the figure bench.py's roofline.kernel_ms must agree with
B3="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --hash-clips 0 --no-windowed --c4-hashes 0 --c5-cands 0 --dup-heavy 0 --cache-entries 0 --no-valu --no-refs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_headline -- $B3 > $O/bench_headline_under_profiler.json 2> /dev/null
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_search -- $B1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_search -- $B1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_hash -- $H1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_hash -- $H1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_hd -- $H2 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_hd -- $H2 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_sd -- $H3 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_sd -- $H3 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_uhd -- $H4 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_uhd -- $H4 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq_uhd -- $H4 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq_hd -- $H2 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq_search -- $B1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq_search2 -- $B1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq_hash -- $H1 > /dev/null 2>&1
# round 6: the fused small-frame letterbox kernel (20 000 clips of 16 x 64 x 64, six bar patterns, three entry points each)
H5="python3 $R/tools/bench_letterbox_small.py --child --clips 20000 --steps 1 --check 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lbs -- $H5 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_lbs -- $H5 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_lbs -- $H5 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq_lbs -- $H5 > /dev/null 2>&1
ls $O
