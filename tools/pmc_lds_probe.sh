set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ldspmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --hash-clips 0 --no-windowed"
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/a0 -- $B1 > /dev/null 2>&1
export VDF_MFMA_ABLATE=2
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/a2 -- $B1 > /dev/null 2>&1
unset VDF_MFMA_ABLATE
timeout 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $O/v0 -- $B1 > /dev/null 2>&1
find $O -name "*counter_collection.csv" | head
