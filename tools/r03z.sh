#!/bin/bash
mkdir -p gpurun_out/r03z
VDF_FUZZ_SEEDS=60 python -m pytest tests/test_gpu_letterbox.py tests/test_gpu_fuzz.py tests/test_gpu_hash_parity.py tests/test_gpu_hash_queue.py tests/test_golden.py tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert|^FAILED" | head -20 > gpurun_out/r03z/tests.log
cat gpurun_out/r03z/tests.log
