#!/bin/bash
python -m pytest tests/test_gpu_search_parity.py -k "pinned" -m gpu -x -q 2>&1 | tail -40
