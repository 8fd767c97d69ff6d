#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export SALT_PARITY_COUNTS=$PWD/gpurun_out/r04_parity_counts_c4.json
rm -f $SALT_PARITY_COUNTS
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu --tb=short --timeout 1200 -s -k c4 2>&1 | tail -12
cat $SALT_PARITY_COUNTS
