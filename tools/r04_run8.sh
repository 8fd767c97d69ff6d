#!/bin/bash
# round 4, run 8: parity hardening tests (fixed-point cubic, conditioned C4, convergence)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export SALT_PARITY_COUNTS=$PWD/gpurun_out/r04_parity_counts.json
rm -f $SALT_PARITY_COUNTS
timeout 900 python -m pytest tests/test_gpu_inference.py -q -m gpu --tb=short --timeout 600 -k "input_pipeline" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_convergence.py -q -m gpu --tb=short --timeout 900 -s 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu --tb=short --timeout 1200 -s -k c4 2>&1 | tail -12
cat $SALT_PARITY_COUNTS
