#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=open-solution-salt-identification_amd/csrc/_variants
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_fused_step.py -m gpu -q -x 2>&1 | tail -3
echo "== A/B (default: first loads ahead of the statistics prologue; ewold: HEAD)"
bash tools/ab_libs.sh default $V/libsaltnet_hip.ewold.so
bash tools/ab_libs.sh default $V/libsaltnet_hip.ewold.so
echo "== full suite"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
