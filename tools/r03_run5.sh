#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_conv_ws.py -m gpu -q 2>&1 | tail -6
tools/prof_run.sh r03b_new
SALT_CONV_LS=0 SALT_CONV_WS=0 tools/prof_run.sh r03b_old
for t in r03b_new r03b_old; do echo "=== $t"; head -3 gpurun_out/timeline_$t.txt; cat gpurun_out/streams_$t.txt | head -34; done
