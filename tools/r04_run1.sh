#!/bin/bash
# round 4, run 1: first GPU contact of conv_wgrad_ls_kernel - parity tests, then isolated timing new / old
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wgrad_ls.py -q -m gpu --tb=short --timeout 300 -x > gpurun_out/r04_wl_tests.log 2>&1
echo "tests rc=$?"; tail -n 15 gpurun_out/r04_wl_tests.log
for cfg in "SALT_WGRAD_LS=1 SALT_WL_KU=4" "SALT_WGRAD_LS=1 SALT_WL_KU=8" "SALT_WGRAD_LS=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/wgrad_ls_bench.py 20 2>&1 | tee -a gpurun_out/r04_wl_bench.log
done
