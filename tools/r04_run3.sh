#!/bin/bash
# round 4, run 3: conv_wgrad_ls_kernel v2 (buffer-addressed DMA, rolling row buffers, cross-entry prefetch)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/open-solution-salt-identification_amd/csrc/_variants
timeout 900 python -m pytest tests/test_gpu_wgrad_ls.py -q -m gpu --tb=short --timeout 300 > gpurun_out/r04_wl_tests.log 2>&1
echo "tests rc=$?"; tail -n 8 gpurun_out/r04_wl_tests.log
for cfg in "SALT_WL_KU=4" "SALT_WL_KU=8" "SALT_WGRAD_LS=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/wgrad_ls_bench.py 20 2>&1 | grep "^P\|^sum" | tee -a gpurun_out/r04_wl_bench.log
done
for a in 4 1 2 3; do
  echo "== SALT_WL_ABLATE=$a"
  SALT_LIB=$V/libsaltnet_hip.wlab$a.so timeout 300 python tools/wgrad_ls_bench.py 20 2>&1 | grep "^P\|^sum" | tee -a gpurun_out/r04_wl_ablate.log
done
for s in 32,64,64,64,64 32,16,16,256,256 32,8,8,512,512; do
  SALT_LIB=$V/libsaltnet_hip.wlclk.so timeout 300 python tools/wl_clocks.py $s 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_wl_clocks.log
done
