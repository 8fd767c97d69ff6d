#!/bin/bash
# round 4, run 11: stride-2 variant of conv_wgrad_ls_kernel (tests, timing, step A/B) + the CU-sharing experiment of VERDICT r3 #3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/open-solution-salt-identification_amd/csrc/_variants
timeout 900 python -m pytest tests/test_gpu_wgrad_ls.py -q -m gpu --tb=short --timeout 300 2>&1 | tail -8
for cfg in "SALT_WGRAD_LS_S2=1" "SALT_WGRAD_LS_S2=0"; do
  echo "== $cfg"; env $cfg timeout 300 python tools/wgrad_ls_bench.py 20 2>&1 | grep "^P\|^sum" | tail -5
done
tools/ab_env2.sh "" "SALT_WGRAD_LS_S2=0" "SALT_WGRAD_LS_S2=1" 2>&1 | tee gpurun_out/r04_step_ab4.log
echo "== CU sharing: conv_ls<NI=1> ring depth 2 (79 KB) beside conv_wgrad_fast8 (59 KB): SALT_WGRAD_LS=0, default lib vs lsd2"
export SALT_WGRAD_LS=0
tools/ab_libs.sh default open-solution-salt-identification_amd/csrc/_variants/libsaltnet_hip.lsd2.so 2>&1 | tee gpurun_out/r04_cu_sharing.log
SALT_LIB=$V/libsaltnet_hip.lsd2.so tools/prof_run.sh r04_lsd2 > /dev/null 2>&1
tools/prof_run.sh r04_lsd4 > /dev/null 2>&1
head -3 gpurun_out/timeline_r04_lsd2.txt gpurun_out/timeline_r04_lsd4.txt
