#!/bin/bash
# round 4, run 7: split-rule sweep of conv_wgrad_ls_kernel inside the step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/ab_env2.sh "" "SALT_WGRAD_TPW=8" "SALT_WGRAD_TPW=12" "SALT_WGRAD_TPW=12 SALT_WGRAD_WGS=256" "SALT_WGRAD_TPW=16 SALT_WGRAD_WGS=256" "SALT_WGRAD_TPW=10 SALT_WGRAD_WGS=384" "SALT_WGRAD_TPW=12 SALT_WL_KU=8" 2>&1 | tee gpurun_out/r04_step_ab3.log
