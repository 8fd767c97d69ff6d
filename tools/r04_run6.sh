#!/bin/bash
# round 4, run 6: what the weight-gradient stream costs the step, and CU partitioning between the two streams
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/ab_env2.sh "" "SALT_WGRAD_LS=1" "SALT_WGRAD_LS=1 SALT_TIMING_ONLY=1 SALT_EXP_NO_WGRAD=1" "SALT_WGRAD_LS=1 SALT_CONV_WPX_BWD=24" "SALT_WGRAD_LS=1 SALT_CONV_WPX_BWD=16" "SALT_WGRAD_LS=1 SALT_WGRAD_WGS=256" "SALT_WGRAD_LS=1 SALT_WGRAD_WGS=256 SALT_WGRAD_TPW=4" "SALT_WGRAD_LS=1 SALT_WGRAD_TPW=12 SALT_CONV_WPX_BWD=24" 2>&1 | tee gpurun_out/r04_step_ab2.log
