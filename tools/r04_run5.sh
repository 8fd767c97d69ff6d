#!/bin/bash
# round 4, run 5: step timelines (rocprofv3 kernel trace) with the previous and the new weight-gradient kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SALT_WGRAD_LS=0 tools/prof_run.sh r04_ls0
SALT_WGRAD_LS=1 tools/prof_run.sh r04_ls1
head -4 gpurun_out/timeline_r04_ls0.txt gpurun_out/timeline_r04_ls1.txt
head -30 gpurun_out/streams_r04_ls1.txt
