#!/bin/bash
# round 4, run 4: whole-step A/B of conv_wgrad_ls_kernel (same box, alternated) + PMC of the isolated launches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/ab_env2.sh "" "SALT_WGRAD_LS=0" "SALT_WGRAD_LS=1" "SALT_WGRAD_LS=1 SALT_WL_KU=8" "SALT_WGRAD_LS=1 SALT_WGRAD_TPW=16" "SALT_WGRAD_LS=1 SALT_WGRAD_TPW=12" 2>&1 | tee gpurun_out/r04_step_ab.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d /root/repo/gpurun_out/r04_pmc_wl -o pmc --output-format csv -- python /root/repo/tools/wgrad_ls_bench.py 3 > /root/repo/gpurun_out/r04_pmc_wl.log 2>&1
echo "pmc rc=$?"; ls /root/repo/gpurun_out/r04_pmc_wl | head
