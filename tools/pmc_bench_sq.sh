#!/bin/bash
# SQ counters of the training step's kernels (one PMC pass, --kernel-trace only).  usage: tools/pmc_bench_sq.sh <tag>
tag=${1:-sq}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_$tag -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-iou --no-configs > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
python tools/pmc_summary.py $(find gpurun_out/pmc_$tag -name "*.db" | head -1) > gpurun_out/pmc_${tag}_summary.txt 2>&1
rm -rf gpurun_out/pmc_$tag
grep -A9 "wgrad_kernelItLi9\|conv_mfma_kernelItLi2ELi1ELi2ELi2ELi9" gpurun_out/pmc_${tag}_summary.txt | head -40
