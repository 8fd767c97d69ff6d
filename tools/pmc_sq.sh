#!/bin/bash
# SQ counters of the training step per kernel family: two rocprofv3 --pmc passes (--kernel-trace only, no other tracing).  usage: tools/pmc_sq.sh <tag>
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0; dbs=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); d=$R/gpurun_out/pmc_sq_${tag}_$i; rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-iou --no-configs > $R/gpurun_out/pmc_sq_${tag}_$i.log 2>&1
  db=$(find $d -name "*.db" | head -1); [ -n "$db" ] && dbs="$dbs $db"
done
cd $R
SALT_PMC_TRAIN_STEPS=9 python tools/pmc_sq.py $dbs > gpurun_out/${tag}_pmc_sq.json 2> gpurun_out/${tag}_pmc_sq.err
head -c 3000 gpurun_out/${tag}_pmc_sq.json; tail -3 gpurun_out/${tag}_pmc_sq.err; tail -3 gpurun_out/pmc_sq_${tag}_1.log
rm -rf gpurun_out/pmc_sq_${tag}_[0-9]
