#!/bin/bash
# HBM traffic of the training step from rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots,
# FETCH_SIZE takes 3 and WRITE_SIZE 2), each with --kernel-trace only.  usage: tools/pmc_traffic.sh <tag>
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${tag}_$c
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${tag}_$c -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-iou --no-configs > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
cd $R
SALT_PMC_TRAIN_STEPS=9 python tools/pmc_traffic.py $(find gpurun_out/pmc_${tag}_FETCH_SIZE -name "*.db" | head -1) $(find gpurun_out/pmc_${tag}_WRITE_SIZE -name "*.db" | head -1) > gpurun_out/pmc_traffic_${tag}.json
cat gpurun_out/pmc_traffic_${tag}.json | head -60
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
