#!/bin/bash
# Run the GPU test files one process each (a faulting kernel must not hide the other results).
# usage: tools/gpu_suite.sh [pytest -k expression]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
K="$1"
rc=0
for f in tests/test_gpu_loss_optim.py tests/test_gpu_inference.py tests/test_gpu_trainer.py tests/test_gpu_blocks.py tests/test_gpu_models.py; do
  n=$(basename $f .py)
  if [ -n "$K" ]; then
    timeout 900 python -m pytest $f -q -m gpu --tb=short --timeout 300 -k "$K" > gpurun_out/$n.log 2>&1
  else
    timeout 900 python -m pytest $f -q -m gpu --tb=short --timeout 300 > gpurun_out/$n.log 2>&1
  fi
  r=$?
  [ $r -ne 0 ] && rc=$r
  echo "== $n rc=$r"; tail -n 3 gpurun_out/$n.log
done
exit $rc
