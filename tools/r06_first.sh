#!/bin/bash
# round 6, first GPU call: the new tests + the fused-step / trainer suites + a zero-copy A/B of the headline step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_blocks.py tests/test_gpu_baseline_configs.py tests/test_gpu_models.py tests/test_gpu_fused_step.py tests/test_gpu_trainer.py \
  -m gpu -q -x -k "two_ranks or idempotent or batch_tiles or c3_batch_64 or c2_shape or c1_shape or fused or trainer or fit or bucketed or scse" > $O/r06_first_tests.log 2>&1
tail -15 $O/r06_first_tests.log
for rep in 1 2; do for z in 1 0; do
  echo "== SALT_STEP_ZERO_COPY=$z"
  SALT_STEP_ZERO_COPY=$z python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-iou --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['roofline'].get('kernel_symbol'))"
done; done > $O/r06_zero_copy_ab.txt 2>&1
cat $O/r06_zero_copy_ab.txt
