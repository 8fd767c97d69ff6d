#!/bin/bash
# Round 5: the factored hypercolumn - new parity tests, the whole-network parity tests that now run through it, and same-box A/B
# of the step / the C4 pass with SALT_HYPER_FACTOR = 4 (default) / 0 (round 4's materialised hypercolumn) / 2 (level 2 factored too).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_hyper_factor.py -m gpu -q -x --tb=short > $O/r05_hyper_tests.log 2>&1
tail -15 $O/r05_hyper_tests.log
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_fused_step.py -m gpu -q --tb=short > $O/r05_net_tests.log 2>&1
tail -8 $O/r05_net_tests.log
for rep in 1 2; do
for f in 4 0 2; do
  echo "== SALT_HYPER_FACTOR=$f"
  SALT_HYPER_FACTOR=$f python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-iou --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'])"
done
done > $O/r05_hyper_ab.txt 2>&1
for f in 4 0 2; do
  echo "== SALT_HYPER_FACTOR=$f"
  SALT_HYPER_FACTOR=$f python tools/bench_c4.py 2>/dev/null
done >> $O/r05_hyper_ab.txt 2>&1
cat $O/r05_hyper_ab.txt
