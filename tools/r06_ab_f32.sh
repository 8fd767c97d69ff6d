#!/bin/bash
# usage: tools/r06_ab_f32.sh <tag> <reps> "<VAR=a>" ...   - alternated fp32 runs (C1 vanilla, R34 hypercolumn) over several settings
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
tag=$1; reps=$2; shift 2
for rep in $(seq $reps); do for v in "$@"; do
  echo "== $v"
  for w in vanilla r34_hyper; do
  env $v python bench.py --workload $w --dtype f32 --steps 30 --warmup 8 --no-cpu-baseline --no-iou --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done; done > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
