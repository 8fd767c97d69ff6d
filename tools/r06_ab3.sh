#!/bin/bash
# usage: tools/r06_ab3.sh <tag> <reps> "<VAR=a>" "<VAR=b>" ...   - alternated C2 headline runs over several settings (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
tag=$1; reps=$2; shift 2
for rep in $(seq $reps); do for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-iou --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
done; done > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
