#!/bin/bash
# usage: tools/r06_ab.sh <tag> "<VAR=a>" "<VAR=b>" [reps]   - alternated C2 headline runs (bench.py, no extra legs) for a same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
tag=$1; A=$2; Bv=$3; reps=${4:-3}
for rep in $(seq $reps); do for v in "$A" "$Bv"; do
  echo "== $v"
  env $v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-iou --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
done; done > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
