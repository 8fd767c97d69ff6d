#!/bin/bash
# A/B of environment switches inside one box: tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   (each spec run twice, interleaved)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for spec in "$@"; do
  echo -n "[$spec] "; env $spec python bench.py --no-cpu-baseline --no-iou --no-configs --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
done
