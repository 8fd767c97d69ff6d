#!/bin/bash
# tools/ab_env2.sh "<bench args>" "VAR=a" "VAR=b" ...
cd "$(dirname "$0")/.."
args=$1; shift
for rep in 1 2; do
for spec in "$@"; do
  echo -n "[$spec] "; env $spec python bench.py --no-cpu-baseline --no-iou --no-configs --steps 30 --warmup 8 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['op_time_ms'].get('conv_wgrad'))"
done
done
