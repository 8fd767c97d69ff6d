#!/bin/bash
# A/B of environment switches on another workload: tools/ab_env_wl.sh "<bench args>" "VAR=a" "VAR=b" ...   (each spec run twice, interleaved)
cd "$(dirname "$0")/.."
wl="$1"; shift
for rep in 1 2; do
for spec in "$@"; do
  echo -n "[$wl | $spec] "; env $spec python bench.py $wl --no-cpu-baseline --no-iou --no-configs --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
done
