#!/bin/bash
cd "$(dirname "$0")/.."
for v in 0 1 0 1; do
  echo -n "SALT_STEP_GRAPH=$v: "; SALT_STEP_GRAPH=$v python bench.py --no-cpu-baseline --no-iou --no-configs --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
