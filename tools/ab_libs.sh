#!/bin/bash
# A/B of library builds via SALT_LIB inside one box: tools/ab_libs.sh <path.so|default> ...   (each run twice, interleaved)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for so in "$@"; do
  if [ "$so" = default ]; then unset SALT_LIB; else export SALT_LIB=$PWD/$so; fi
  echo -n "[$so] "; python bench.py --no-cpu-baseline --no-iou --no-configs --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['op_time_ms'])"
done
done
