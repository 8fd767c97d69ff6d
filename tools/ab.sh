#!/bin/bash
# usage: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one bench run with that environment; use X=1 for the default)
for cfg in "$@"; do echo "== $cfg"; env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iou 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])"; done
