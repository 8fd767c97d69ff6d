#!/bin/bash
cd "$(dirname "$0")/.."
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["op_time_ms"])'
for i in 1 2 3; do
echo -n "default: "; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline --no-iou 2>/dev/null | python -c "$P"
echo -n "SALT_NO_BNB_FUSE=1: "; SALT_NO_BNB_FUSE=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline --no-iou 2>/dev/null | python -c "$P"
echo -n "SALT_NO_BNB_RES=1: "; SALT_NO_BNB_RES=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline --no-iou 2>/dev/null | python -c "$P"
done
