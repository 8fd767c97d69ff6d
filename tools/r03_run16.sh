#!/bin/bash
cd "$(dirname "$0")/.."
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["op_time_ms"])'
for i in 1 2 3; do
for t in 8 16 32 12; do
echo -n "SALT_WGRAD_TPW=$t: "; SALT_WGRAD_TPW=$t timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline --no-iou 2>/dev/null | python -c "$P"
done; done
