#!/bin/bash
cd "$(dirname "$0")/.."
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["op_time_ms"])'
for i in 1 2 3; do
for t in 2 8 4 1; do
echo -n "SALT_WGRAD_GENERIC_TPW=$t: "; SALT_WGRAD_GENERIC_TPW=$t timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline --no-iou 2>/dev/null | python -c "$P"
done; done
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_fused_step.py tests/test_gpu_models.py -m gpu -q -x 2>&1 | tail -3
