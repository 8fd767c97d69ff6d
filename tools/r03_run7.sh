#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do
echo "== bench new"; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
echo "== bench ws/ls off"; SALT_CONV_LS=0 SALT_CONV_WS=0 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
done
python tools/op_profile.py --top 400 > gpurun_out/r03d_ops.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
