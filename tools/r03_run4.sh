#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_conv_ws.py -m gpu -x -q 2>&1 | tail -12
SH="32,256,16,16,256:0 32,128,32,32,128:0 32,128,64,64,64:0 32,192,32,32,128:0 32,320,16,16,256:0 32,320,128,128,64:0 32,64,64,64,128:0"
echo "== conv_bench ls on"; timeout 300 python tools/conv_bench.py $SH 2>/dev/null
echo "== conv_bench ls off"; SALT_CONV_LS=0 timeout 300 python tools/conv_bench.py $SH 2>/dev/null
for i in 1 2; do
echo "== bench ls on"; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
echo "== bench ls off"; SALT_CONV_LS=0 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
done
