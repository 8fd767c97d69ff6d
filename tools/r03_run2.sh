#!/bin/bash
# round 3, GPU call 2: conv_ws_kernel - parity tests, then micro-benchmark and whole-step A/B against conv_mfma_kernel
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_ws.py -m gpu -x -q 2>&1 | tail -15
SH="32,64,64,64,64:0 32,64,128,128,32:0 32,32,128,128,64:0"
echo "== conv_bench ws on"; timeout 300 python tools/conv_bench.py $SH 2>/dev/null
echo "== conv_bench ws off"; SALT_CONV_WS=0 timeout 300 python tools/conv_bench.py $SH 2>/dev/null
for i in 1 2; do
echo "== bench ws on"; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
echo "== bench ws off"; SALT_CONV_WS=0 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
done
