#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_conv_ws.py -m gpu -x -q 2>&1 | tail -8
SH="32,64,64,64,64:0 32,64,128,128,32:0 32,32,128,128,64:0"
echo "== conv_bench ws on"; timeout 300 python tools/conv_bench.py $SH 2>/dev/null
echo "== conv_bench ws off"; SALT_CONV_WS=0 timeout 300 python tools/conv_bench.py $SH 2>/dev/null
V=open-solution-salt-identification_amd/csrc/_variants/libsaltnet_hip.wsclk.so
for s in 32,64,64,64,64 32,32,128,128,64; do echo "=== clocks $s"; SALT_LIB=$V timeout 200 python tools/ws_clocks.py $s 2>/dev/null; done
echo "=== clocks train 32,64,64,64,64"; SALT_LIB=$V timeout 200 python tools/ws_clocks.py 32,64,64,64,64 train 2>/dev/null
for i in 1 2; do
echo "== bench ws on"; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
echo "== bench ws off"; SALT_CONV_WS=0 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['val_iou'], d['op_time_ms'])"
done
