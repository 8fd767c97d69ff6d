#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_inference.py -m gpu -q -x -k "residual_epilogue or transform_values or align_corners or input_pipeline" 2>&1 | tail -8
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["val_iou"], d["op_time_ms"])'
for i in 1 2; do
echo "== bench slabs"; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "== bench wgrad atomics"; SALT_WGRAD_ATOMIC=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
done
for shp in "32 64 64 64 64" "32 64 128 128 32" "32 128 32 32 128" "32 256 16 16 256" "32 512 8 8 512"; do
echo "== wgrad micro slabs $shp"; timeout 300 python tools/wgrad_micro.py $shp 2>&1 | grep wgrad
echo "== wgrad micro atomics $shp"; SALT_WGRAD_ATOMIC=1 timeout 300 python tools/wgrad_micro.py $shp 2>&1 | grep wgrad
done
echo "== wgrad parity under atomics"; SALT_WGRAD_ATOMIC=1 timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k wgrad 2>&1 | tail -5
echo "== full suite"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
