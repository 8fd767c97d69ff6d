#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks.py tests/test_gpu_inference.py tests/test_gpu_conv_ws.py -m gpu -q -x 2>&1 | tail -5
V=open-solution-salt-identification_amd/csrc/_variants
echo "== A/B elementwise units in flight (default: bn_bwd 4 / affine 4; ew2: 2 / 2; ew1: 1 / 2)"
bash tools/ab_libs.sh default $V/libsaltnet_hip.ew2.so $V/libsaltnet_hip.ew1.so
echo "== C4"; timeout 600 python tools/bench_c4.py --steps 6 --warmup 2 2>&1 | tail -1
echo "== C4 interleaved"; SALT_NO_PLANAR=1 timeout 600 python tools/bench_c4.py --steps 6 --warmup 2 2>&1 | tail -1
echo "== bench planar vs not"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["val_iou"], d["op_time_ms"])'
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
SALT_NO_PLANAR=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
done
python tools/op_profile.py --top 400 > gpurun_out/r03f_ops.txt 2>/dev/null
echo "== full suite"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
