#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_inference.py tests/test_gpu_conv_ws.py tests/test_gpu_elementwise.py -m gpu -q 2>&1 | tail -5
V=open-solution-salt-identification_amd/csrc/_variants
echo "== A/B (default: bn_bwd 1 unit, affine 2; ew2: bn_bwd 2; ew4: affine 4)"
bash tools/ab_libs.sh default $V/libsaltnet_hip.ew2.so $V/libsaltnet_hip.ew4.so
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["op_time_ms"])'
for nb in 512 1024 2048 4096; do
echo "== SALT_EW_BLOCKS=$nb"; SALT_EW_BLOCKS=$nb timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline --no-iou 2>/dev/null | python -c "$P"
done
echo "== C4"; timeout 600 python tools/bench_c4.py --steps 6 --warmup 2 2>&1 | tail -1
echo "== full suite"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
