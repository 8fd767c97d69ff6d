#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_conv_ws.py tests/test_gpu_inference.py -m gpu -q -x -k "consumer_loader or align_corners" 2>&1 | tail -8
echo "== fold bench"
for shp in "32 64 64 64" "32 64 128 128" "32 128 32 32" "32 256 16 16"; do timeout 300 python tools/fold_bench.py $shp 2>&1 | tail -10; done
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["val_iou"], d["op_time_ms"])'
for i in 1 2; do
echo "== bench base"; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "== bench SALT_EXP_BN_FOLD (timing only)"; SALT_EXP_BN_FOLD=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "== bench SALT_EXP_VHYPER_SKIP (timing only)"; SALT_EXP_VHYPER_SKIP=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
done
echo "== final conv 320->64 @128: DMA kernels vs register-staged loader"
timeout 300 python tools/conv_bench.py "32,320,128,128,64:0" "32,320,128,128,64:2" "32,320,128,128,64:3" "32,320,128,128,64:1" 2>&1 | tail -6
echo "== full suite"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
