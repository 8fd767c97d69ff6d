#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_inference.py -m gpu -q -x -k "planar or hyper_rows or other_sizes" 2>&1 | tail -8
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["val_iou"], d["op_time_ms"])'
for i in 1 2; do
echo "== bench planar (default)"; timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "== bench SALT_NO_PLANAR=1"; SALT_NO_PLANAR=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-configs --no-cpu-baseline 2>/dev/null | python -c "$P"
done
echo "== C4 planar"; timeout 600 python tools/bench_c4.py --steps 6 --warmup 2 2>&1 | tail -1
echo "== C4 interleaved"; SALT_NO_PLANAR=1 timeout 600 python tools/bench_c4.py --steps 6 --warmup 2 2>&1 | tail -1
echo "== C4 planar"; timeout 600 python tools/bench_c4.py --steps 6 --warmup 2 2>&1 | tail -1
echo "== full suite"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python tools/op_profile.py --top 400 > gpurun_out/r03e_ops.txt 2>/dev/null
