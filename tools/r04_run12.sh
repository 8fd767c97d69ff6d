#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bilinear_rows.py -q -m gpu --tb=short --timeout 600 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_blocks.py -q -m gpu --tb=short --timeout 600 -k "upsample or block_vs_reference" 2>&1 | tail -4
tools/ab_env2.sh "" "SALT_BILINEAR_ROWS=0" "SALT_BILINEAR_ROWS=1" 2>&1 | tee gpurun_out/r04_step_ab5.log
for r in 0 1; do SALT_BILINEAR_ROWS=$r python bench.py --no-cpu-baseline --no-iou --no-configs --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows=$r', d['ms_per_step'], d['op_time_ms'].get('bilinear'))"; done
for r in 0 1; do SALT_BILINEAR_ROWS=$r timeout 600 python tools/bench_c4.py 2>/dev/null | tail -2; done
