#!/bin/bash
# the configurations quoted in DESIGN.md / README (one line each)
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-44s %9.1f img/s %7.3f ms/step  iou %s  conv %.0f TF/s' % (' '.join(sys.argv[1:]) or 'default', d['value'], d['ms_per_step'], d.get('val_iou'), d['roofline']['achieved']))" "$@"; }
run
run --dtype f32
run --batch 64
run --workload ternaus34
run --workload ternaus34 --dtype f32
run --workload vanilla --dtype f32
run --workload vanilla
run --loss bce_dice
python tools/bench_c4.py 2>/dev/null | tail -1
python tools/bench_c4.py --dtype f32 --steps 3 --warmup 1 2>/dev/null | tail -1
python tools/bench_c4.py --depth 34 --size 128 --batch 32 2>/dev/null | tail -1
