#!/usr/bin/env python
"""One-off sweep of the CPU baseline leg (the oracle on the host cores) over thread counts and batch sizes, so that bench.py's
`cpu_baseline` uses what the box actually delivers.  usage: python tools/cpu_sweep.py > profiles/rNN_cpu_sweep.json"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ['SALT_CPU_LEG_STEPS'], os.environ['SALT_CPU_LEG_WARMUP'] = '4', '1'
import bench
bench.CPU_LEG_TIMEOUT_S = 120
ncpu = os.cpu_count() or 1
rows = []
for arch in ('UNetResNet', 'VanillaUNet'):
    for batch in (8, 32):
        for thr in (8, 16, 32, 64, 128, 256):
            if thr > ncpu:
                continue
            r = bench.cpu_baseline_subprocess('lovasz', arch, thr, batch)
            rows.append({'arch': arch, 'batch': batch, 'threads': thr, 'images_per_s': r['value'], 'note': r['sample'][:80]})
            sys.stderr.write(json.dumps(rows[-1]) + '\n')
best = {}
for r in rows:
    if r['images_per_s'] and (r['arch'] not in best or r['images_per_s'] > best[r['arch']]['images_per_s']):
        best[r['arch']] = r
print(json.dumps({'host_logical_cpus': ncpu, 'rows': rows, 'best': best}, indent=1))
