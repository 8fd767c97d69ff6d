#!/usr/bin/env python
"""Where the bucketed data-parallel path spends time BEFORE any wire time: the headline step on one GPU, alternated
inside one process between
   plain      net.bwd.run (what a 1-GPU run executes)
   dp         parallel.DataParallel.backward against a 1-rank RCCL communicator (segments, events, collectives)
   dp_nocoll  the same segments and events WITHOUT the collective calls
   dp_1bucket one bucket (no segmentation; the collective after backward)
Prints ms per step of each (3 alternations) and the per-bucket timeline of `dp`.
usage (GPU box): python tools/dp_overhead.py [steps]"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
os.environ['SALT_FORCE_DP_PATH'] = '1'
bench.init_rccl(0)
model, batches, _, _ = bench.train_config('r34_hyper', 'bf16', 32, 'lovasz', 4, 6, dev)
dp = model.dp
from salt_amd import parallel


def run(mode, n):
    os.environ.pop('SALT_FORCE_DP_PATH', None)
    dp.skip_collectives = False
    dp.bucket_bytes = parallel.DEFAULT_BUCKET_BYTES
    if mode != 'plain':
        os.environ['SALT_FORCE_DP_PATH'] = '1'
    if mode == 'dp_nocoll':
        dp.skip_collectives = True
    if mode == 'dp_1bucket':
        dp.bucket_bytes = 1 << 40
    dp.drop_plans()
    dp.measure = False
    for i in range(4):
        model._fit_loop(list(batches[i % 8]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        model._fit_loop(list(batches[i % 8]))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {m: [] for m in ('plain', 'dp', 'dp_nocoll', 'dp_1bucket')}
for rep in range(3):
    for m in res:
        res[m].append(round(run(m, steps), 3))
print(json.dumps(res))
os.environ['SALT_FORCE_DP_PATH'] = '1'
dp.skip_collectives = False
dp.bucket_bytes = parallel.DEFAULT_BUCKET_BYTES
dp.drop_plans()
dp.timeline = True
for i in range(10):
    model._fit_loop(list(batches[i % 8]))
tl = dp.bucket_timeline()
print(json.dumps(tl, indent=1))
